#!/usr/bin/env python3
"""Race screen for the mid-M kernel (GPU box): the kernel is deterministic by construction (slabs summed in slice order), so
ANY difference between repeated launches on the same inputs is a synchronisation bug (a fragment read that overtook its
DMA, a slab read before it landed).  Repeats every (shape, token count) many times -- alone, and with a second stream
hammering HBM beside it to shift the DMA timing -- and requires bit-identical outputs that also match the fp32 reference.
    python tools/midm_stress.py [--reps 300]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import packing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    noise = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    bad = 0
    for N, K in [(4096, 4096), (11008, 4096), (4096, 11008), (208, 2176), (4096, 192)]:
        g = torch.Generator(device=dev).manual_seed(N + K)
        for compact in (False, True):
            p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half(), compact_meta=compact)
            wd = packing.dequant(p).float()
            for M in (21, 49, 64, 100, 128, 200, 256):
                x = torch.randn(M, K, generator=g, device=dev).half()
                ref = x.float() @ wd.t()
                first = packing.linear(x, p).clone()
                err = ((first.float() - ref).abs().max() / ref.abs().max()).item()
                assert err <= 1e-3, (N, K, M, compact, err)
                diff = 0
                for r in range(args.reps):
                    if r % 3 == 0:                      # every third launch runs beside a 512-MB fill on another stream
                        with torch.cuda.stream(side):
                            noise.fill_(r & 255)
                    y = packing.linear(x, p)
                    diff += int(not torch.equal(y, first))
                torch.cuda.synchronize()
                bad += diff
                print(f"N={N} K={K} M={M} compact={compact}: max-rel {err:.1e}, {diff} of {args.reps} repeats differ", flush=True)
    print("RACE SCREEN", "FAILED" if bad else "clean", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
