#!/usr/bin/env python3
"""Endurance / fuzz run of every split-K schedule the library ships (GPU box): random (tokens, out, in) shapes, layouts and paths
-- the 256 x 128 / 128 x 128 / 128 x 64 / 64 x 128 builds of the fused kernel with their stream-K tail forced or chosen, the slices modes, the mid-M
kernel, the product dispatch -- each result against the fp32 product on the bit-exact dequantised weight, launched twice for
bit-identical output, the workspace's counter head checked to be zero again.  A progress line every 50 cases.

    python tools/stress_streamk.py --seconds 600 [--seed 1] > gpurun_out/stress.log"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402

PATHS = ["auto", "gemm8", "gemm9", "gemm8h", "gemm8h_split", "gemm8h_slices", "gemm8q_split", "gemm8q_slices", "gemm8n_split",
         "gemm8n_slices", "midm"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    g = torch.Generator(device=dev).manual_seed(args.seed)
    ws = packing.gemm_workspace(dev)
    t0 = time.time()
    n = 0
    worst = 0.0
    by_path = {}
    while time.time() - t0 < args.seconds:
        kind = rng.integers(0, 4)
        if kind == 0:      # Llama shapes, token counts around every dispatch boundary
            N, K = [(4096, 4096), (11008, 4096), (4096, 11008)][rng.integers(0, 3)]
            M = int(rng.choice([21, 41, 48, 64, 65, 100, 128, 129, 192, 256, 257, 300, 384, 512, 640, 768, 1000, 1024, 1536, 2048]))
        elif kind == 1:    # anything
            M = int(rng.integers(5, 2200))
            N = 16 * int(rng.integers(1, 700))
            K = 64 * int(rng.choice([1, 2, 3, 4, 8, 16, 33, 64, 100, 172]))
        elif kind == 2:    # few tiles, long K: many contributors per tile
            M = int(rng.integers(5, 400))
            N = 16 * int(rng.integers(1, 96))
            K = 64 * int(rng.choice([64, 128, 172, 256]))
        else:              # just over a round of tiles: the small-tail branches
            M = int(rng.choice([256, 512, 640, 768, 1024, 1280, 1536]))
            N = 128 * int(rng.choice([33, 43, 65, 86, 129, 131, 172, 258]))
            K = 64 * int(rng.choice([16, 64, 100]))
        layout = ["mixed", "mixed", "mixedc", "w2g16", "w4row"][rng.integers(0, 5)]
        W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
        if layout in ("mixed", "mixedc"):
            p = packing.quantize_pack(W)
            wd = packing.dequant(p)
            if layout == "mixedc":
                p = packing.compact(p)
                wd = packing.dequant(p)
            path = PATHS[rng.integers(0, len(PATHS))] if layout == "mixed" else "auto"
        else:
            p = packing.quantize_pack_uniform(W, layout)
            wd = packing.expand_uniform(p, codes=False)[0]
            path = "auto"
        x = torch.randn(M, K, generator=g, device=dev).half()
        run = (lambda: packing.linear(x, p, path=path)) if layout == "mixed" else (lambda: packing.linear_layout(x, p, path="auto"))
        y = run()
        err = 0.0
        for n0 in range(0, N, 4096):
            r = x.float() @ wd[n0:n0 + 4096].float().t()
            err = max(err, ((y[:, n0:n0 + 4096].float() - r).abs().max() / r.abs().max().clamp(min=1e-6)).item())
        if not err <= 1e-3:
            print(f"FAIL error {err:.3e}: M={M} N={N} K={K} layout={layout} path={path}", flush=True)
            sys.exit(1)
        if not torch.equal(run(), y):
            print(f"FAIL not deterministic: M={M} N={N} K={K} layout={layout} path={path}", flush=True)
            sys.exit(1)
        if int(ws[:32768].view(torch.int32).abs().sum().item()) != 0:
            print(f"FAIL counters not zero: M={M} N={N} K={K} layout={layout} path={path}", flush=True)
            sys.exit(1)
        # the status words of a wait that gave up (the last 16 bytes of the head) must stay zero: an expiry is a failure here
        if int(ws[65536 - 16:65536].view(torch.int32).abs().sum().item()) != 0:
            print(f"FAIL stream-K wait expired {ws[65536 - 16:65536].view(torch.int32).tolist()}: M={M} N={N} K={K} layout={layout} "
                  f"path={path}", flush=True)
            sys.exit(1)
        worst = max(worst, err)
        by_path[path + "/" + layout] = by_path.get(path + "/" + layout, 0) + 1
        n += 1
        if n % 50 == 0:
            print(f"{n} cases, {time.time() - t0:.0f} s, worst rel err {worst:.2e}", flush=True)
    packing.workspace_status(dev)          # every cached workspace (raises on a flagged one)
    print(f"DONE {n} cases in {time.time() - t0:.0f} s, worst rel err {worst:.2e}, all deterministic, counters and status words zero", flush=True)
    print("cases by path/layout:", dict(sorted(by_path.items())), flush=True)


if __name__ == "__main__":
    main()
