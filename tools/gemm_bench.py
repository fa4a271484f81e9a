#!/usr/bin/env python3
"""A/B timing of the GEMM kernel variants on the Llama-2-7B Linear shapes (GPU box).
Interleaved rounds in ONE process (cdna guide rule 24); random data (rule 25).

    python tools/gemm_bench.py [--m 2048] [--rounds 5] [--variants gemm1,gemm8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import packing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--variants", default="gemm1,gemm8")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008", help="NxK list")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    variants = args.variants.split(",")
    shapes = [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]
    M = args.m
    for N, K in shapes:
        g = torch.Generator(device=dev).manual_seed(N + K)
        W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
        p = packing.quantize_pack(W)
        wd = packing.dequant(p)
        x = torch.randn(M, K, generator=g, device=dev).half()
        yref = x.float() @ wd.float().t()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        res = {v: [] for v in variants}
        for v in variants:
            y = packing.linear(x, p, path=v).float()
            err = ((y - yref).abs().max() / yref.abs().max()).item()
            assert err < 1e-3, (v, err)
        for _ in range(args.rounds):
            for v in variants:
                for _ in range(3):
                    packing.linear(x, p, out=out, path=v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    packing.linear(x, p, out=out, path=v)
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / args.iters * 1e3)
        fl = 2.0 * M * N * K
        # fp16 dense reference (hipBLASLt through torch) on the dequantised weight, same data
        ts = []
        for _ in range(args.rounds):
            for _ in range(3):
                torch.matmul(x, wd.t())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                torch.matmul(x, wd.t())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.iters * 1e3)
        ts.sort()
        line = f"M={M} N={N} K={K}: " + "  ".join(
            f"{v} med {sorted(t)[len(t)//2]:.1f}us min {min(t):.1f}us = {fl/sorted(t)[len(t)//2]/1e6:.0f} TF/s"
            for v, t in res.items())
        print(line + f"  | torch fp16 matmul {ts[len(ts)//2]:.1f}us = {fl/ts[len(ts)//2]/1e6:.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
