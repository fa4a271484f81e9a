#!/usr/bin/env python3
"""GPU-side time per launch of GEMM kernel variants / profiling builds, measured by replaying a hipGraph of
20 launches (no host launch overhead in the number).
    python tools/gemm_graph_bench.py --variants gemm1,gemm8 [--m 2048] [--shapes 4096x4096,...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import packing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--variants", default="gemm1,gemm8")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008", help="NxK list")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for N, K in [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]:
        g = torch.Generator(device=dev).manual_seed(N + K)
        p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
        x = torch.randn(args.m, K, generator=g, device=dev).half()
        out = torch.empty(args.m, N, device=dev, dtype=torch.float16)
        row = []
        for path in args.variants.split(","):
            for _ in range(3):
                packing.linear(x, p, out=out, path=path)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(20):
                    packing.linear(x, p, out=out, path=path)
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            ts.sort()
            row.append(f"{path} {ts[3]:.1f}us")
        print(f"M={args.m} N={N} K={K}: " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
