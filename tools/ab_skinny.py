#!/usr/bin/env python3
"""Time per launch of the quantised Linear across small token counts (GPU box): streaming GEMV (M <= 4), skinny MFMA
kernel (1..32), prefill GEMM, hipGraph replay over distinct weight copies (HBM-cold), Llama shapes.
    python tools/ab_skinny.py [--ms 1,4,5,8,16,32,64]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="1,4,5,8,16,32,33,48,64")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008")
    ap.add_argument("--compact", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for N, K in [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]:
        g = torch.Generator(device=dev).manual_seed(N + K)
        base = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half(), compact_meta=args.compact)
        nw = max(2, int(640e6 / base.nbytes()) + 1)
        ws = [base] + [packing.PackedMXQ(base.qweight.clone(), base.rowmeta, N, K, base.compact) for _ in range(nw - 1)]
        for M in [int(m) for m in args.ms.split(",")]:
            x = torch.randn(M, K, generator=g, device=dev).half()
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            parts = []
            for path in ("gemv", "skinny", "gemm"):
                if (path == "gemv" and M > 4) or (path == "skinny" and M > 64):
                    continue
                packing.linear(x, base, out=out, path=path)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for p in ws:
                        packing.linear(x, p, out=out, path=path)
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    gr.replay()
                    e0.record()
                    gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / nw * 1e3)
                t = sorted(ts)[2]
                parts.append(f"{path} {t:6.2f}us ({base.nbytes() / t / 1e6:.2f} TB/s)")
            print(f"N={N} K={K} M={M:3d}: " + "  ".join(parts), flush=True)


if __name__ == "__main__":
    main()
