#!/usr/bin/env python3
"""Fixed cost per decode GEMV launch, measured: the three K = 4096 weights of a decoder layer (q|k|v 12288, o_proj 4096,
gate|up 22016 rows) as three launches and as ONE launch over their row-concatenation (timing only: not a valid layer).
hipGraph replay over 8 distinct layers (weights stream from HBM).   python tools/decode_concat_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402
from mxq_amd import llama_shapes as LS  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H, I = LS.HIDDEN, LS.INTERMEDIATE
    g = torch.Generator(device=dev).manual_seed(0)
    L = 8
    def mk(n, k):
        return packing.quantize_pack((torch.randn(n, k, generator=g, device=dev) * 0.02).half())
    sep = [[mk(3 * H, H), mk(H, H), mk(2 * I, H)] for _ in range(L)]
    cat = [packing.concat_packed(ws) for ws in sep]
    x = torch.randn(1, H, generator=g, device=dev).half()
    nw = torch.ones(H, device=dev, dtype=torch.float16)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / L * 1e3)
        return sorted(ts)[3]

    mb = sum(p.nbytes() for p in sep[0]) / 1e6
    t3 = timed(lambda: [packing.linear_fused(x, p, prologue=1, norm_w=nw) for ws in sep for p in ws])
    t1 = timed(lambda: [packing.linear_fused(x, p, prologue=1, norm_w=nw) for p in cat])
    each = [timed(lambda i=i: [packing.linear_fused(x, ws[i], prologue=1, norm_w=nw) for ws in sep]) for i in range(3)]
    print(f"three launches {t3:6.1f} us  (alone: {each[0]:.1f} + {each[1]:.1f} + {each[2]:.1f})   one launch over the concatenation {t1:6.1f} us   "
          f"{mb:.1f} MB packed -> {mb / t3 / 1e0:.2f} vs {mb / t1:.2f} MB/us", flush=True)


if __name__ == "__main__":
    main()
