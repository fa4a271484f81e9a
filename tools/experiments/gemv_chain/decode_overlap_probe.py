#!/usr/bin/env python3
"""Upper bound for what hiding launch ramps / tails could buy the decode layer: the four GEMV launches of a decoder layer
(RMSNorm -> q|k|v, o_proj + residual, RMSNorm -> gate|up, SwiGLU -> down + residual) run (a) in stream order, as the decode
loop runs them, and (b) as four INDEPENDENT branches of one hipGraph (no data dependencies -- not a valid layer, timing only).
(b) is what a dependency-free overlap of ramps and tails would reach.   python tools/decode_overlap_probe.py"""
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
    L = 8                                   # distinct layers, so that weights stream from HBM
    def mk(n, k):
        return packing.quantize_pack((torch.randn(n, k, generator=g, device=dev) * 0.02).half())
    layers = [dict(qkv=mk(3 * H, H), o=mk(H, H), gu=mk(2 * I, H), down=mk(H, I)) for _ in range(L)]
    x = torch.randn(1, H, generator=g, device=dev).half()
    xi = torch.randn(1, 2 * I, generator=g, device=dev).half()
    nw = torch.ones(H, device=dev, dtype=torch.float16)
    res = torch.randn(1, H, generator=g, device=dev).half()

    def layer(w, streams=None):
        calls = [lambda: packing.linear_fused(x, w["qkv"], prologue=1, norm_w=nw),
                 lambda: packing.linear_fused(x, w["o"], residual=res),
                 lambda: packing.linear_fused(x, w["gu"], prologue=1, norm_w=nw),
                 lambda: packing.linear_fused(xi, w["down"], prologue=2, residual=res)]
        if streams is None:
            for c in calls:
                c()
        else:
            cur = torch.cuda.current_stream()
            for s, c in zip(streams, calls):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    c()
            for s in streams:
                cur.wait_stream(s)

    streams = [torch.cuda.Stream() for _ in range(4)]
    for name, st in (("stream order", None), ("four independent branches", streams)):
        for w in layers:
            layer(w, st)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for w in layers:
                layer(w, st)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / L * 1e3)
        print(f"{name:28s}: {sorted(ts)[3]:6.1f} us per layer (4 GEMV launches, {sum(p.nbytes() for p in layers[0].values()) / 1e6:.1f} MB packed)", flush=True)


if __name__ == "__main__":
    main()
