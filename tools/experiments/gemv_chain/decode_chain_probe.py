#!/usr/bin/env python3
"""The GEMV chain of a decoder layer (o_proj -> gate|up -> down -> next q|k|v) as four launches and as ONE chained launch
(csrc/gemv_chain.hip): hipGraph replay over 8 distinct layers, us per layer.   python tools/decode_chain_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import _lib, packing  # noqa: E402
from mxq_amd import llama_shapes as LS  # noqa: E402

if len(sys.argv) > 1:          # another build of the library (tools/build_variant.sh): timing experiments
    _lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])


def main():
    dev = torch.device("cuda:0")
    H, I = LS.HIDDEN, LS.INTERMEDIATE
    g = torch.Generator(device=dev).manual_seed(0)
    L = 8
    def mk(n, k):
        return packing.quantize_pack((torch.randn(n, k, generator=g, device=dev) * 0.02).half())
    layers = [dict(o=mk(H, H), gu=mk(2 * I, H), down=mk(H, I), qkv=mk(3 * H, H)) for _ in range(L)]
    a = torch.randn(1, H, generator=g, device=dev).half()
    h0 = torch.randn(1, H, generator=g, device=dev).half()
    nw = torch.ones(H, device=dev, dtype=torch.float16)

    def separate(w):
        h1 = packing.linear_fused(a, w["o"], 0, residual=h0)
        gg = packing.linear_fused(h1, w["gu"], 1, nw)
        h2 = packing.linear_fused(gg, w["down"], 2, residual=h1)
        return packing.linear_fused(h2, w["qkv"], 1, nw)

    def chained(w, n=4):
        ops = [(a, w["o"], 0, None, 1e-5, h0), ("y0", w["gu"], 1, nw, 1e-5, None), ("y1", w["down"], 2, None, 1e-5, "y0"),
               ("y2", w["qkv"], 1, nw, 1e-5, None)]
        return packing.linear_chain(ops[:n])[-1]

    def timed(fn):
        for w in layers:
            fn(w)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for w in layers:
                fn(w)
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

    same = torch.equal(separate(layers[0]), chained(layers[0]))
    t_sep = timed(separate)
    t_ch = timed(chained)
    t_ch3 = timed(lambda w: chained(w, 3))
    print("chains of 1 / 2 / 3 / 4 ops: " + " / ".join(f"{timed(lambda w, n=n: chained(w, n)):.1f}" for n in (1, 2, 3, 4)) + " us", flush=True)
    err = int(packing.chain_workspace(dev)[-1].item())
    print(f"four launches {t_sep:6.1f} us per layer   one chained launch {t_ch:6.1f} us   (chain of the first three: {t_ch3:.1f})   error word {err}   last output equal to the separate launches': {same}", flush=True)


if __name__ == "__main__":
    main()
