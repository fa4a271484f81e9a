#!/usr/bin/env python3
"""Probe of the persistent decode-layer kernel (decode_engine.hip; build.sh first): o_proj -> gate|up -> down -> next q|k|v
in one launch against the four fused GEMV launches: results (<= 1e-3 of the output scale; the summation order differs),
error word, and us per layer under hipGraph replay over 8 distinct layers.
    python tools/experiments/decode_engine/engine_probe.py [--compact]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402
from mxq_amd import llama_shapes as LS  # noqa: E402


class Op(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("qweight", ctypes.c_void_p), ("rowmeta", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("norm_w", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("N", ctypes.c_int), ("K", ctypes.c_int),
                ("prologue", ctypes.c_int), ("eps", ctypes.c_float)]


def main():
    compact = "--compact" in sys.argv
    lib = ctypes.CDLL(os.path.join(ROOT, "abtmp", "lib_decode_engine.so"))
    lib.mxq_exp_decode_engine_ws_bytes.restype = ctypes.c_size_t
    fn = lib.mxq_exp_decode_engine_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(Op), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    H, I = LS.HIDDEN, LS.INTERMEDIATE
    g = torch.Generator(device=dev).manual_seed(0)
    L = 8
    def mk(n, k):
        return packing.quantize_pack((torch.randn(n, k, generator=g, device=dev) * 0.02).half(), compact_meta=compact)
    layers = [dict(o=mk(H, H), gu=mk(2 * I, H), down=mk(H, I), qkv=mk(3 * H, H)) for _ in range(L)]
    a = torch.randn(1, H, generator=g, device=dev).half()
    h0 = torch.randn(1, H, generator=g, device=dev).half()
    nw = (1.0 + 0.1 * torch.randn(H, generator=g, device=dev)).half()
    ws = torch.zeros(lib.mxq_exp_decode_engine_ws_bytes() // 4, dtype=torch.int32, device=dev)
    bufs = [dict(h1=torch.empty(1, H, device=dev, dtype=torch.float16), gg=torch.empty(1, 2 * I, device=dev, dtype=torch.float16),
                 h2=torch.empty(1, H, device=dev, dtype=torch.float16), q=torch.empty(1, 3 * H, device=dev, dtype=torch.float16))
            for _ in range(L)]

    def separate(w):
        h1 = packing.linear_fused(a, w["o"], 0, residual=h0)
        gg = packing.linear_fused(h1, w["gu"], 1, nw)
        h2 = packing.linear_fused(gg, w["down"], 2, residual=h1)
        return h1, gg, h2, packing.linear_fused(h2, w["qkv"], 1, nw)

    def engine(w, b, n=4):
        ops = (Op * 4)()
        P = lambda t: t.data_ptr() if t is not None else None
        spec = [(a, w["o"], b["h1"], None, h0, 0), (b["h1"], w["gu"], b["gg"], nw, None, 1), (b["gg"], w["down"], b["h2"], None, b["h1"], 2),
                (b["h2"], w["qkv"], b["q"], nw, None, 1)]
        for i, (x, p, y, n_w, res, pro) in enumerate(spec):
            ops[i] = Op(P(x), P(p.qweight), P(p.rowmeta), P(y), P(n_w), P(res), p.N, p.K, pro, 1e-5)
        rc = fn(ops, n, int(compact), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return b["h1"], b["gg"], b["h2"], b["q"]

    want = separate(layers[0])
    for rep in range(3):
        got = engine(layers[0], bufs[0])
        torch.cuda.synchronize()
        errs = [((x.float() - y.float()).abs().max() / x.float().abs().max()).item() for x, y in zip(want, got)]
        print(f"run {rep}: max rel err per op (h1, gate|up, h2, qkv) = " + ", ".join(f"{e:.2e}" for e in errs) + f"   error word {int(ws[-32].item())}   generation {int(ws[-64].item())}", flush=True)
        assert max(errs) <= 1e-3, errs

    def timed(fn_):
        for i in range(L):
            fn_(i)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(L):
                fn_(i)
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

    t_sep = timed(lambda i: separate(layers[i]))
    t_e4 = timed(lambda i: engine(layers[i], bufs[i], 4))
    t_e3 = timed(lambda i: engine(layers[i], bufs[i], 3))
    print(f"four launches {t_sep:6.1f} us per layer   engine (4 ops) {t_e4:6.1f} us   engine (3 ops, no q|k|v) {t_e3:6.1f} us   error word {int(ws[-32].item())}", flush=True)


if __name__ == "__main__":
    main()
