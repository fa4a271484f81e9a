#!/usr/bin/env python3
"""Same-process A/B of the MXAsymQuantizer forward of two builds of libmxq_hip.so (GPU box): one Llama-2-7B decoder
block's seven weights (BASELINE configs[3]) per pass, GPU-side time under hipGraph replay, outputs compared bit for bit.

    python tools/ab_fakequant.py [--dtype bf16] [--libs mxq_amd/libmxq_hip.so,abtmp/libmxq_hip_before.so]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import llama_shapes as LS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--libs", default="mxq_amd/libmxq_hip.so,abtmp/libmxq_hip_before.so")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--shape", default=None, help="N,K: four weights of this one shape instead of the decoder block")
    args = ap.parse_args()
    dt = {"bf16": (torch.bfloat16, 2), "f16": (torch.float16, 1), "f32": (torch.float32, 0)}[args.dtype]
    dev = torch.device("cuda:0")
    H, I = LS.HIDDEN, LS.INTERMEDIATE
    shapes = [(H, H)] * 4 + [(I, H)] * 2 + [(H, I)]
    if args.shape:
        shapes = [tuple(int(v) for v in args.shape.split(","))] * 4
    g = torch.Generator(device=dev).manual_seed(0)
    ws = [(torch.randn(n, k, generator=g, device=dev) * 0.02).to(dt[0]) for n, k in shapes]
    outs = [torch.empty_like(w) for w in ws]
    nbytes = sum(2 * w.numel() * w.element_size() for w in ws)
    names = args.libs.split(",")
    graphs, ref = {}, None
    for name in names:
        lib = ctypes.CDLL(os.path.join(ROOT, name))
        fwd = lib.mxq_fakequant_fwd
        fwd.restype = ctypes.c_int
        fwd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

        def run(fwd=fwd):
            st = torch.cuda.current_stream().cuda_stream
            for w, o in zip(ws, outs):
                rc = fwd(w.data_ptr(), o.data_ptr(), w.shape[0], w.shape[1], 2, dt[1], st)
                assert rc == 0, rc
        run()
        torch.cuda.synchronize()
        as_int = torch.int32 if dt[0] == torch.float32 else torch.int16
        if ref is None:
            ref = [o.clone() for o in outs]
        else:
            assert all(torch.equal(a.view(as_int), b.view(as_int)) for a, b in zip(ref, outs)), f"{name}: output differs"
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            run()
        graphs[name] = cg
    ts = {n: [] for n in names}
    for _ in range(args.rounds):
        for n in names:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graphs[n].replay()
            e0.record()
            graphs[n].replay()
            e1.record()
            torch.cuda.synchronize()
            ts[n].append(e0.elapsed_time(e1) * 1e3)
    for n in names:
        t = sorted(ts[n])
        med = t[len(t) // 2]
        print(f"{n:40s}: {med:7.1f} us per pass ({nbytes / med / 1e6:.2f} TB/s of read + write), min {t[0]:.1f}")


if __name__ == "__main__":
    main()
