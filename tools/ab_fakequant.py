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


_HIP = None


def _memcpy(dst, src):
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
        _HIP.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    rc = _HIP.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size(), 3,
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--libs", default="mxq_amd/libmxq_hip.so,abtmp/libmxq_hip_before.so")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--bwd", action="store_true", help="time the straight-through backward (3 tensors per element) instead")
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
    nbytes = sum((3 if args.bwd else 2) * w.numel() * w.element_size() for w in ws)
    gos = [torch.randn(w.shape, generator=g, device=dev).to(dt[0]) for w in ws] if args.bwd else None
    names = args.libs.split(",")
    graphs, ref = {}, None
    for name in names:
        lib = ctypes.CDLL(os.path.join(ROOT, name))
        fwd = lib.mxq_fakequant_fwd
        fwd.restype = ctypes.c_int
        fwd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

        bwd = lib.mxq_fakequant_bwd
        bwd.restype = ctypes.c_int
        bwd.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]

        def run(fwd=fwd, bwd=bwd):
            st = torch.cuda.current_stream().cuda_stream
            for i, (w, o) in enumerate(zip(ws, outs)):
                if args.bwd:
                    rc = bwd(gos[i].data_ptr(), w.data_ptr(), o.data_ptr(), w.numel(), -0.03, 0.03, dt[1], st)
                else:
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
    # the transfer floor of the same pass: a plain device copy of the same seven tensors (torch's elementwise copy kernel
    # and hipMemcpyAsync device-to-device) -- what "one read + one write per element" costs with no arithmetic at all
    for name, fn in (("copy: torch out.copy_(w)", lambda w, o: o.copy_(w)),
                     ("copy: hipMemcpyAsync D2D", lambda w, o: _memcpy(o, w))):
        for w, o in zip(ws, outs):
            fn(w, o)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            for w, o in zip(ws, outs):
                fn(w, o)
        graphs[name] = cg
        names.append(name)
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
        print(f"{n:40s}: {med:7.1f} us per pass ({nbytes / med / 1e6:.2f} TB/s of algorithmic bytes), min {t[0]:.1f}")


if __name__ == "__main__":
    main()
