#!/usr/bin/env python3
"""Same-process A/B of prefill GEMM kernels and their profiling builds (GPU box).

GPU-side time per launch under hipGraph replay (20 launches per graph, no host launch overhead in the number),
variants interleaved over several rounds (cdna guide rule 24), random data (rule 25).

    python tools/ab_gemm.py --variants gemm1,gemm8,abl8:4,torch [--m 2048] [--shapes 4096x4096,...]

variant names: midm:BM:S = the mid-M kernel at an explicit tile height / slice count (profiling library's entry, correct
results); gemmN = product kernel through mxq_gemm_f16_ws (checked against the fp32 matmul on the bit-exact
dequantised weight); ablK:B = libmxq_hip_prof.so's mxq_prof_gemmK_ablate_f16 with ablation bits B (WRONG results by
construction, timing only; `make -C mxq_amd/csrc prof`); torch = torch.matmul on the dequantised fp16 weight
(hipBLASLt), the dense yardstick.
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402

_prof = {}


def prof_lib(path="mxq_amd/libmxq_hip_prof.so"):
    if path not in _prof:
        _prof[path] = ctypes.CDLL(os.path.join(ROOT, path))
    return _prof[path]


def make_call(v, x, p, wd, out):
    if v == "torch":
        return lambda: torch.matmul(x, wd.t(), out=out)
    if v.startswith("abl"):
        k, bits, *lib = v[3:].split(":")      # ablK:B[:path of another profiling build]
        fn = getattr(prof_lib(*lib), f"mxq_prof_gemm{k}_ablate_f16")
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        M = x.shape[0]

        def call():
            rc = fn(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, p.N, p.K, int(bits),
                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (v, rc)
        return call
    if v.startswith("midm:"):        # midm:BM:S -- the mid-M kernel with an explicit tile height (64 | 128) and slice count (0 = by CUs)
        _, bm, sl = v.split(":")
        fn = prof_lib().mxq_prof_midm_f16
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        ws = packing.gemm_workspace(x.device)
        M = x.shape[0]

        def call():
            rc = fn(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, p.N, p.K, int(bm), int(sl),
                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (v, rc)
        return call
    if v == "dense":                 # the hoisted mode's MFMA kernel alone, on the dequantised weight
        fn = prof_lib().mxq_prof_gemm8_dense_f16
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
        M = x.shape[0]

        def call():
            rc = fn(x.data_ptr(), wd.data_ptr(), out.data_ptr(), M, p.N, p.K,
                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (v, rc)
        return call
    if v.startswith("xlib:"):                               # xlib:PATH:symbol -- an experiment's dense entry (x, w16, y, M, N, K, stream)
        _, so, sym = v.split(":")
        fn = getattr(prof_lib(so), sym)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
        M = x.shape[0]

        def call():
            rc = fn(x.data_ptr(), wd.data_ptr(), out.data_ptr(), M, p.N, p.K, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (v, rc)
        return call
    if v.startswith("dlib:"):                               # dlib:PATH[:variant] -- another build's mxq_dense_f16 (default 2 = dense256)
        _, so, *var = v.split(":")
        fn = prof_lib(so).mxq_dense_f16
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        M = x.shape[0]

        def call():
            rc = fn(x.data_ptr(), wd.data_ptr(), out.data_ptr(), M, p.N, p.K, int(var[0]) if var else 2,
                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (v, rc)
        return call
    if v.startswith("hlib:") or v.startswith("lib:"):      # another build of libmxq_hip.so (tools/build_variant.sh)
        kind, so, *var = v.split(":")
        lib = prof_lib(so)
        M = x.shape[0]
        if kind == "hlib":                                  # hlib:PATH -- its mxq_linear_f16_hoisted
            fn = lib.mxq_linear_f16_hoisted
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            scratch = packing.hoist_scratch(x.device, packing._lib.load().mxq_hoist_scratch_bytes(p.N, p.K))
            a = (x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, p.N, p.K, 0,
                 scratch.data_ptr(), scratch.numel())
        else:                                               # lib:PATH[:variant] -- its mxq_gemm_f16_ws (default gemm8)
            fn = lib.mxq_gemm_f16_ws
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            ws = packing.gemm_workspace(x.device)
            a = (x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, p.N, p.K,
                 int(var[0]) if var else 8, ws.data_ptr(), ws.numel())

        def call():
            rc = fn(*a, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (v, rc)
        return call
    if v in packing.DENSE_VARIANTS and v != "auto":        # the plain GEMM on the dequantised weight: dense128 / dense256
        return lambda: packing.linear_dense(x, wd, out=out, variant=v)
    if v == "hoist":
        return lambda: packing.linear_hoisted(x, p, out=out)
    if v == "fused":
        return lambda: packing.linear(x, p, out=out, path="fused")
    return lambda: packing.linear(x, p, out=out, path=v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="gemm8,torch")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008", help="NxK list")
    ap.add_argument("--json", default=None)
    ap.add_argument("--reps", type=int, default=20, help="launches per graph")
    ap.add_argument("--nocheck", action="store_true", help="skip the result check (debug builds with parts switched off)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    variants = args.variants.split(",")
    M = args.m
    report = []
    for N, K in [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]:
        g = torch.Generator(device=dev).manual_seed(N + K)
        p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
        wd = packing.dequant(p)
        x = torch.randn(M, K, generator=g, device=dev).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        yref = x.float() @ wd.float().t()
        graphs = {}
        for v in variants:
            call = make_call(v, x, p, wd, out)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            if not v.startswith("abl") and not args.nocheck:
                err = ((out.float() - yref).abs().max() / yref.abs().max()).item()
                assert err < 1e-3, (v, N, K, err)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(args.reps):
                    call()
            graphs[v] = gr
        ts = {v: [] for v in variants}
        for _ in range(args.rounds):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                graphs[v].replay()
                e0.record()
                graphs[v].replay()
                e1.record()
                torch.cuda.synchronize()
                ts[v].append(e0.elapsed_time(e1) / args.reps * 1e3)
        fl = 2.0 * M * N * K
        row = {"M": M, "N": N, "K": K}
        parts = []
        for v in variants:
            t = sorted(ts[v])
            med = t[len(t) // 2]
            row[v] = {"us_med": round(med, 2), "us_min": round(t[0], 2), "tflops": round(fl / med / 1e6, 1)}
            parts.append(f"{v} {med:.1f}us ({fl / med / 1e6:.0f} TF)")
        report.append(row)
        print(f"M={M} N={N} K={K}: " + "  ".join(parts), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
