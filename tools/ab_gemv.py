#!/usr/bin/env python3
"""Same-process A/B of the decode GEMV kernels (GPU box): GPU-side time per launch under hipGraph replay over
enough DISTINCT weights that every launch streams from HBM (>= 600 MB per graph: the 256 MiB Infinity Cache
would otherwise serve the replays), variants interleaved over several rounds.

    python tools/ab_gemv.py [--m 1] [--variants v1,v2:0,v2:2,v2:4] [--shapes 4096x4096,...]

variants: v1 = the product GEMV (mxq_gemv_f16), v1:T = the same with T threads per workgroup (libmxq_hip_prof.so),
lib:PATH = mxq_gemv_f16 of another build of libmxq_hip.so (repo-relative path),
v2:T = tools/experiments/gemv2.hip with T teams per workgroup (a 16-byte-load remap that measured ~10 % SLOWER: kept
as a record, not built by default), torch = fp16 torch matmul on the dequantised weight.  Every variant is checked against the
fp32 product on the bit-exact dequantised weight (<= 1e-3)."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="v1,v1:256,v1:512,v1:1024,torch")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008,12288x4096,22016x4096", help="NxK list")
    ap.add_argument("--mb", type=float, default=640.0, help="distinct packed bytes per graph (MB)")
    ap.add_argument("--repeat", type=int, default=1, help="walk the distinct weights this many times per graph (with a "
                    "small --mb: cache-resident weights without the graph-launch overhead dominating)")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    prof = ctypes.CDLL(os.path.join(ROOT, "mxq_amd", "libmxq_hip_prof.so"))
    fn2 = getattr(prof, "mxq_prof_gemv2_f16", None)     # only in a build that links tools/experiments/gemv2.hip
    if fn2 is not None:
        fn2.restype = ctypes.c_int
        fn2.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    fn1 = prof.mxq_prof_gemv_f16          # v1 kernel with an explicit workgroup size: "v1:256" / "v1:512" / "v1:1024"
    fn1.restype = ctypes.c_int
    fn1.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    variants = args.variants.split(",")
    M = args.m
    report = []
    for N, K in [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]:
        g = torch.Generator(device=dev).manual_seed(N + K)
        base = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
        nw = max(2, int(args.mb * 1e6 / base.nbytes()) + 1)
        # distinct buffers with the same (valid) contents: what matters is that the addresses differ
        ws = [base] + [packing.PackedMXQ(base.qweight.clone(), base.rowmeta.clone(), N, K) for _ in range(nw - 1)]
        wd = packing.dequant(base)
        x = torch.randn(M, K, generator=g, device=dev).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        yref = x.float() @ wd.float().t()
        wds = None
        graphs = {}
        for v in variants:
            if v == "torch":
                if wds is None:
                    wds = [wd] + [wd.clone() for _ in range(min(nw, int(args.mb * 1e6 / (N * K * 2)) + 1) - 1)]
                calls = [(lambda w=w: torch.matmul(x, w.t(), out=out)) for w in wds]
            elif v.startswith("lib:"):      # another build of the product library (e.g. abtmp/libmxq_hip_before.so)
                other = ctypes.CDLL(os.path.join(ROOT, v[4:]))
                fo = other.mxq_gemv_f16
                fo.restype = ctypes.c_int
                fo.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]

                def mko(p):
                    def call():
                        rc = fo(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, N, K,
                                torch.cuda.current_stream().cuda_stream)
                        assert rc == 0, (v, rc)
                    return call
                calls = [mko(p) for p in ws]
            elif v == "v1":
                calls = [(lambda p=p: packing.linear(x, p, out=out, path="gemv")) for p in ws]
            elif v.startswith("v1:"):
                th = int(v.split(":")[1])

                def mk1(p):
                    def call():
                        rc = fn1(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, N, K, th,
                                 torch.cuda.current_stream().cuda_stream)
                        assert rc == 0, (v, rc)
                    return call
                calls = [mk1(p) for p in ws]
            else:
                teams = int(v.split(":")[1])

                def mk(p):
                    def call():
                        rc = fn2(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, N, K, 0,
                                 teams, torch.cuda.current_stream().cuda_stream)
                        assert rc == 0, (v, rc)
                    return call
                calls = [mk(p) for p in ws]
            calls[0]()
            torch.cuda.synchronize()
            err = ((out.float() - yref).abs().max() / yref.abs().max()).item()
            assert err < 1e-3, (v, N, K, err)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(args.repeat):
                    for c in calls:
                        c()
            graphs[v] = (gr, len(calls) * args.repeat)
        ts = {v: [] for v in variants}
        for _ in range(args.rounds):
            for v in variants:
                gr, n = graphs[v]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                gr.replay()
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                ts[v].append(e0.elapsed_time(e1) / n * 1e3)
        row = {"M": M, "N": N, "K": K, "packed_MB": round(base.nbytes() / 1e6, 2), "weights_per_graph": nw}
        parts = []
        for v in variants:
            t = sorted(ts[v])
            med = t[len(t) // 2]
            byts = N * K * 2 if v == "torch" else base.nbytes()
            row[v] = {"us_med": round(med, 2), "us_min": round(t[0], 2), "TBps": round(byts / med / 1e6, 3)}
            parts.append(f"{v} {med:.2f}us ({byts / med / 1e6:.2f} TB/s)")
        report.append(row)
        print(f"M={M} N={N} K={K} ({base.nbytes() / 1e6:.1f} MB x {nw}): " + "  ".join(parts), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
