#!/usr/bin/env python3
"""Time per launch of the quantised Linear in the mid-size token regime (48 < M <= 1024), GPU box.

For every (Llama shape, M): the product dispatch ("auto": what mxq_linear_f16_ws picks), explicit kernels, and
PyTorch's fp16 GEMM (hipBLASLt) on the dequantised weight -- the 16-bit baseline the packed Linear has to beat
(reference analogue: the split-K launcher gemm_cuda_gen.cu:429-475).  hipGraph replay over
enough distinct weight copies that every launch streams its weight from HBM (no L2 / Infinity-Cache reuse
between launches), median of 5 replays.

    python tools/midm_bench.py [--ms 64,128,256,512,1024] [--paths auto,midm,gemm8,skinny] [--json out.json]
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


def timed(fn_for_copy, n_copies, reps=5):
    for i in range(min(2, n_copies)):
        fn_for_copy(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(n_copies):
            fn_for_copy(i)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr.replay()
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n_copies * 1e3)
    return sorted(ts)[len(ts) // 2]


_LIBS = {}


def lib_linear(path, x, p, out, variant=10):
    """The quantised Linear through ANOTHER build of libmxq_hip.so (path "lib:<repo-relative .so>[:variant]"; variant
    of mxq_gemm_f16_ws, default 10 = the mid-M kernel): same-process A/B of builds (tools/build_variant.sh)."""
    lib = _LIBS.get(path)
    if lib is None:
        lib = _LIBS[path] = ctypes.CDLL(os.path.join(ROOT, path))
        lib.mxq_gemm_f16_ws.restype = ctypes.c_int
        lib.mxq_gemm_f16_ws.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_size_t,
                                                                                    ctypes.c_void_p]
    ws = packing.gemm_workspace(x.device, counters=variant != 10)
    rc = lib.mxq_gemm_f16_ws(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), x.shape[0], p.N, p.K,
                             variant, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(x.device).cuda_stream)
    assert rc == 0, rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="64,128,256,512,1024")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008")
    ap.add_argument("--paths", default="auto,gemm8,skinny")
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-torch", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for N, K in [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]:
        g = torch.Generator(device=dev).manual_seed(N + K)
        base = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
        nw = max(2, int(400e6 / base.nbytes()) + 1)          # > the 256 MiB Infinity Cache in packed bytes
        ws = [base] + [packing.PackedMXQ(base.qweight.clone(), base.rowmeta, N, K) for _ in range(nw - 1)]
        w16 = packing.dequant(base)
        n16 = max(2, int(400e6 / (N * K * 2)) + 1)
        w16s = [w16] + [w16.clone() for _ in range(n16 - 1)]
        for M in [int(m) for m in args.ms.split(",")]:
            x = torch.randn(M, K, generator=g, device=dev).half()
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            row = {"N": N, "K": K, "M": M}
            for path in args.paths.split(","):
                if path == "skinny" and M > 64:
                    continue
                try:
                    if path.startswith("lib:"):
                        so, _, var = path[4:].partition(":")
                        call = lambda i: lib_linear(so, x, ws[i], out, int(var or 10))
                        call(0)
                        err = ((out.float() - x.float() @ w16.float().t()).abs().max() / (x.float() @ w16.float().t()).abs().max()).item()
                        assert err <= 1e-3 or "abl" in so, (path, err)          # (ablation builds are wrong by construction)
                        row[path] = round(timed(call, nw), 2)
                        continue
                    row[path] = round(timed(lambda i: packing.linear(x, ws[i], out=out, path=path), nw), 2)
                except ValueError as e:       # a path this build does not have
                    row[path] = None
                    print(f"# {path}: {e}", flush=True)
            if not args.no_torch:
                row["torch_f16"] = round(timed(lambda i: torch.mm(x, w16s[i].t(), out=out), n16), 2)
            flop = 2.0 * M * N * K
            best = min(v for k, v in row.items() if k in args.paths.split(",") and v)
            row["auto_TFLOPs"] = round(flop / (row.get("auto") or best) / 1e6, 1)
            rows.append(row)
            print("  ".join(f"{k}={v}" for k, v in row.items()), flush=True)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
