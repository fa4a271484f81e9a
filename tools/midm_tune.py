#!/usr/bin/env python3
"""Sweep the mid-M kernel's tile height (64 | 128 tokens) and K-slice count per (shape, M) on the GPU box, through the
profiling library's explicit entry point (libmxq_hip_prof.so: mxq_prof_midm_f16; correct results, checked <= 1e-3
against the fp32 product on the dequantised weight).  hipGraph replay over HBM-cold weight copies, median of 5.

    python tools/midm_tune.py [--ms 64,128,256,512,1024] [--splits 1,2,4,8,16] [--bms 64,128]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402
from tools.midm_bench import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="64,128,256,512,1024")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008")
    ap.add_argument("--splits", default="0,1,2,4,8,16")
    ap.add_argument("--bms", default="64,128")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    prof = ctypes.CDLL(os.path.join(ROOT, "mxq_amd", "libmxq_hip_prof.so"))
    fn = prof.mxq_prof_midm_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    ws = packing.gemm_workspace(dev)
    st = torch.cuda.current_stream(dev)
    for N, K in [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]:
        g = torch.Generator(device=dev).manual_seed(N + K)
        base = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
        nw = max(2, int(400e6 / base.nbytes()) + 1)
        wl = [base] + [packing.PackedMXQ(base.qweight.clone(), base.rowmeta, N, K) for _ in range(nw - 1)]
        wd = packing.dequant(base).float()
        for M in [int(m) for m in args.ms.split(",")]:
            x = torch.randn(M, K, generator=g, device=dev).half()
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            yref = x.float() @ wd.t()
            cells = []
            for bm in [int(v) for v in args.bms.split(",")]:
                for sp in [int(v) for v in args.splits.split(",")]:
                    def call(i):
                        rc = fn(x.data_ptr(), wl[i].qweight.data_ptr(), wl[i].rowmeta.data_ptr(), out.data_ptr(), M, N, K,
                                bm, sp, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
                        assert rc == 0, rc
                    call(0)
                    torch.cuda.synchronize()
                    err = ((out.float() - yref).abs().max() / yref.abs().max()).item()
                    assert err <= 1e-3, (bm, sp, err)
                    cells.append(f"bm{bm}/s{sp}={timed(call, nw):.1f}")
            print(f"N={N} K={K} M={M}: " + "  ".join(cells), flush=True)


if __name__ == "__main__":
    main()
