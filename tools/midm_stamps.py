#!/usr/bin/env python3
"""Where a workgroup of the mid-M kernel spends its time (GPU box; profiling build libmxq_hip_prof.so): per-workgroup
wall-clock stamps at start / prologue published / K loop done / output stored, and the launch time of the two step
orders (0 = every wave converts first, 1 = SIMD partners staggered).

    python tools/midm_stamps.py [--shape 4096x4096] [--m 128] [--bm 0] [--splits 0]
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


def pct(t, q):
    return t.float().quantile(q).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x4096")
    ap.add_argument("--m", type=int, default=128)
    ap.add_argument("--bm", type=int, default=0)
    ap.add_argument("--splits", type=int, default=0)
    args = ap.parse_args()
    N, K = (int(v) for v in args.shape.split("x"))
    M = args.m
    dev = torch.device("cuda:0")
    prof = ctypes.CDLL(os.path.join(ROOT, "mxq_amd", "libmxq_hip_prof.so"))
    prof.mxq_prof_midm_set.restype = ctypes.c_int
    prof.mxq_prof_midm_set.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn = prof.mxq_prof_midm_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    g = torch.Generator(device=dev).manual_seed(1)
    base = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
    nw = max(2, int(400e6 / base.nbytes()) + 1)
    wl = [base] + [packing.PackedMXQ(base.qweight.clone(), base.rowmeta, N, K) for _ in range(nw - 1)]
    x = torch.randn(M, K, generator=g, device=dev).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    ws = packing.gemm_workspace(dev)

    def call(i):
        rc = fn(x.data_ptr(), wl[i].qweight.data_ptr(), wl[i].rowmeta.data_ptr(), out.data_ptr(), M, N, K, args.bm,
                args.splits, ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
        assert rc == 0, rc

    for order in (0, 1, 0, 1):
        assert prof.mxq_prof_midm_set(None, order) == 0
        print(f"order {order}: {timed(call, nw):.2f} us per launch pair (graph replay, HBM-cold weights)", flush=True)
    # in-kernel cycle sums per wave (s_memtime), both orders
    prof.mxq_prof_midm_set_cycles.restype = ctypes.c_int
    prof.mxq_prof_midm_set_cycles.argtypes = [ctypes.c_void_p]
    cyc = torch.zeros(4096 * 8 * 6, dtype=torch.int64, device=dev)
    for order in (0, 1):
        cyc.zero_()
        assert prof.mxq_prof_midm_set(None, order) == 0 and prof.mxq_prof_midm_set_cycles(cyc.data_ptr()) == 0
        for i in range(3):
            call(i + 1)
        torch.cuda.synchronize()
        prof.mxq_prof_midm_set_cycles(None)
        c = cyc.view(4096, 8, 6).cpu()
        c = c[c[:, 0, 5] > 0].float()
        per = c[:, :, :5] / c[:, :, 5:6]
        print(f"order {order}: cycles per double-step and wave (median over {c.shape[0]} workgroups; {int(c[0, 0, 5])} steps)")
        for w in range(8):
            m = per[:, w].median(0).values
            print(f"  wave {w}: issue {m[0]:6.0f}  convert+publish {m[1]:6.0f}  multiply {m[2]:6.0f}  vmcnt wait {m[3]:6.0f}  "
                  f"barrier {m[4]:6.0f}  sum {m.sum():6.0f}")
    nwg = 4096
    stamps = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
    for order in (0, 1):
        stamps.zero_()
        assert prof.mxq_prof_midm_set(stamps.data_ptr(), order) == 0
        for i in range(3):
            call(i + 1)                                   # the last launch's stamps stay
        torch.cuda.synchronize()
        prof.mxq_prof_midm_set(None, order)
        t = stamps.view(nwg, 4).cpu()
        t = t[t[:, 0] > 0] * 10                           # ns; workgroups that ran
        rel = (t - t[:, 0].min()).float() / 1e3
        print(f"order {order}: {t.shape[0]} workgroups")
        print(f"  start ramp : median {pct(rel[:, 0], .5):.2f}  p90 {pct(rel[:, 0], .9):.2f}  max {rel[:, 0].max():.2f} us")
        for name, a, b in (("prologue", 0, 1), ("K loop", 1, 2), ("output", 2, 3), ("whole wg", 0, 3)):
            d = rel[:, b] - rel[:, a]
            print(f"  {name:11s}: median {pct(d, .5):.2f}  p10 {pct(d, .1):.2f}  p90 {pct(d, .9):.2f}  max {d.max():.2f} us")
        print(f"  finish     : median {pct(rel[:, 3], .5):.2f}  p90 {pct(rel[:, 3], .9):.2f}  max {rel[:, 3].max():.2f} us after the first start")


if __name__ == "__main__":
    main()
