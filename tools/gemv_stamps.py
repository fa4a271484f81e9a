#!/usr/bin/env python3
"""Where does a decode GEMV launch spend its time?  (GPU box; needs `make -C mxq_amd/csrc prof`)

Runs the fused one-token GEMV of the profiling library over distinct weights (so that every launch streams from
HBM) with per-workgroup phase stamps (100 MHz wall clock: start, activations staged, weight loop done, stored) and
prints, for the last launch, the start ramp, the duration of each phase and when the workgroups finish relative to
the first start.

    python tools/gemv_stamps.py [--shape 22016x4096] [--prologue 1]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402


def pct(t, q):
    return t.float().quantile(q).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="22016x4096")
    ap.add_argument("--prologue", type=int, default=1)
    ap.add_argument("--copies", type=int, default=14)
    ap.add_argument("--threads", type=int, default=0, help="workgroup size (128 / 256 / 512; 0 = the launcher's choice)")
    args = ap.parse_args()
    N, K = (int(v) for v in args.shape.split("x"))
    dev = torch.device("cuda:0")
    prof = ctypes.CDLL(os.path.join(ROOT, "mxq_amd", "libmxq_hip_prof.so"))
    prof.mxq_prof_gemv_set_stamps.restype = ctypes.c_int
    prof.mxq_prof_gemv_set_stamps.argtypes = [ctypes.c_void_p]
    fn = prof.mxq_prof_gemv_fused_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p,
                                                               ctypes.c_int, ctypes.c_void_p]
    g = torch.Generator(device=dev).manual_seed(1)
    base = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
    ws = [base] + [packing.PackedMXQ(base.qweight.clone(), base.rowmeta.clone(), N, K) for _ in range(args.copies - 1)]
    xin = torch.randn(1, 2 * K if args.prologue == 2 else K, generator=g, device=dev).half()
    norm_w = torch.ones(K, device=dev, dtype=torch.float16)
    out = torch.empty(1, N, device=dev, dtype=torch.float16)
    nwg = N // 16
    stamps = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
    assert prof.mxq_prof_gemv_set_stamps(stamps.data_ptr()) == 0
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        e0.record()
        for p in ws:
            rc = fn(xin.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), N, K, args.prologue,
                    norm_w.data_ptr(), 1e-5, None, args.threads, st)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
    print(f"N={N} K={K} prologue={args.prologue}: {e0.elapsed_time(e1) / len(ws) * 1e3:.2f} us per launch "
          f"(stream-ordered launches, stamps on), {nwg} workgroups, {base.nbytes() / 1e6:.1f} MB")
    prof.mxq_prof_gemv_set_stamps(None)
    t = stamps.view(nwg, 4).cpu() * 10          # ns
    t = t[t[:, 0] > 0]                          # the workgroups that ran: beyond 768 row blocks a workgroup takes TWO row blocks
    nwg = t.shape[0]                            # (RB = 2, csrc/gemv.hip gemv_shape), so the launch has half as many workgroups
    print(f"({nwg} workgroups stamped)")
    t0 = t[:, 0].min()
    rel = (t - t0).float() / 1e3                 # us since the first workgroup's start
    print(f"start ramp   : median {pct(rel[:, 0], .5):.2f}  p90 {pct(rel[:, 0], .9):.2f}  max {rel[:, 0].max():.2f} us")
    for name, a, b in (("stage x", 0, 1), ("weight loop", 1, 2), ("reduce+store", 2, 3), ("whole wg", 0, 3)):
        d = rel[:, b] - rel[:, a]
        print(f"{name:13s}: median {pct(d, .5):.2f}  p10 {pct(d, .1):.2f}  p90 {pct(d, .9):.2f}  max {d.max():.2f} us")
    print(f"finish       : median {pct(rel[:, 3], .5):.2f}  p10 {pct(rel[:, 3], .1):.2f}  p90 {pct(rel[:, 3], .9):.2f}  "
          f"max {rel[:, 3].max():.2f} us after the first start")
    # who finishes late?  blocks b and b + 8 share an XCD (label b % 8); blocks are dealt to CUs in order
    fin = rel[:, 3]
    by_xcd = [fin[e::8] for e in range(8)]
    print("finish by XCD label (median / max us): " + "  ".join(f"{e}: {pct(v, .5):.1f}/{v.max():.1f}" for e, v in enumerate(by_xcd)))
    q = max(1, nwg // 8)
    print("finish by launch-order octile (median us): " + "  ".join(f"{pct(fin[i * q:(i + 1) * q], .5):.1f}" for i in range(8)))
    st = rel[:, 1] - rel[:, 0]
    print("stage-x by XCD label (median / max us):  " + "  ".join(f"{e}: {pct(st[e::8], .5):.1f}/{st[e::8].max():.1f}" for e in range(8)))
    late = (rel[:, 0] > pct(rel[:, 3], .1)).sum().item()
    print(f"workgroups that start after the first 10 % have finished (a second round): {late}")


if __name__ == "__main__":
    main()
