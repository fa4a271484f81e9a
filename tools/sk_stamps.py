#!/usr/bin/env python3
"""Where the stream-K fix-up of the fused prefill kernel spends its time (GPU box; needs `make -C mxq_amd/csrc prof`).

    python tools/sk_stamps.py --m 512 --shapes 4096x4096[,11008x4096] [--dist3]

libmxq_hip_prof.so's mxq_prof_gemm8_skstamps_f16 runs the product kernel (tail always split) with wall-clock stamps
(s_memrealtime, 10 ns) taken by MFMA wave 0 of every workgroup around the phases of its LAST piece:
  0 kernel start | 1 K loop of the last piece done | owner: 7 spin starts, 2 the contributors' counts are in, 3 their slots
  are added | parker: 2 slot stores issued, 3 stores retired + count bumped | all-contributors mode: 4 first task's counts are
  in, 5 tasks done | 6 everything this workgroup stored has retired.
Prints medians (and p90 / max) over the workgroups of each role, in us relative to stamp 1 unless said otherwise."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else float("nan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--shapes", default="4096x4096")
    ap.add_argument("--dist3", action="store_true", help="all-contributors reduction from 3 contributors per tile on")
    ap.add_argument("--half", action="store_true", help="the 128-token build (gemm8h)")
    ap.add_argument("--by-xcd", action="store_true")
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(ROOT, "mxq_amd/libmxq_hip_prof.so"))
    fn = lib.mxq_prof_gemm8h_skstamps_f16 if args.half else lib.mxq_prof_gemm8_skstamps_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    M = args.m
    for N, K in [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]:
        g = torch.Generator(device=dev).manual_seed(N + K)
        p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
        x = torch.randn(M, K, generator=g, device=dev).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        yref = x.float() @ packing.dequant(p).float().t()
        ws = packing.gemm_workspace(dev)
        head = ws[:65536].view(torch.int64)           # stamps: u64 [workgroup][8] from byte 32768
        st = torch.cuda.current_stream().cuda_stream

        def call():
            rc = fn(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, N, K, int(args.dist3),
                    ws.data_ptr(), ws.numel(), st)
            assert rc == 0, rc
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        err = ((out.float() - yref).abs().max() / yref.abs().max()).item()
        assert err < 1e-3, err
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        head[4096:].zero_()
        torch.cuda.synchronize()
        call()
        torch.cuda.synchronize()
        s = head[4096:4096 + 256 * 8].cpu().numpy().reshape(256, 8).astype(np.float64) / 100.0   # us
        live = s[:, 0] > 0
        s = s[live]
        t0 = s[:, 0].min()
        print(f"M={M} N={N} K={K} {'128' if args.half else '256'}-token tile dist3={int(args.dist3)}: {us:.1f} us per launch (stream-ordered, stamps on), "
              f"{int(live.sum())} workgroups, max rel err {err:.1e}")
        split = s[:, 1] > 0
        print(f"  workgroups with a split piece: {int(split.sum())}; kernel start spread {pct(s[:, 0] - t0, 50):.2f} / "
              f"{(s[:, 0] - t0).max():.2f} us (median / max)")
        print(f"  last piece's K loop done at {pct(s[split, 1] - t0, 50):.2f} us after the first start (p10 "
              f"{pct(s[split, 1] - t0, 10):.2f}, p90 {pct(s[split, 1] - t0, 90):.2f}, max {(s[split, 1] - t0).max():.2f}); "
              f"last workgroup done at {(s[:, 6] - t0).max():.2f}")
        if args.by_xcd:      # where the K loops end, by XCD label (workgroup id & 7) and by unit index (id >> 3)
            ids = np.nonzero(live)[0]
            d1 = s[:, 1] - t0
            for e in range(8):
                m = split & ((ids & 7) == e)
                if m.any():
                    print(f"    XCD label {e}: loop end median {pct(d1[m], 50):7.2f}  min {d1[m].min():7.2f}  max {d1[m].max():7.2f}  (n={int(m.sum())})")
            order = np.argsort(d1)
            print("    latest 12 workgroups (id: loop end, done):", ", ".join(f"{int(ids[i])}: {d1[i]:.1f}, {s[i, 6] - t0:.1f}" for i in order[-12:]))
            print("    earliest 6:", ", ".join(f"{int(ids[i])}: {d1[i]:.1f}" for i in order[:6] if split[i]))
        own = split & (s[:, 7] > 0)
        park = split & ~own
        def row(name, a):
            print(f"    {name:<58s} median {pct(a, 50):6.2f}  p90 {pct(a, 90):6.2f}  max {a.max() if len(a) else float('nan'):6.2f}")
        if own.any():
            print(f"  owners ({int(own.sum())}):")
            row("wait for the contributors' counts (1 -> 2)", s[own, 2] - s[own, 1])
            row("read + add their slots (2 -> 3)", s[own, 3] - s[own, 2])
            row("output retired (3 -> 6)", s[own, 6] - s[own, 3])
            row("whole fix-up (1 -> 6)", s[own, 6] - s[own, 1])
        if park.any():
            d = s[park]
            print(f"  parkers ({int(park.sum())}):")
            row("slot stores issued (1 -> 2)", d[:, 2] - d[:, 1])
            row("stores retired, count bumped (2 -> 3)", d[:, 3] - d[:, 2])
            has4 = d[:, 4] > 0
            if has4.any():
                row("all-contributors: counts in (3 -> 4)", d[has4, 4] - d[has4, 3])
                row("all-contributors: slots summed, block written (4 -> 5)", d[has4, 5] - d[has4, 4])
                row("all-contributors: output retired (5 -> 6)", d[has4, 6] - d[has4, 5])
            row("whole fix-up (1 -> 6)", d[:, 6] - d[:, 1])


if __name__ == "__main__":
    main()
