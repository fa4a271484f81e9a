#!/usr/bin/env python3
"""Where a K-step of the gemm8 prefill kernel spends its cycles on the MFMA waves (diagnostic build with s_memtime
stamps in libmxq_hip_prof.so; the stamps perturb the run: read the SHARES, not the length).  The dequant waves'
stamps went away with their per-step structure (they now work in bursts of three chunks); their rows in
profiles/r02_*stamps* are from the per-step kernel earlier in round 2.
    python tools/gemm_stamps.py [--m 2048] [--n 4096] [--k 4096] [--abl 0,2048,4,2]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import packing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--abl", default="0,2048,1024,4,2")
    ap.add_argument("--lib", default="mxq_amd/libmxq_hip_prof.so")
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--per-wave", action="store_true")
    ap.add_argument("--by-xcd", action="store_true")
    ap.add_argument("--half", action="store_true", help="the 128-token tile build of the kernel (gemm8h: 4 MFMA waves)")
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(ROOT, args.lib))
    fn = lib.mxq_prof_gemm8h_stamps_f16 if args.half else lib.mxq_prof_gemm8_stamps_f16
    bm, nm = (128, 4) if args.half else (256, 8)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    M, N, K = args.m, args.n, args.k
    g = torch.Generator(device=dev).manual_seed(1)
    p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
    x = torch.randn(M, K, generator=g, device=dev).half()
    y = torch.empty(M, N, device=dev, dtype=torch.float16)
    grid = ((M + bm - 1) // bm) * ((N + 127) // 128)
    for abl in [int(a) for a in args.abl.split(",")]:
        dbg = torch.zeros(grid * 12 * 4, dtype=torch.int64, device=dev)   # 8 + 4 or 4 + 8 waves
        for _ in range(args.launches):   # sustained load; the last launch's sums are read
            rc = fn(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), y.data_ptr(), M, N, K, abl, dbg.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        raw = dbg.view(grid, 12, 4).cpu()
        rt = (raw[:, :nm, 3] >> 32).double()                     # 100 MHz ticks over the stamped K-steps
        d = raw.double()
        d[:, :, 3] = (raw[:, :, 3] & 0xFFFFFFFF).double()
        steps = d[:, :, 3].clamp(min=1)
        per = d[:, :, :3] / steps[:, :, None]            # cycles per K-step
        for name, sl in ((f"MFMA waves 0-{nm - 1}", slice(0, nm)),):
            w = per[:, sl, :].mean(dim=(0, 1))
            ghz = (d[:, sl, :3].sum(dim=2) / rt.clamp(min=1) / 10.0).mean().item()
            print(f"abl {abl:5d} {name:28s} work {w[0]:7.0f}  wait {w[1]:6.0f}  barrier {w[2]:6.0f}  total {w.sum():7.0f} cycles/step"
                  f"   core clock held {ghz:.2f} GHz", flush=True)
        if args.by_xcd:        # clock held and cycles per step by XCD label (workgroup id & 7; whole-tile launches: id = tile)
            ids = torch.arange(grid)
            for e in range(8):
                m = (ids & 7) == e
                w = per[m][:, :nm, :].mean(dim=(0, 1))
                g = (d[m][:, :nm, :3].sum(dim=2) / rt[m].clamp(min=1) / 10.0).mean().item()
                print(f"          XCD label {e}: {w.sum():7.0f} cycles/step  clock {g:.3f} GHz  -> {w.sum() / g / 1e3:6.3f} us/step", flush=True)
        if args.per_wave:      # the two MFMA waves of a SIMD (w and w + 4) are not served alike: the older one goes first
            for wv in range(nm):
                w = per[:, wv, :].mean(dim=0)
                print(f"          MFMA wave {wv} (SIMD {wv % 4})       work {w[0]:7.0f}  wait {w[1]:6.0f}  barrier {w[2]:6.0f}", flush=True)


if __name__ == "__main__":
    main()
