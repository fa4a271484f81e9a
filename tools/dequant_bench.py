#!/usr/bin/env python3
"""Time of the hoisted mode's first half, the dequant kernel (packed weight -> dense fp16 [N, K]), GPU box.
hipGraph replay over distinct packed weights and distinct outputs (nothing cache-resident between launches);
bytes = packed weight read + fp16 matrix written.   python tools/dequant_bench.py [--lib mxq_amd/libmxq_hip.so,...]"""
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
    ap.add_argument("--libs", default="mxq_amd/libmxq_hip.so")
    ap.add_argument("--compact", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for N, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
        g = torch.Generator(device=dev).manual_seed(N + K)
        base = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half(), compact_meta=args.compact)
        nw = 8
        qs = [base.qweight.clone() for _ in range(nw)]
        outs = [torch.empty(N, K, dtype=torch.float16, device=dev) for _ in range(nw)]
        ref = packing.dequant(base)
        nbytes = base.nbytes() + N * K * 2
        for name in args.libs.split(","):
            lib = ctypes.CDLL(os.path.join(ROOT, name))
            fn = lib.mxq_dequant_f16_compact if args.compact else lib.mxq_dequant_f16
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

            def run():
                st = torch.cuda.current_stream().cuda_stream
                for q, o in zip(qs, outs):
                    rc = fn(q.data_ptr(), base.rowmeta.data_ptr(), o.data_ptr(), N, K, st)
                    assert rc == 0, rc
            run()
            torch.cuda.synchronize()
            assert all(torch.equal(o, ref) for o in outs), f"{name}: output differs"
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                run()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                gr.replay()
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / nw * 1e3)
            t = sorted(ts)[len(ts) // 2]
            print(f"[{N:5d}, {K:5d}] {name:36s}: {t:6.1f} us per weight  {nbytes / t / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
