#!/usr/bin/env python3
"""Time mxq_gemm_awq_f16 of several builds of the library in one process (same box, same inputs), hipGraph replay.
    python tools/awq_variant_bench.py [--ms 16,64] [--shapes 4096x4096,...] tools/_variants/lib_a.so tools/_variants/lib_b.so ..."""
import argparse
import ctypes

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="2048")
ap.add_argument("--shapes", default="4096x4096")          # ICxOC
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
dev = torch.device("cuda:0")
G = 128
ws = torch.zeros(80 << 20, dtype=torch.uint8, device=dev)
libs = []
for path in args.libs:
    lib = ctypes.CDLL(path)
    fn = lib.mxq_gemm_awq_f16
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    libs.append((path, fn))
for shape in args.shapes.split(","):
    IC, OC = (int(v) for v in shape.split("x"))
    kern = torch.randint(-2**31, 2**31 - 1, (IC, OC // 8), dtype=torch.int32, device=dev)
    zeros = torch.randint(-2**31, 2**31 - 1, (IC // G, OC // 8), dtype=torch.int32, device=dev)
    scales = (torch.rand(IC // G, OC, device=dev) * 0.004 + 0.001).half()
    for M in (int(v) for v in args.ms.split(",")):
        x = torch.randn(M, IC, device=dev).half()
        y = torch.empty(M, OC, device=dev, dtype=torch.float16)
        res = []
        for path, fn in libs:
            st = torch.cuda.current_stream().cuda_stream
            call = lambda: fn(x.data_ptr(), kern.data_ptr(), scales.data_ptr(), zeros.data_ptr(), y.data_ptr(), M, IC, OC, G, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                assert call() == 0
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(10):
                    call()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g.replay(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
            res.append(best)
        print(f"M={M:5d} IC={IC} OC={OC}: " + "  ".join(f"{r:7.1f} us" for r in res) + "   (" + " | ".join(p.split('/')[-1] for p, _ in libs) + ")", flush=True)
