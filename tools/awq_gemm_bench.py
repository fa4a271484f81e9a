#!/usr/bin/env python3
"""Timing of gemm_forward_cuda (the reference GEMM's operand format, csrc/gemm_awq.hip) on Llama shapes, with torch's fp16
GEMM on the dequantised weight beside it.   python tools/awq_gemm_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxq_inference_engine as eng  # noqa: E402

dev = torch.device("cuda:0")


def us(fn, calls=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(calls):
            fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / calls * 1e3)
    return best


for IC, OC in ((4096, 4096), (4096, 11008), (11008, 4096)):
    G = 128
    kern = torch.randint(-2**31, 2**31 - 1, (IC, OC // 8), dtype=torch.int32, device=dev)
    zeros = torch.randint(-2**31, 2**31 - 1, (IC // G, OC // 8), dtype=torch.int32, device=dev)
    scales = (torch.rand(IC // G, OC, device=dev) * 0.004 + 0.001).half()
    wd = torch.randn(OC, IC, device=dev).half()
    for M, S in ((16, 8), (64, 8), (128, 4), (256, 2), (512, 1), (1024, 1), (2048, 1)):
        x = torch.randn(M, IC, device=dev).half()
        t = us(lambda: eng.gemm_forward_cuda(x, kern, scales, zeros, S))
        tt = us(lambda: torch.matmul(x, wd.t()))
        fl = 2.0 * M * IC * OC
        print(f"M={M:5d} IC={IC} OC={OC} split_k={S}: {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s   (torch fp16 GEMM {tt:8.1f} us)", flush=True)
