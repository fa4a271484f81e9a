#!/usr/bin/env python3
"""The reference's kernel benchmark (mxq_quant/cuda_kernel/test_mxq_gemv.py:23-82) on this build's drop-in
``mxq_inference_engine``: fp16 ``torch.matmul`` vs the AWQ-format 4-bit GEMV (``gemv_forward_cuda``, group 128) vs the
MXQ "2.8-bit" prototype-format GEMV (``gemv_mxq_forward_cuda``), M = 1, N = K = 4096, same operand shapes and the same
measurement (a host-synchronised loop, ms per call and speed-up over fp16).  Next to it, the GPU-side time per launch
under hipGraph replay over distinct operand copies (HBM-cold), and the native v1-format GEMV on the same shape.

    python tools/compat_gemv_bench.py [--count 2000] > profiles/r02_compat_gemv_bench.txt
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mxq_inference_engine  # noqa: E402
from mxq_amd import packing  # noqa: E402


def host_loop(fn, count):
    fn()
    torch.cuda.synchronize()
    tick = time.time()
    for _ in range(count):
        fn()
        torch.cuda.synchronize()
    return (time.time() - tick) / count


def graph_time(make_call, n_copies):
    calls = [make_call(i) for i in range(n_copies)]
    calls[0]()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for c in calls:
            c()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n_copies * 1e3)
    return sorted(ts)[3]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=2000)     # the reference loops 100000 times
    args = ap.parse_args()
    DEV = torch.device("cuda")
    M, N, K = 1, 4096, 4096
    DTYPE = torch.half
    print(f"# reference harness cuda_kernel/test_mxq_gemv.py on MI355X: M={M} N={N} K={K}, host-synchronised loop of {args.count}")

    B = torch.randn((K, N), device=DEV, dtype=DTYPE)
    A = torch.randn((M, K), device=DEV, dtype=DTYPE)
    C = torch.zeros((M, N), device=DEV, dtype=DTYPE)
    t_fp16 = host_loop(lambda: torch.matmul(A, B, out=C), args.count)
    print(f"FP16: {t_fp16 * 1000:.5f} ms")

    group_size, pack_num = 128, 8
    Bq = torch.randint(-1000000000, 1000000000, (N, K // pack_num), device=DEV, dtype=torch.int)
    scales = torch.randn((N, K // group_size), device=DEV, dtype=DTYPE)
    zeros = torch.ones((N, K // group_size // pack_num), device=DEV, dtype=torch.int)
    t_awq = host_loop(lambda: mxq_inference_engine.gemv_forward_cuda(A, Bq, scales, zeros, group_size), args.count)
    print(f"awq_4bit: {t_awq * 1000:.5f} ms\nspeedup with fp16 {t_fp16 / t_awq:.4f}X")

    group_size, groupsize_2nd, pack_num = 16, 4, 16
    Bm = torch.randint(-1000000000, 1000000000, (N, K // pack_num), device=DEV, dtype=torch.int)
    scales_1nd = torch.ones((N, K // group_size // pack_num * 2), device=DEV, dtype=torch.int)
    scales_2nd = torch.randn((N // groupsize_2nd, K // group_size), device=DEV, dtype=DTYPE)
    zeros_2nd = torch.ones((N // groupsize_2nd, K // group_size // pack_num * 2), device=DEV, dtype=torch.int)
    scales_4b = torch.randn(N, device=DEV, dtype=DTYPE)
    zeros_4b = torch.ones(N // 8, device=DEV, dtype=torch.int)
    t_mxq = host_loop(lambda: mxq_inference_engine.gemv_mxq_forward_cuda(A, Bm, Bm, scales_1nd, scales_2nd, zeros_2nd,
                                                                        scales_4b, zeros_4b, group_size), args.count)
    print(f"mxq_2.8bit: {t_mxq * 1000:.5f} ms\nspeedup with fp16 {t_fp16 / t_mxq:.4f}X")

    # GPU-side time per launch, operands rotated through enough copies to stream from HBM
    print("\n# GPU-side time per launch (hipGraph replay over distinct operand copies, >= 600 MB per graph)")
    n16 = 20
    Bs = [B] + [B.clone() for _ in range(n16 - 1)]
    t = graph_time(lambda i: (lambda: torch.matmul(A, Bs[i], out=C)), n16)
    print(f"fp16 torch.matmul         {t:7.2f} us   {N * K * 2 / t / 1e6:.2f} TB/s of weight bytes")
    nq = 72
    Bqs = [Bq] + [Bq.clone() for _ in range(nq - 1)]
    t = graph_time(lambda i: (lambda: mxq_inference_engine.gemv_forward_cuda(A, Bqs[i], scales, zeros, 128)), nq)
    by = Bq.numel() * 4 + scales.numel() * 2 + zeros.numel() * 4
    print(f"gemv_forward_cuda (awq4)  {t:7.2f} us   {by / t / 1e6:.2f} TB/s of operand bytes ({by / 1e6:.1f} MB)")
    Bms = [Bm] + [Bm.clone() for _ in range(nq - 1)]
    t = graph_time(lambda i: (lambda: mxq_inference_engine.gemv_mxq_forward_cuda(
        A, Bms[i], Bms[i], scales_1nd, scales_2nd, zeros_2nd, scales_4b, zeros_4b, 16)), nq)
    by = N * K // 16 * 4 + N * K // 64 * 4 + scales_1nd.numel() * 4 + (N // 4) * 192 * 2 + (N // 4) * 32 * 4 + N * 2 + N // 8 * 4
    print(f"gemv_mxq_forward_cuda     {t:7.2f} us   {by / t / 1e6:.2f} TB/s of operand bytes ({by / 1e6:.1f} MB)")
    g = torch.Generator(device=DEV).manual_seed(0)
    p = packing.quantize_pack((torch.randn(N, K, generator=g, device=DEV) * 0.02).half())
    ps = [p] + [packing.PackedMXQ(p.qweight.clone(), p.rowmeta.clone(), N, K) for _ in range(nq - 1)]
    out = torch.empty(M, N, device=DEV, dtype=DTYPE)
    t = graph_time(lambda i: (lambda: packing.linear(A, ps[i], out=out, path="gemv")), nq)
    print(f"native mxq_gemv_f16 (v1)  {t:7.2f} us   {p.nbytes() / t / 1e6:.2f} TB/s of packed bytes ({p.nbytes() / 1e6:.1f} MB, exact metadata)")


if __name__ == "__main__":
    main()
