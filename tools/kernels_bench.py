#!/usr/bin/env python3
"""Timing of the HBM-bound kernels on Llama-2-7B shapes (GPU box): decode GEMV (config 3's inner
op), MXAsymQuantizer fwd / STE bwd (config 4's inner loop), quantise-and-pack.  Reports achieved
algorithmic GB/s against the 8 TB/s spec / 6.3 TB/s achievable HBM rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import packing  # noqa: E402
from mxq_amd.utils_quant import mx_fake_quant, ste_clip_backward  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(4096, 4096), (11008, 4096), (4096, 11008)]


def timeit(fn, iters=50, warm=5, flush=None):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(5):
        if flush is not None:
            flush.zero_()       # evict L2 / Infinity Cache between rounds (cold-ish weights)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    print("== decode GEMV (M tokens, packed weights streamed once) ==")
    for N, K in SHAPES:
        g = torch.Generator(device=dev).manual_seed(N + K)
        # several distinct weights so that consecutive calls do not hit the 256 MiB Infinity Cache
        ps = [packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half()) for _ in range(24)]
        wd = packing.dequant(ps[0])
        for M in (1, 2, 4):
            x = torch.randn(M, K, generator=g, device=dev).half()
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            y = packing.linear(x, ps[0], path="gemv").float()
            yref = x.float() @ wd.float().t()
            err = ((y - yref).abs().max() / yref.abs().max()).item()
            # replay 24 calls (24 distinct weights) from a hipGraph: per-call host overhead of the
            # Python/ctypes launch (~11 us) would otherwise hide the kernel time
            for p_ in ps:
                packing.linear(x, p_, out=out, path="gemv")
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for p_ in ps:
                    packing.linear(x, p_, out=out, path="gemv")
            us = timeit(graph.replay, iters=4) / len(ps)
            nbytes = ps[0].nbytes() + 2 * M * K + 2 * M * N
            print(f"  M={M} N={N} K={K}: {us:7.2f} us  {nbytes/us/1e3:7.1f} GB/s ({nbytes/us/1e3/8000*100:4.1f}% of 8 TB/s) "
                  f"err {err:.1e}", flush=True)
        ws = [(torch.randn(N, K, generator=g, device=dev) * 0.02).half() for _ in range(8)]
        x = torch.randn(1, K, generator=g, device=dev).half()
        for w_ in ws:
            torch.matmul(x, w_.t())
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for w_ in ws:
                torch.matmul(x, w_.t())
        us = timeit(graph.replay, iters=4) / len(ws)
        print(f"  (torch fp16 GEMV M=1: {us:7.2f} us  {2*N*K/us/1e3:7.1f} GB/s)")
        del ps, ws

    print("== MXAsymQuantizer fwd / STE bwd, bf16 ==")
    tot_f = tot_b = 0.0
    for N, K in SHAPES:
        g = torch.Generator(device=dev).manual_seed(N * 3 + K)
        w = (torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16()
        go = torch.randn(N, K, generator=g, device=dev).bfloat16()
        uf = timeit(lambda: mx_fake_quant(w, 2), iters=20)
        ub = timeit(lambda: ste_clip_backward(go, w, -2.0, 2.0), iters=20)
        n = N * K
        print(f"  [{N},{K}]: fwd {uf:7.1f} us {4*n/uf/1e3:7.1f} GB/s | bwd {ub:7.1f} us {6*n/ub/1e3:7.1f} GB/s", flush=True)
        mult = {(4096, 4096): 4, (11008, 4096): 2, (4096, 11008): 1}[(N, K)]
        tot_f += mult * uf
        tot_b += mult * ub
    n_blk = 202_375_168
    print(f"  one decoder block (7 weights): fwd {tot_f:.1f} us = {4*n_blk/tot_f/1e3:.0f} GB/s, "
          f"bwd {tot_b:.1f} us = {6*n_blk/tot_b/1e3:.0f} GB/s  (floors at 6.3 TB/s: 128 / 193 us)")

    print("== SymQuantizer / AsymQuantizer forward, bf16 (activation [2, 2048, 4096] per token; KV [2, 32, 2048, 128] per head) ==")
    from mxq_amd.utils_quant import act_fake_quant
    for shape, what in (((2, 2048, 4096), "activation"), ((2, 32, 2048, 128), "kv"), ((4096, 11008), "2-D groups")):
        xa = (torch.randn(*shape, device=dev) * 1.5).bfloat16()
        n = xa.numel()
        for sym, bits in ((True, 16), (False, 8)):
            us = timeit(lambda: act_fake_quant(xa, bits, False, sym), iters=20)
            print(f"  {what:<11} {'sym' if sym else 'asym'} b{bits}: {us:7.1f} us  {4*n/us/1e3:7.1f} GB/s algorithmic (read + write)", flush=True)

    print("== quantise-and-pack (fp16 in) ==")
    for N, K in SHAPES:
        W = (torch.randn(N, K, device=dev) * 0.02).half()
        us = timeit(lambda: packing.quantize_pack(W), iters=10)
        print(f"  [{N},{K}]: {us:7.1f} us  {(2*N*K + 0.5625*N*K)/us/1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
