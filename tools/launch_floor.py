#!/usr/bin/env python3
"""What a dependent decode launch costs when it moves (next to) no bytes: hipGraph chains of the fused one-token GEMV on tiny
weights, by prologue / residual / row count, next to tools/probes/edge_probe.hip's empty-kernel chain (1.72 us).
    python tools/launch_floor.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import packing  # noqa: E402

dev = torch.device("cuda:0")
L = 160


def chain_us(fn, h0):
    def run():
        h = h0
        for _ in range(L):
            h = fn(h)
        return h
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / L * 1e3)
    return best


gen = torch.Generator(device=dev).manual_seed(1)
for K in (64, 256, 4096):
    for N in (4096,):
        p = packing.quantize_pack((torch.randn(N, K, generator=gen, device=dev) * 0.02).half())
        nw = torch.ones(K, device=dev, dtype=torch.float16)
        h0 = torch.randn(1, max(N, K), generator=gen, device=dev).half()
        pad = lambda y: y if y.shape[1] >= K else torch.cat([y, y.new_zeros(1, K - y.shape[1])], 1)
        for pro, res in ((0, False), (0, True), (1, False), (1, True)):
            def step(h, pro=pro, res=res):
                x = h[:, :K]
                y = packing.linear_fused(x, p, pro, nw if pro == 1 else None, residual=h[:, :N] if res else None)
                return y
            t = chain_us(step, h0[:, :max(N, K)].contiguous())
            print(f"N={N:5d} K={K:5d} ({p.nbytes() / 1e3:8.1f} KB) prologue={pro} residual={int(res)}: {t:.2f} us per dependent launch", flush=True)

# where does the RMSNorm prologue's +0.75 us come from?  the same launch with its norm weight taken from (a) a long-lived tensor
# (as above), (b) the previous launch's OUTPUT buffer (written a few microseconds ago, like x itself)
K, N = 64, 4096
p = packing.quantize_pack((torch.randn(N, K, generator=gen, device=dev) * 0.02).half())
nw_static = torch.ones(K, device=dev, dtype=torch.float16)
h0 = torch.randn(1, N, generator=gen, device=dev).half()
for name, pick in (("long-lived norm weight", lambda h: nw_static), ("norm weight = a slice of the previous launch's output", lambda h: h[0, 1024:1024 + K])):
    t = chain_us(lambda h: packing.linear_fused(h[:, :K], p, 1, pick(h)), h0)
    print(f"N={N} K={K} prologue=1, {name}: {t:.2f} us per dependent launch", flush=True)
