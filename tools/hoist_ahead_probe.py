#!/usr/bin/env python3
"""Feasibility probe: the hoisted mode with the NEXT weight's dequant pass running on a side stream under the current
weight's dense GEMM (two scratch slots), against the fused kernel and the plain hoisted mode, on a sequence of distinct
Llama-shaped weights at M tokens (GPU box).   python tools/hoist_ahead_probe.py [--m 2048] [--layers 4]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import _lib, packing  # noqa: E402
from mxq_amd import llama_shapes as LS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    M = args.m
    H, I = LS.HIDDEN, LS.INTERMEDIATE
    shapes = ([(H, H)] * 4 + [(I, H)] * 2 + [(H, I)]) * args.layers
    g = torch.Generator(device=dev).manual_seed(0)
    base = {s: packing.quantize_pack((torch.randn(*s, generator=g, device=dev) * 0.02).half()) for s in set(shapes)}
    ws = [packing.PackedMXQ(base[s].qweight.clone(), base[s].rowmeta, s[0], s[1]) for s in shapes]
    xs = {K: torch.randn(M, K, generator=g, device=dev).half() for K in (H, I)}
    ys = {N: torch.empty(M, N, device=dev, dtype=torch.float16) for N in (H, I)}
    lib = _lib.load()
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    slots = [torch.empty(I * H, dtype=torch.float16, device=dev) for _ in range(2)]
    flops = sum(2.0 * M * n * k for n, k in shapes)

    def fused():
        for p in ws:
            packing.linear(xs[p.K], p, out=ys[p.N], path="gemm")

    def hoisted():
        for p in ws:
            packing.linear(xs[p.K], p, out=ys[p.N], path="hoist")

    def deq(p, slot, stream):
        _lib.check(lib.mxq_dequant_f16(p.qweight.data_ptr(), p.rowmeta.data_ptr(), slot.data_ptr(), p.N, p.K, stream.cuda_stream), "deq")

    def ahead():
        ready = [torch.cuda.Event() for _ in ws]
        done = [torch.cuda.Event() for _ in ws]
        side.wait_stream(main_s)
        deq(ws[0], slots[0], side)
        ready[0].record(side)
        for i, p in enumerate(ws):
            if i + 1 < len(ws):
                if i >= 1:
                    side.wait_event(done[i - 1])          # slot (i + 1) % 2 was read by GEMM i - 1
                deq(ws[i + 1], slots[(i + 1) % 2], side)
                ready[i + 1].record(side)
            main_s.wait_event(ready[i])
            w16 = slots[i % 2][:p.N * p.K].view(p.N, p.K)
            packing.linear_dense(xs[p.K], w16, out=ys[p.N])
            done[i].record(main_s)

    ref = None
    for name, fn in (("fused", fused), ("hoisted", hoisted), ("hoist-ahead", ahead)):
        fn()
        torch.cuda.synchronize()
        chk = ys[H].float().abs().sum().item()
        ref = chk if ref is None else ref
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2]
        print(f"M={M} {len(ws)} linears  {name:12s}: {t * 1e3 / len(ws):7.1f} us per linear  {flops / t / 1e9:7.1f} TFLOP/s  (checksum {chk:.6g}, ref {ref:.6g})", flush=True)


if __name__ == "__main__":
    main()
