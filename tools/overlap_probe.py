#!/usr/bin/env python3
"""Probe: the dequant HOISTED OUT of the GEMM and run one Linear AHEAD on a second stream (dequant of Linear i + 1 into a second
fp16 scratch while the dense MFMA kernel multiplies Linear i), against the fused kernel, at the headline's shapes (2048 tokens).
    python tools/overlap_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import _lib, packing  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
M = 2048
g = torch.Generator(device=dev).manual_seed(0)
for N, K in ((4096, 4096), (11008, 4096), (4096, 11008)):
    ps = [packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half()) for _ in range(4)]
    x = torch.randn(M, K, generator=g, device=dev).half()
    y = torch.empty(M, N, device=dev, dtype=torch.float16)
    scratch = [torch.empty(N, K, device=dev, dtype=torch.float16) for _ in range(2)]
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    L = 40

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / L * 1e3)
        return best

    def deq(p, out, stream):
        _lib.check(lib.mxq_dequant_f16(p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), p.N, p.K, stream.cuda_stream), "dequant")

    def fused():
        for i in range(L):
            packing.linear(x, ps[i % 4], out=y, path="gemm")

    def dense_only():
        for i in range(L):
            packing.linear_dense(x, scratch[i % 2], out=y, variant="dense128")

    def serial():
        for i in range(L):
            deq(ps[i % 4], scratch[0], main)
            packing.linear_dense(x, scratch[0], out=y, variant="dense128")

    def overlapped():
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        free = [torch.cuda.Event(), torch.cuda.Event()]
        deq(ps[0], scratch[0], main)
        ready[0].record(main)
        for i in range(L):
            s, n = i % 2, (i + 1) % 2
            # side stream: next Linear's weight into the other scratch, once the GEMM that last read it is done
            if i >= 1:
                side.wait_event(free[n])
            deq(ps[(i + 1) % 4], scratch[n], side)
            ready[n].record(side)
            main.wait_event(ready[s])
            packing.linear_dense(x, scratch[s], out=y, variant="dense128")
            free[s].record(main)
        main.wait_stream(side)

    print(f"{M} x {N} x {K}: fused {timed(fused):.1f} us | dense kernel alone {timed(dense_only):.1f} | dequant pass + dense, one stream "
          f"{timed(serial):.1f} | dequant one Linear ahead on a second stream {timed(overlapped):.1f}", flush=True)
