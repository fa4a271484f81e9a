#!/usr/bin/env python3
"""Launch the HBM-bound kernels a few times each on distinct buffers (for rocprofv3 --pmc runs):
GEMV (M=1) on the three Llama shapes, fake-quant fwd / bwd (bf16) on the three shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import packing  # noqa: E402
from mxq_amd.utils_quant import mx_fake_quant, ste_clip_backward  # noqa: E402

dev = torch.device("cuda:0")
for N, K in [(4096, 4096), (11008, 4096), (4096, 11008)]:
    g = torch.Generator(device=dev).manual_seed(N + K)
    ps = [packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half()) for _ in range(6)]
    x = torch.randn(1, K, generator=g, device=dev).half()
    for p in ps:
        packing.linear(x, p, path="gemv")
    ws = [(torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16() for _ in range(6)]
    go = torch.randn(N, K, generator=g, device=dev).bfloat16()
    for w in ws:
        mx_fake_quant(w, 2)
    for w in ws:
        ste_clip_backward(go, w, -2.0, 2.0)
torch.cuda.synchronize()
