#!/usr/bin/env python3
"""Run ONE GEMM kernel variant at ONE shape a few times (for rocprofv3 --pmc runs).
    python3 tools/gemm_prof.py gemm8 2048 4096 4096 [iters] [layout: mixed | w2g16 | w4row]
variant dense128 / dense256: the plain GEMM on the dequantised weight (mxq_dense_f16) with that kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import _lib, packing  # noqa: E402

if os.environ.get("MXQ_PROF_LIB"):      # a variant build of the library (tools/build_variant.sh), for counter runs of an A/B
    _lib.LIB_PATH = os.path.abspath(os.environ["MXQ_PROF_LIB"])

variant, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
layout = sys.argv[6] if len(sys.argv) > 6 else "mixed"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
p = packing.quantize_pack(W) if layout == "mixed" else packing.quantize_pack_uniform(W, layout)
x = torch.randn(M, K, generator=g, device=dev).half()
out = torch.empty(M, N, device=dev, dtype=torch.float16)
wd = packing.dequant(p) if variant in packing.DENSE_VARIANTS else None
for _ in range(iters):
    if wd is not None:
        packing.linear_dense(x, wd, out=out, variant=variant)
    elif layout == "mixed":
        packing.linear(x, p, out=out, path=variant)
    else:
        packing.linear_layout(x, p, out=out)
torch.cuda.synchronize()
