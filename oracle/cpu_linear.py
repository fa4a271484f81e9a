"""CPU baseline leg for bench.py -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

"The reference's CPU dequant + matmul": the reference has no CPU inference path of its
own; config 1 of BASELINE.json is its Python arithmetic on host cores, i.e.
Quantizer.dequantize = scale * (q - zero) in fp32 (mxq_quant/lib/quantizer.py:19-20) over
the MXQ layout of MXQGPT.fasterquant (mxq_quant/lib/mxqgpt.py:404-443), cast to fp16
(:448), then nn.functional.linear.  Restated with torch CPU ops (what the reference would
run); checked against the numpy oracle in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np
import torch


def to_torch(params):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()
            if isinstance(v, np.ndarray) and k != "w_deq32"}


def dequant_torch(p, N, K):
    """codes + params (torch CPU tensors) -> fp16 [N, K] fake-quant weight."""
    nc = K // 64
    s2 = p["qs2"].repeat_interleave(16, 0) * (p["sc2"].float() - p["qz2"].repeat_interleave(16, 0))
    s4 = p["qs4"].repeat_interleave(16, 0) * (p["sc4"].float() - p["qz4"].repeat_interleave(16, 0))
    w2 = s2[:, :, None] * (p["codes2"].reshape(N, nc * 3, 16).float() - p["zero2"][:, :, None])
    w4 = s4[:, None] * (p["codes4"].float() - p["zero4"][:, None])
    W = torch.empty(N, nc, 64)
    W[:, :, :48] = w2.reshape(N, nc, 48)
    W[:, :, 48:] = w4.reshape(N, nc, 16)
    return W.reshape(N, K).half()


def dequant_linear(p, N, K, x16):
    """One call of the CPU path: dequantise, then y = x . W'^T in fp32."""
    w16 = dequant_torch(p, N, K)
    return torch.nn.functional.linear(x16.float(), w16.float())
