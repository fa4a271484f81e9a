"""``Quantizer`` with the reference's API (mxq_quant/lib/quantizer.py:23-180) for the
configuration MXQ uses: ``configure(bits, perchannel=True, sym=False, qq_scale_bits=4)``.

In the reference a ``Quantizer`` is created per 16-column group inside a Python loop
(mxqgpt.py:417-428, ~25 torch kernels per call, 3*K/64 + 1 calls per Linear).  Here the
object is a *view* over the result of the fused HIP quantise-and-pack kernel: it is
obtained from ``MXQGPT`` (``mxq.quantizer(chunk, group)`` / ``mxq.quantizer_4b``) and exposes
the same attributes -- ``scale``, ``zero``, ``quant_scale``, ``qq_scale.scale/zero``, ``maxq`` --
and ``quantize`` / ``dequantize`` / ``quantize_dequantize`` on the group it belongs to.
All tensors come from the packed buffer via the unpack kernel; no quantisation arithmetic
is done in Python.
"""
from __future__ import annotations

import torch


class _QQ:
    def __init__(self, scale, zero):
        self.scale, self.zero = scale, zero


class Quantizer:
    def __init__(self, bits, codes, scale_code, zero, qs, qz, dequant_cols):
        self.bits = bits
        self.maxq = torch.tensor(2 ** bits - 1)
        self.perchannel, self.sym, self.round_zero = True, False, False
        self.qq_scale_bits, self.qq_groupsize = 4, 16
        self._codes = codes                      # uint8 [rows, cols]
        self.quant_scale = scale_code.reshape(-1, 16).float()     # [rows/16, 16] like the reference
        self.qq_scale = _QQ(qs.reshape(-1, 1), qz.reshape(-1, 1))
        rep = lambda t: t.reshape(-1, 1).repeat_interleave(16, dim=0)
        # quantizer.py:121 -- scale = qs * (code - qz); two fp32 roundings, same as the kernels
        self.scale = (rep(qs) * (scale_code.reshape(-1, 1).float() - rep(qz)))
        self.zero = zero.reshape(-1, 1)
        self._deq = dequant_cols                 # fp16 [rows, cols] from the dequant kernel

    def quantize(self, x=None):
        """Integer codes of the group (float tensor, like the reference's return)."""
        return self._codes.float()

    def dequantize(self, q=None):
        return self._deq.float()

    def quantize_dequantize(self, x=None):
        return self._deq.float()

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return torch.all(self.scale != 0)
