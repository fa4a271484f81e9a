"""``Quantizer`` with the reference's API (mxq_quant/lib/quantizer.py:23-180), for the configurations MXQ uses.

    q = Quantizer()
    q.configure(bits=2 | 4, perchannel=True, sym=False, qq_scale_bits=4)     # mxqgpt.py:417-423, :433-435
    q.find_params(W1, weight=True)        # W1 [rows, cols] on the GPU: one (scale, zero) per row
    codes = q.quantize(W1); w = q.dequantize(codes); w = q.quantize_dequantize(W1)

In the reference this object is created per 16-column group inside a Python loop (~25 torch kernels per
``find_params``, 3*K/64 + 1 objects per Linear).  Here ``find_params`` is ONE launch of the fused HIP
quantise-and-pack kernel on the given slice (``mxq_quantize_pack_layout``: per-row range, scale, un-rounded
zero-point, 4-bit second-order scale coding over 16 consecutive rows -- quantizer.py:81-121) followed by the
integer unpack; ``scale`` / ``zero`` / ``quant_scale`` / ``qq_scale.scale`` / ``qq_scale.zero`` / ``maxq`` then
hold exactly the reference's values.  ``quantize(x)`` / ``quantize_dequantize(x)`` on the tensor the parameters
were found on (held alive by the object, so a recycled address can never pass for it) return the kernel's codes /
dequantised values; on any OTHER tensor they apply the stored parameters with the reference's formula
(quantizer.py:5-20) as elementwise torch device ops -- GPU arithmetic, but not the HIP kernel.  GPU only, like
everything in this package; unsupported configurations raise instead of approximating.

``GroupView`` is the read-only view over one group of an already packed weight that ``MXQGPT.quantizer(chunk,
group)`` hands out for inspection (round 1 called it ``Quantizer``; it never had the reference's methods).
"""
from __future__ import annotations

import torch

from .. import packing


class _QQ:
    """The nested second-order quantiser's parameters (``Quantizer.qq_scale`` in the reference)."""

    def __init__(self, scale, zero):
        self.scale, self.zero = scale, zero
        self.maxq = torch.tensor(15)


class Quantizer:
    def __init__(self, shape=1):
        self.maxq = torch.tensor(0)
        self.scale = torch.zeros(shape)
        self.zero = torch.zeros(shape)
        self.bits = None
        self._src = None        # identity of the tensor find_params ran on (see _src_key)
        self._src_ref = None    # ... and the tensor itself: while it is held its memory cannot be recycled
        self._codes = None
        self._deq = None

    # -- reference signature (quantizer.py:30-59); only MXQ's own settings are implemented --------------
    def configure(self, bits, perchannel=False, sym=True, norm=2.0, grid=100, maxshrink=0.8, round_zero: bool = False,
                  qq_scale_bits=None, qq_zero_bits=None, qq_groupsize=16, qq_zero_sym=False, reserved_bins: int = 0,
                  qqq_params=None):
        if bits not in (2, 4):
            raise NotImplementedError("the HIP quantise kernels cover bits = 2 (one 16-column group) and bits = 4 (per row)")
        if not perchannel or sym or round_zero or reserved_bins:
            raise NotImplementedError("MXQ quantises per row, asymmetric, with an un-rounded zero-point "
                                      "(mxqgpt.py:419, :434); other settings are not on the hot path")
        if qq_scale_bits != 4 or qq_zero_bits is not None or qq_groupsize != 16 or qqq_params:
            raise NotImplementedError("second-order coding is fixed at 4-bit scales over 16 rows, zero-points uncoded "
                                      "(mxqgpt.py:419-421)")
        self.bits = bits
        self.maxq = torch.tensor(2 ** bits - 1)
        self.perchannel, self.sym, self.round_zero = perchannel, sym, round_zero
        self.norm, self.grid, self.maxshrink = norm, grid, maxshrink
        self.qq_scale_bits, self.qq_zero_bits, self.qq_groupsize, self.qq_zero_sym = 4, None, 16, qq_zero_sym
        self.qqq_params = {}

    def find_params(self, x, weight=False):
        if self.bits is None:
            raise RuntimeError("configure() first")
        if not weight or x.dim() != 2:
            raise NotImplementedError("find_params(x, weight=True) on a 2-D weight slice is the only form MXQ calls")
        if not x.is_cuda:
            raise ValueError("mxq_amd ops run on the GPU only (no CPU fallback): got a CPU tensor")
        rows, cols = x.shape
        if rows % 16 != 0:
            raise ValueError("second-order scale groups are 16 consecutive rows: rows % 16 must be 0 (quantizer.py:115)")
        if self.bits == 2:
            if cols != 16:
                raise NotImplementedError("a 2-bit Quantizer covers one 16-column group (mxqgpt.py:415-423)")
            xp = x.repeat(1, 4)                                     # the kernel's block is 64 columns = 4 independent groups
            layout, take = "w2g16", slice(0, 16)
        else:
            pad = (-cols) % 64                                      # a row's range does not change when its last value repeats
            xp = torch.cat([x, x[:, -1:].expand(rows, pad)], dim=1) if pad else x
            layout, take = "w4row", slice(0, cols)
        p = packing.quantize_pack_uniform(xp.contiguous(), layout)
        w16, got = packing.expand_uniform(p)
        sc = got["sc"][:, :1].float()                               # [rows, 1] scale codes (group 0 / the row)
        qs, qz = got["qs"][:, :1], got["qz"][:, :1]                 # [rows/16, 1]
        rep = lambda t: t.repeat_interleave(16, dim=0)
        # quantizer.py:121 -- scale = qs * (code - qz): the same two fp32 roundings as the kernels
        self.scale = rep(qs) * (sc - rep(qz))
        self.zero = got["zero"][:, :1].clone()
        self.quant_scale = sc.reshape(-1, 16)
        self.qq_scale = _QQ(qs.clone(), qz.clone())
        self.maxq = self.maxq.to(x.device)
        self._codes = got["codes"][:, take].float()
        self._deq = w16[:, take].float()                            # fp16(scale * (q - zero)): exact in fp32
        self._src = self._src_key(x)
        self._src_ref = x       # keeps the storage alive: a freed temporary's address would otherwise be handed to the
                                # next same-shape temporary (version 0, same dtype, other data) and match by accident

    @staticmethod
    def _src_key(x):
        return (x.untyped_storage().data_ptr(), x.storage_offset(), tuple(x.shape), tuple(x.stride()), x.dtype,
                x._version)

    def _is_source(self, x) -> bool:
        """True only for the very tensor (or an identical view of the very storage, unmodified since) that
        ``find_params`` ran on; that tensor is held in ``_src_ref``, so the comparison cannot alias a recycled block."""
        return self._src_ref is not None and self._src_key(x) == self._src

    def quantize(self, x):
        if not self.ready():
            return x
        if self._is_source(x):
            return self._codes
        return torch.clamp(torch.round(x / self.scale.clamp_min(1e-9) + self.zero), 0, self.maxq)

    def dequantize(self, q):
        if not self.ready():
            return q
        return self.scale * (q - self.zero)

    def quantize_dequantize(self, x):
        if not self.ready():
            return x
        return self.scale * (self.quantize(x) - self.zero)

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        ok = torch.all(self.scale != 0)
        assert ok                               # the reference asserts too (quantizer.py:178-180)
        return ok


class GroupView:
    """Read-only view over one quantisation group of a packed weight (``MXQGPT.quantizer`` / ``quantizer_4b``):
    the reference's loop-local ``Quantizer`` attributes -- ``scale``, ``zero``, ``quant_scale``,
    ``qq_scale.scale/zero``, ``maxq`` -- rebuilt from the integer unpack.  ``codes()`` / ``dequantized()`` return
    what the kernels produced for this group; there is no ``quantize(x)`` on new data here (use ``Quantizer``)."""

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

    def codes(self):
        """Integer codes of the group (float tensor, like the reference's ``quantize`` return)."""
        return self._codes.float()

    def dequantized(self):
        return self._deq.float()

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return torch.all(self.scale != 0)
