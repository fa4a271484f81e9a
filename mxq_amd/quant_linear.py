"""``QuantLinear`` -- the packed W2/4 x A16 replacement for ``nn.Linear``.

The reference only ever holds fake-quantised fp16 weights in a plain ``nn.Linear``
(mxq_quant/lib/mxqgpt.py:448, evaluated through mxq_quant/main.py:85) and has no module
around its CUDA prototype; this is the module the name
``QuantLinear`` in BASELINE.json maps to on the inference side.  ``nas_quant``
(mxq_quant/lib/prune.py:409-414) can swap it in after ``MXQGPT.fasterquant``.

state_dict: ``qweight`` int32 [N/16 * K/64 * 144] (format version 1, exact metadata) or [N/16 * K/64 * 120]
(version 2, compact metadata: fp16 zero-points, 3.75 bit/weight), ``rowmeta`` float32 [N, 4],
``fmt`` int32 [3] = (format version, N, K), optional ``bias`` float16 [N].
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import packing


class QuantLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool = False, device=None, compact: bool = False,
                 _buffers=None):
        super().__init__()
        packing.check_shape(out_features, in_features)
        self.in_features, self.out_features, self.compact = in_features, out_features, bool(compact)
        nq = packing.qweight_bytes(out_features, in_features) // 4 * (120 if compact else 144) // 144
        if _buffers is None:          # an empty module (to be filled by load_state_dict): zero-filled buffers of the right size
            qweight = torch.zeros(nq, dtype=torch.int32, device=device)
            rowmeta = torch.zeros((out_features, 4), dtype=torch.float32, device=device)
        else:                         # from_packed: adopt the packed tensors, nothing allocated and filled to be thrown away
            qweight, rowmeta = _buffers
            if qweight.numel() != nq or tuple(rowmeta.shape) != (out_features, 4):
                raise ValueError("packed buffers do not match [out_features, in_features]")
        self.register_buffer("qweight", qweight)
        self.register_buffer("rowmeta", rowmeta)
        self.register_buffer("fmt", torch.tensor([2 if compact else 1, out_features, in_features], dtype=torch.int32,
                                                 device=device))
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float16, device=device))
        else:
            self.bias = None

    # -- construction -------------------------------------------------------------------
    @classmethod
    def from_packed(cls, p: packing.PackedMXQ, bias: Optional[torch.Tensor] = None) -> "QuantLinear":
        m = cls(p.K, p.N, bias=bias is not None, device=p.device, compact=p.compact, _buffers=(p.qweight, p.rowmeta))
        if bias is not None:
            m.bias = bias.detach().to(device=p.device, dtype=torch.float16)
        return m

    @classmethod
    def from_linear(cls, linear: nn.Linear, dead: Optional[torch.Tensor] = None, compact: bool = False) -> "QuantLinear":
        """MXQ-quantise ``linear.weight`` (fp16 / bf16 / fp32, on the GPU) and pack it.
        ``dead``: optional bool [in_features] mask of never-activated input channels
        (diag(H) == 0, mxqgpt.py:401-403).  ``compact``: compact metadata (packing.compact)."""
        p = packing.quantize_pack(linear.weight.data, dead, compact_meta=compact)
        return cls.from_packed(p, linear.bias.data if linear.bias is not None else None)

    @classmethod
    def fuse(cls, mods) -> "QuantLinear":
        """One module computing several QuantLinears of a shared input in one launch (q | k | v, gate | up):
        format v1 is row-block-major, so the packed weights concatenate along out_features
        (packing.concat_packed); the output is the concatenation of the parts' outputs."""
        mods = list(mods)
        if any((m.bias is None) != (mods[0].bias is None) for m in mods):
            raise ValueError("fuse: either every part has a bias or none")
        bias = torch.cat([m.bias for m in mods]) if mods[0].bias is not None else None
        return cls.from_packed(packing.concat_packed([m.packed() for m in mods]), bias)

    # -- views ----------------------------------------------------------------------------
    def packed(self) -> packing.PackedMXQ:
        return packing.PackedMXQ(self.qweight, self.rowmeta, self.out_features, self.in_features, self.compact)

    def dequantize(self) -> torch.Tensor:
        """fp16 [out, in] weight, bit-identical to the reference's fake-quant write-back."""
        return packing.dequant(self.packed())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = packing.linear(x, self.packed())
        if self.bias is not None:
            y = y + self.bias
        return y

    def extra_repr(self) -> str:
        return (f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, "
                f"format=mxq-v{2 if self.compact else 1} (48x2b+16x4b per 64, {'compact' if self.compact else 'exact'} "
                f"metadata), {self.packed().bits_per_weight():.2f} bit/weight")
