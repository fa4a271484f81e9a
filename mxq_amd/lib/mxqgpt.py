"""``MXQGPT`` with the reference's driver API (mxq_quant/lib/mxqgpt.py:353-452):
``MXQGPT(layer)``, ``.add_batch(inp, out)``, ``.fasterquant(percdamp, blocksize)``, ``.free()``
-- the calls ``nas_quant`` makes (mxq_quant/lib/prune.py:385,391,409,414).

Differences in mechanism, not in result:
* ``add_batch`` keeps only what ``fasterquant`` ever uses of the Hessian -- whether
  ``diag(H) == 0`` (mxqgpt.py:401-403), i.e. whether an input channel was ever non-zero --
  instead of a K x K fp32 GEMM per sample.
* ``fasterquant`` is one fused HIP kernel (quantise + pack) followed by the dequant kernel
  that writes the fake-quant weight back into ``layer.weight.data`` (same values, same
  dtype, as mxqgpt.py:448).  The packed form is kept in ``self.packed`` so the caller can
  swap in ``QuantLinear.from_packed``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import packing
from .quantizer import GroupView


class MXQGPT:
    def __init__(self, layer):
        if not isinstance(layer, nn.Linear):
            raise TypeError("MXQGPT supports nn.Linear layers (the Llama projections)")
        self.layer = layer
        self.dev = layer.weight.device
        self.rows, self.columns = layer.weight.shape
        self.nsamples = 0
        self.seen = torch.zeros(self.columns, dtype=torch.bool, device=self.dev)
        self.packed = None
        self._params = None

    def add_batch(self, inp, out=None):
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        self.nsamples += inp.shape[0]
        # diag(H)[k] == 0  <=>  every sample had x[k] == 0  (H = sum of x x^T, :381-383)
        self.seen |= (inp.reshape(-1, inp.shape[-1]) != 0).any(dim=0).to(self.seen.device)

    def fasterquant(self, blocksize=16, percdamp=0.01):
        if blocksize != 16:
            raise ValueError("the MXQ layout fixes the 2-bit group size at 16 (prune.py:409)")
        W = self.layer.weight.data
        dead = ~self.seen if self.nsamples > 0 else None
        self.packed = packing.quantize_pack(W, dead)
        self._params = None
        wq = packing.dequant(self.packed)
        self.layer.weight.data = wq.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)

    # -- inspection helpers (what the reference keeps in loop-local Quantizer objects) ----
    def params(self):
        if self._params is None:
            self._params = packing.unpack(self.packed)
        return self._params

    def quantizer(self, chunk: int, group: int) -> GroupView:
        p, j = self.params(), 3 * chunk + group
        lo = chunk * 64 + group * 16
        return GroupView(2, p["codes2"][:, chunk * 48 + group * 16: chunk * 48 + group * 16 + 16], p["sc2"][:, j],
                         p["zero2"][:, j], p["qs2"][:, j], p["qz2"][:, j], self.layer.weight.data[:, lo:lo + 16])

    @property
    def quantizer_4b(self) -> GroupView:
        p = self.params()
        idx = torch.arange(self.columns, device=self.dev).reshape(-1, 64)[:, 48:].reshape(-1)
        return GroupView(4, p["codes4"], p["sc4"], p["zero4"], p["qs4"], p["qz4"], self.layer.weight.data[:, idx])

    def free(self):
        self.seen = None
        self._params = None
