"""mxq_amd -- MI355X-native (gfx950) implementation of MXQ's mixed 2/4-bit hot path.

Host side in Python over a C-ABI HIP library (libmxq_hip.so, include/mxq_hip.h):

* ``mxq_amd.packing``      packed weight container: quantise-and-pack, pack, unpack, dequant
* ``mxq_amd.quant_linear`` ``QuantLinear`` -- nn.Linear replacement on packed W2/4 weights
* ``mxq_amd.utils_quant``  ``MXAsymQuantizer`` / ``QuantizeLinear`` / ``SymQuantizer`` with the
  reference's signatures (LLM-QAT/models/utils_quant.py)
* ``mxq_amd.lib``          ``Quantizer`` / ``MXQGPT`` with the reference's PTQ API
  (mxq_quant/lib/quantizer.py, lib/mxqgpt.py)
* ``mxq_inference_engine`` (top-level module) -- the reference extension's two entry points
"""
from . import _lib  # noqa: F401
from ._lib import MXQLibraryError  # noqa: F401

__version__ = "0.1.0"
