"""PTQ op surface mirroring the reference's ``mxq_quant/lib`` package names
(``lib.quantizer.Quantizer``, ``lib.mxqgpt.MXQGPT``)."""
