"""The layer-by-layer PTQ loop of ``nas_quant`` (reference mxq_quant/lib/prune.py:338-420) and its helpers
``find_layers``, ``prepare_calibration_input`` and ``check_sparsity`` (prune.py:17-102), minus everything
that needs the network or a checkpoint (model / dataset loading): ``quantize_sequential`` takes the decoder
layers and the calibration hidden states, exactly the state ``nas_quant`` is in at prune.py:366 ("Ready.").

For every layer, as the reference does (prune.py:368-417):
  1. ``find_layers`` collects its ``nn.Linear`` s, one ``MXQGPT`` each (:381-385);
  2. forward hooks feed ``add_batch`` while the calibration samples run through the layer (:389-400);
  3. ``fasterquant(percdamp=0.01, blocksize=16)`` quantises each Linear -- here one fused HIP kernel that
     also leaves the packed weight (:406-414);
  4. the samples run again through the now-quantised layer and become the next layer's inputs (:416-419).
With ``pack=True`` step 3 additionally swaps each ``nn.Linear`` for a ``QuantLinear`` holding the packed
weight (4.5 bit/weight in HBM, HIP GEMM / GEMV forward), which is what a serving checkpoint stores
(mxq_amd/checkpoint.py); the re-run of step 4 then already uses the packed kernels.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch
import torch.nn as nn

from ..quant_linear import QuantLinear
from .mxqgpt import MXQGPT


def find_layers(module: nn.Module, layers=(nn.Linear,), name: str = "") -> Dict[str, nn.Module]:
    """Same contract as the reference's ``find_layers`` (prune.py:17-37): exact-type match,
    dotted names relative to ``module``."""
    if type(module) in tuple(layers):
        return {name: module}
    res: Dict[str, nn.Module] = {}
    for name1, child in module.named_children():
        res.update(find_layers(child, layers=layers, name=name + "." + name1 if name != "" else name1))
    return res


def _set_submodule(root: nn.Module, dotted: str, new: nn.Module) -> None:
    parent = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        parent = getattr(parent, p)
    setattr(parent, parts[-1], new)


def _first(out):
    return out[0] if isinstance(out, (tuple, list)) else out


@torch.no_grad()
def quantize_sequential(layers: Sequence[nn.Module], inps: torch.Tensor, layer_kwargs: Optional[dict] = None,
                        pack: bool = False, percdamp: float = 0.01, blocksize: int = 16,
                        log: Optional[Callable[[str], None]] = None) -> Dict[str, object]:
    """Quantise ``layers`` in order on the calibration batch ``inps`` [nsamples, seq, hidden].

    ``layer_kwargs`` are passed to every layer call (the reference passes ``attention_mask`` and
    ``position_ids``, prune.py:400).  Returns ``{"<layer index>.<linear name>": PackedMXQ}``; the
    layers are modified in place (fake-quant fp16 weights, or QuantLinear modules when ``pack``)."""
    layer_kwargs = dict(layer_kwargs or {})
    if inps.dim() != 3:
        raise ValueError("inps must be [nsamples, seq, hidden]")
    outs = torch.zeros_like(inps)
    packed: Dict[str, object] = {}
    for i, layer in enumerate(layers):
        subset = find_layers(layer)
        gpts = {name: MXQGPT(subset[name]) for name in subset}

        def hook(name):
            def tmp(_, inp, out):
                gpts[name].add_batch(inp[0].data, out.data)
            return tmp

        handles = [subset[name].register_forward_hook(hook(name)) for name in gpts]
        try:
            for j in range(inps.shape[0]):
                outs[j] = _first(layer(inps[j].unsqueeze(0), **layer_kwargs))
        finally:
            for h in handles:
                h.remove()
        for name, g in gpts.items():
            if log:
                log(f"{i} {name}")
            g.fasterquant(percdamp=percdamp, blocksize=blocksize)
            packed[f"{i}.{name}"] = g.packed
            if pack:
                _set_submodule(layer, name, QuantLinear.from_packed(g.packed, bias=subset[name].bias))
            g.free()
        for j in range(inps.shape[0]):
            outs[j] = _first(layer(inps[j].unsqueeze(0), **layer_kwargs))
        inps, outs = outs, inps
    return packed


class _StopForward(Exception):
    """Raised by the input recorder to abandon the rest of the model's forward."""


def prepare_calibration_input(model: nn.Module, dataloader, device, nsamples: int = 128, return_kwargs: bool = False):
    """Record what reaches the first decoder layer for every calibration batch (the reference's Catcher,
    prune.py:64-102): returns ``(inps [nsamples, seqlen, hidden], outs, attention_mask, position_ids)`` --
    the arguments ``quantize_sequential`` / ``nas_quant``'s layer loop start from.  ``model`` is HF-shaped
    (``model.model.layers``, ``model.config.hidden_size / use_cache``, ``model.seqlen``); ``dataloader``
    yields ``(input_ids, ...)`` tuples as ``lib/data.py``'s loaders do.

    ``return_kwargs=True`` appends a fifth item: every keyword argument the model passed to its first layer
    (newer ``transformers`` hand the rotary ``position_embeddings`` to the layers; the reference's pinned
    version needed only the two it names), ready to be given to ``quantize_sequential`` as ``layer_kwargs``."""
    use_cache = getattr(model.config, "use_cache", None)
    model.config.use_cache = False
    layers = model.model.layers
    device = getattr(model, "hf_device_map", {}).get("model.embed_tokens", device)
    dtype = next(iter(model.parameters())).dtype
    inps = torch.zeros((nsamples, model.seqlen, model.config.hidden_size), dtype=dtype, device=device)
    seen = {"n": 0, "attention_mask": None, "position_ids": None}

    class Recorder(nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, inp, **kwargs):
            if seen["n"] < nsamples:
                inps[seen["n"]] = inp
            seen["n"] += 1
            seen["attention_mask"] = kwargs.get("attention_mask")
            seen["position_ids"] = kwargs.get("position_ids")
            seen["kwargs"] = dict(kwargs)
            raise _StopForward

    first = layers[0]
    layers[0] = Recorder(first)
    try:
        for batch in dataloader:
            try:
                model(batch[0].to(device))
            except _StopForward:
                pass
    finally:
        layers[0] = first
        model.config.use_cache = use_cache
    if return_kwargs:
        return inps, torch.zeros_like(inps), seen["attention_mask"], seen["position_ids"], seen.get("kwargs", {})
    return inps, torch.zeros_like(inps), seen["attention_mask"], seen["position_ids"]


def check_sparsity(model: nn.Module, log: Optional[Callable[[str], None]] = print) -> float:
    """Fraction of exactly-zero weights over the decoder layers' Linears, with the reference's per-layer
    report (prune.py:39-63).  Packed ``QuantLinear`` modules are counted through their dequantised weight."""
    use_cache = getattr(model.config, "use_cache", None)
    model.config.use_cache = False
    count = total = 0
    for i, layer in enumerate(model.model.layers):
        sub_count = sub_total = 0
        for lin in find_layers(layer, layers=(nn.Linear, QuantLinear)).values():
            w = lin.dequantize() if isinstance(lin, QuantLinear) else lin.weight.data
            sub_count += int((w == 0).sum().item())
            sub_total += w.numel()
        if log and sub_total:
            log(f"layer {i} sparsity {float(sub_count) / sub_total:.6f}")
        count += sub_count
        total += sub_total
    model.config.use_cache = use_cache
    return float(count) / max(total, 1)


def check_sparsity_linear(layers: Iterable[nn.Module]) -> float:
    """Fraction of exactly-zero weights over all nn.Linear s (the reference's sanity print,
    prune.py:39-66, without its dependence on ``model.config``)."""
    zero = total = 0
    for layer in layers:
        for lin in find_layers(layer).values():
            w = lin.weight.data
            zero += int((w == 0).sum().item())
            total += w.numel()
    return zero / max(total, 1)
