"""Packed checkpoint: the on-disk form of a model whose ``nn.Linear`` s are ``QuantLinear`` s.

The reference saves the fake-quantised fp16 model with ``model.save_pretrained(args.save_model)``
(mxq_quant/main.py:96-100), i.e. 16 bit/weight on disk for a 2.5-bit model; this is the packed
counterpart: one ``model.safetensors`` holding the ordinary ``state_dict``
(``<name>.qweight`` int32, ``<name>.rowmeta`` float32 [N, 4], ``<name>.fmt`` int32 (version, N, K) with version 1 =
exact metadata (4.5 bit/weight) or 2 = compact metadata (fp16 zero-points, 3.75 bit/weight),
optional ``<name>.bias``; every non-quantised tensor as it is) plus ``mxq_config.json`` naming the
quantised modules, so that ``load_packed`` can rebuild the module tree before ``load_state_dict``.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from .quant_linear import QuantLinear

FORMAT = "mxq-v1"
CONFIG_NAME = "mxq_config.json"
WEIGHTS_NAME = "model.safetensors"


def _set_submodule(root: nn.Module, dotted: str, new: nn.Module) -> None:
    parent = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        parent = getattr(parent, p)
    setattr(parent, parts[-1], new)


def pack_model(model: nn.Module, skip: Iterable[str] = ("lm_head",), compact: bool = False) -> List[str]:
    """Replace every ``nn.Linear`` of ``model`` (on the GPU) by ``QuantLinear.from_linear`` -- the
    round-to-nearest MXQ quantisation of its weight -- except modules whose dotted name ends with an
    entry of ``skip`` (the reference quantises decoder layers only, prune.py:347,368).  Returns the
    names replaced.  For calibrated quantisation use ``mxq_amd.lib.prune.quantize_sequential(pack=True)``."""
    skip = tuple(skip)
    names = [n for n, m in model.named_modules() if type(m) is nn.Linear and not any(n == s or n.endswith("." + s) for s in skip)]
    for n in names:
        lin = dict(model.named_modules())[n]
        _set_submodule(model, n, QuantLinear.from_linear(lin, compact=compact))
    return names


def quantized_modules(model: nn.Module) -> Dict[str, QuantLinear]:
    return {n: m for n, m in model.named_modules() if isinstance(m, QuantLinear)}


def save_packed(model: nn.Module, directory: str) -> str:
    """Write ``model.safetensors`` + ``mxq_config.json`` into ``directory``; returns the directory."""
    from safetensors.torch import save_file
    os.makedirs(directory, exist_ok=True)
    q = quantized_modules(model)
    if not q:
        raise ValueError("model holds no QuantLinear module: nothing packed to save")
    # format_version: 1 = every module carries exact metadata (fmt v1); 2 = at least one module is compact (fmt v2)
    cfg = {"format": FORMAT, "format_version": 2 if any(m.compact for m in q.values()) else 1,
           "quantized": {n: {"in_features": m.in_features, "out_features": m.out_features, "bias": m.bias is not None,
                             "metadata": "compact" if m.compact else "exact"}
                         for n, m in q.items()}}
    sd = {k: v.detach().contiguous().cpu() for k, v in model.state_dict().items()}
    save_file(sd, os.path.join(directory, WEIGHTS_NAME), metadata={"format": FORMAT})
    with open(os.path.join(directory, CONFIG_NAME), "w") as f:
        json.dump(cfg, f, indent=1, sort_keys=True)
    return directory


def load_packed(model: nn.Module, directory: str, device: Optional[torch.device] = None) -> nn.Module:
    """Turn ``model`` (same architecture, any weights, ``nn.Linear`` s in place) into the packed model
    saved in ``directory``: the modules named in ``mxq_config.json`` become ``QuantLinear`` s, then the
    whole ``state_dict`` is loaded strictly.  ``device``: where the packed buffers go (default: the
    device of the module being replaced)."""
    from safetensors.torch import load_file
    with open(os.path.join(directory, CONFIG_NAME)) as f:
        cfg = json.load(f)
    if cfg.get("format") != FORMAT:
        raise ValueError(f"{directory}: not an {FORMAT} checkpoint (format={cfg.get('format')!r})")
    mods = dict(model.named_modules())
    for n, spec in cfg["quantized"].items():
        if n not in mods:
            raise KeyError(f"checkpoint quantises {n!r}, which the model does not have")
        old = mods[n]
        dev = device
        if dev is None:
            dev = next((t.device for t in list(old.parameters()) + list(old.buffers())), torch.device("cpu"))
        if isinstance(old, nn.Linear) and (old.in_features, old.out_features) != (spec["in_features"], spec["out_features"]):
            raise ValueError(f"{n}: checkpoint is {spec['out_features']}x{spec['in_features']}, "
                             f"model has {old.out_features}x{old.in_features}")
        _set_submodule(model, n, QuantLinear(spec["in_features"], spec["out_features"], bias=spec["bias"], device=dev,
                                             compact=spec.get("metadata", "exact") == "compact"))
    sd = load_file(os.path.join(directory, WEIGHTS_NAME))
    for n, spec in cfg["quantized"].items():
        fmt = sd.get(n + ".fmt")
        ver = 2 if spec.get("metadata", "exact") == "compact" else 1
        if fmt is None or fmt.tolist() != [ver, spec["out_features"], spec["in_features"]]:
            raise ValueError(f"{n}: fmt header {None if fmt is None else fmt.tolist()} does not match mxq_config.json")
    model.load_state_dict(sd, strict=True)
    return model
