"""Host-side logic that needs no GPU: error behaviour of the wrappers, module surface,
state_dict layout, the calibration-driver helpers and the packed checkpoint format."""
import numpy as np
import pytest
import torch

from mxq_amd import packing
from mxq_amd.quant_linear import QuantLinear
from mxq_amd.utils_quant import AsymQuantizer, MXAsymQuantizer, QuantizeLinear, SymQuantizer


def test_no_cpu_fallback():
    with pytest.raises(ValueError, match="GPU only"):
        packing.quantize_pack(torch.zeros(16, 64, dtype=torch.float16))
    with pytest.raises(ValueError, match="GPU only"):
        MXAsymQuantizer.apply(torch.zeros(16, 64), torch.tensor([-2.0, 2.0]), 2, False)
    lin = QuantizeLinear(64, 16, w_bits=2, a_bits=32)
    with pytest.raises(ValueError, match="GPU only"):
        lin(torch.zeros(1, 64))


def test_dead_branches_raise_like_the_reference():
    with pytest.raises(UnboundLocalError):
        MXAsymQuantizer.apply(torch.zeros(16, 64), torch.tensor([-2.0, 2.0]), 2, True)


def test_shape_checks():
    for N, K in ((15, 64), (16, 100), (0, 64)):
        with pytest.raises(ValueError):
            packing.check_shape(N, K)
    assert packing.qweight_bytes(4096, 4096) == 9437184
    with pytest.raises(ValueError):
        QuantLinear(100, 16)


def test_quantlinear_state_dict_layout():
    m = QuantLinear(256, 64, bias=True)
    sd = m.state_dict()
    assert set(sd) == {"qweight", "rowmeta", "fmt", "bias"}
    assert sd["qweight"].dtype == torch.int32 and sd["qweight"].numel() == 4 * 4 * 144
    assert sd["rowmeta"].shape == (64, 4) and sd["fmt"].tolist() == [1, 64, 256]
    m2 = QuantLinear(256, 64, bias=True)
    m2.load_state_dict(sd)
    assert "QuantLinear" in repr(m2) and "bit/weight" in repr(m2)


def test_quantizelinear_surface_matches_reference():
    lin = QuantizeLinear(256, 64, w_bits=32, a_bits=32)
    assert set(lin.state_dict()) == {"weight"} and lin.bias is None        # utils_quant.py:613
    x = torch.randn(2, 3, 256)
    assert torch.equal(lin(x), torch.nn.functional.linear(x, lin.weight))
    assert QuantizeLinear(256, 64, w_bits=2, a_bits=16).act_quantizer is SymQuantizer
    assert QuantizeLinear(256, 64, w_bits=2, a_bits=8, symmetric=False).act_quantizer is AsymQuantizer
    # w_bits == 1 sign path runs in torch and keeps the straight-through gradient
    l1 = QuantizeLinear(64, 16, w_bits=1)
    y = l1(torch.randn(4, 64)); y.sum().backward()
    assert l1.weight.grad is not None and torch.isfinite(y).all()


def test_activation_quantizers_have_no_cpu_path():
    for Q in (SymQuantizer, AsymQuantizer):
        with pytest.raises(ValueError, match="GPU only"):
            Q.apply(torch.zeros(2, 8, 256), torch.tensor([-2.0, 2.0]), 8, False)


def test_inference_engine_validation():
    import mxq_inference_engine as eng
    x = torch.zeros(1, 4096, dtype=torch.float16)
    with pytest.raises(ValueError):
        eng.gemv_forward_cuda(x, x, x, x, 128)          # CPU tensors are rejected
    with pytest.raises(ValueError):
        eng.gemm_forward_cuda(x, x, x, x, 1)            # ... by every entry
    assert callable(eng.gemv_mxq_forward_cuda)


# ----------------------------------------------------------------------------------------
# calibration driver helpers and the packed checkpoint (no GPU: module surgery and file format only)
# ----------------------------------------------------------------------------------------
class _Block(torch.nn.Module):
    def __init__(self, h=64, inter=128):
        super().__init__()
        self.attn = torch.nn.Module()
        self.attn.q_proj = torch.nn.Linear(h, h, bias=False)
        self.attn.o_proj = torch.nn.Linear(h, h, bias=True)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(h, inter, bias=False), torch.nn.SiLU(),
                                       torch.nn.Linear(inter, h, bias=False))
        self.norm = torch.nn.LayerNorm(h)


def test_find_layers_names_like_the_reference():
    from mxq_amd.lib.prune import find_layers
    b = _Block()
    found = find_layers(b)
    assert list(found) == ["attn.q_proj", "attn.o_proj", "mlp.0", "mlp.2"]       # prune.py:17-37: dotted, exact type
    assert found["mlp.2"] is b.mlp[2]
    assert find_layers(b.mlp[0], name="x") == {"x": b.mlp[0]}
    assert find_layers(b, layers=[torch.nn.LayerNorm]) == {"norm": b.norm}


def test_packed_checkpoint_roundtrip_and_errors(tmp_path):
    from mxq_amd import checkpoint
    from mxq_amd.lib.prune import find_layers
    model = torch.nn.ModuleDict({"layers": torch.nn.ModuleList([_Block(), _Block()]), "lm_head": torch.nn.Linear(64, 32, bias=False)})
    with pytest.raises(ValueError, match="no QuantLinear"):
        checkpoint.save_packed(model, str(tmp_path / "none"))
    g = torch.Generator().manual_seed(0)
    for li, blk in enumerate(model["layers"]):
        for name, lin in find_layers(blk).items():
            q = QuantLinear(lin.in_features, lin.out_features, bias=lin.bias is not None)
            q.qweight.copy_(torch.randint(-2**31, 2**31 - 1, q.qweight.shape, generator=g, dtype=torch.int64).to(torch.int32))
            q.rowmeta.copy_(torch.randn(q.rowmeta.shape, generator=g))
            if q.bias is not None:
                q.bias.copy_(torch.randn(q.bias.shape, generator=g).half())
            checkpoint._set_submodule(blk, name, q)
    d = checkpoint.save_packed(model, str(tmp_path / "ck"))
    import json, os
    cfg = json.load(open(os.path.join(d, "mxq_config.json")))
    assert cfg["format"] == "mxq-v1" and set(cfg["quantized"]) == {f"layers.{i}.{n}" for i in (0, 1) for n in
                                                                  ("attn.q_proj", "attn.o_proj", "mlp.0", "mlp.2")}
    assert cfg["quantized"]["layers.0.attn.o_proj"] == {"in_features": 64, "out_features": 64, "bias": True, "metadata": "exact"}
    fresh = torch.nn.ModuleDict({"layers": torch.nn.ModuleList([_Block(), _Block()]), "lm_head": torch.nn.Linear(64, 32, bias=False)})
    checkpoint.load_packed(fresh, d)
    assert isinstance(fresh["layers"][1].mlp[2], QuantLinear) and type(fresh["lm_head"]) is torch.nn.Linear
    a, b = model.state_dict(), fresh.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    # wrong architecture / wrong format fail loudly
    small = torch.nn.ModuleDict({"layers": torch.nn.ModuleList([_Block(128, 128), _Block(128, 128)]), "lm_head": torch.nn.Linear(64, 32, bias=False)})
    with pytest.raises(ValueError, match="checkpoint is"):
        checkpoint.load_packed(small, d)
    with pytest.raises(KeyError):
        checkpoint.load_packed(torch.nn.ModuleDict({"lm_head": torch.nn.Linear(64, 32)}), d)
    cfg["format"] = "other"
    json.dump(cfg, open(os.path.join(d, "mxq_config.json"), "w"))
    with pytest.raises(ValueError, match="not an mxq-v1"):
        checkpoint.load_packed(fresh, d)


def test_pack_model_needs_the_gpu():
    from mxq_amd import checkpoint
    with pytest.raises(ValueError, match="GPU only"):
        checkpoint.pack_model(torch.nn.Sequential(torch.nn.Linear(64, 16)))


def test_prepare_calibration_input_and_check_sparsity():
    """The HF-shaped helpers around the layer loop (prune.py:39-102) on a toy model, CPU only."""
    import types
    from mxq_amd.lib.prune import check_sparsity, prepare_calibration_input

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(64, 64, bias=False)

        def forward(self, x, attention_mask=None, position_ids=None):
            return (self.lin(x),)

    class Inner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embed_tokens = torch.nn.Embedding(100, 64)
            self.layers = torch.nn.ModuleList([Layer(), Layer()])

        def forward(self, ids):
            h = self.embed_tokens(ids)
            pos = torch.arange(ids.shape[1])[None]
            for l in self.layers:
                h = l(h, attention_mask=None, position_ids=pos)[0]
            return h

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Inner()
            self.config = types.SimpleNamespace(use_cache=True, hidden_size=64)
            self.seqlen = 8

        def forward(self, ids):
            return self.model(ids)

    torch.manual_seed(0)
    m = Model()
    loader = [(torch.randint(0, 100, (1, 8)), None) for _ in range(5)]
    inps, outs, mask, pos = prepare_calibration_input(m, loader, "cpu", nsamples=4)
    assert inps.shape == (4, 8, 64) and outs.shape == inps.shape and mask is None and pos.tolist() == [list(range(8))]
    assert torch.equal(inps[2], m.model.embed_tokens(loader[2][0])[0])     # what layer 0 received for batch 2
    assert isinstance(m.model.layers[0], Layer) and m.config.use_cache is True     # model restored
    with torch.no_grad():
        m.model.layers[1].lin.weight[:16] = 0
    lines = []
    assert check_sparsity(m, log=lines.append) == pytest.approx(0.125)
    assert lines == ["layer 0 sparsity 0.000000", "layer 1 sparsity 0.250000"]


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/LLM-QAT/models"),
                    reason="needs the read-only reference checkout (build container only)")
def test_reference_modeling_file_imports_against_this_utils_quant():
    """with ``models.utils_quant`` aliased to mxq_amd.utils_quant, the reference's
    own modeling_llama_quant.py imports and builds its decoder layer out of THIS repo's QuantizeLinear, with
    the state_dict keys the reference checkpoints use.  (Forward / backward parity of that layer is golden G5
    on the GPU.)  Runs in a subprocess so the aliasing does not leak into other tests."""
    import subprocess
    import sys
    code = r'''
import sys
sys.dont_write_bytecode = True
sys.path.insert(0, %r)
import mxq_amd.utils_quant as uq
sys.path.insert(0, "/root/reference/LLM-QAT")
sys.modules["models.utils_quant"] = uq
import importlib
m = importlib.import_module("models.modeling_llama_quant")
from models.configuration_llama import LlamaConfig
cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_attention_heads=4, num_hidden_layers=1, vocab_size=128)
cfg.w_bits, cfg.a_bits, cfg.kv_bits = 2, 16, 16
layer = m.LlamaDecoderLayer(cfg)
lin = [layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj, layer.self_attn.o_proj,
       layer.mlp.gate_proj, layer.mlp.up_proj, layer.mlp.down_proj]
assert all(type(l) is uq.QuantizeLinear and l.w_bits == 2 and l.a_bits == 16 for l in lin)
assert m.SymQuantizer is uq.SymQuantizer and lin[0].act_quantizer is uq.SymQuantizer
keys = set(layer.state_dict())
assert {"self_attn.q_proj.weight", "mlp.down_proj.weight", "input_layernorm.weight"} <= keys
assert not any(k.endswith(".bias") for k in keys)
print("OK")
''' % (__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))),)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stderr[-2000:]


def test_context_window_guards_the_kv_cache():
    """decoding past max_ctx must fail on the host instead of overrunning the cache on the device."""
    from mxq_amd.llama_decode import ContextWindow
    w = ContextWindow(4)
    assert [w.take(), w.take(2)] == [0, 1] and w.pos == 3
    w.take(1)
    with pytest.raises(RuntimeError, match="exceeds the KV cache"):
        w.take(1)
    assert w.pos == 4                       # a refused reservation leaves the position alone
    w.reset()
    assert w.take(4) == 0
    with pytest.raises(RuntimeError):
        w.take(1)
    with pytest.raises(ValueError):
        ContextWindow(0)
