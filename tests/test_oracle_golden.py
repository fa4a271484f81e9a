"""Pin the CPU oracle (oracle/mxq_oracle.py) against golden vectors produced by the
reference's own Python (tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from oracle import mxq_oracle as O

PTQ_KEYS = ("codes2", "sc2", "zero2", "qs2", "qz2", "codes4", "sc4", "zero4", "qs4", "qz4")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_g1_ptq_codes_and_params_bit_exact(g1):
    p = O.mxq_quantize(g1["W"], dead=g1["dead"])
    for k in PTQ_KEYS:
        assert np.array_equal(p[k], g1[k]), k
    s2, s4 = O.mxq_scales(p)
    assert np.array_equal(s2.view(np.uint32), g1["scale2"].view(np.uint32))
    assert np.array_equal(s4.view(np.uint32), g1["scale4"].view(np.uint32))


def test_g1_dequant_bit_exact(g1):
    p = O.mxq_quantize(g1["W"], dead=g1["dead"])
    w16 = p["w_deq32"].astype(np.float16)
    assert np.array_equal(w16.view(np.uint16), g1["w_deq"].view(np.uint16))
    # dequant from the stored parameterisation only (what the kernels see)
    w16b = O.mxq_dequant({k: g1[k] for k in PTQ_KEYS} | {"N": 64, "K": 256}).astype(np.float16)
    assert np.array_equal(w16b.view(np.uint16), g1["w_deq"].view(np.uint16))
    assert g1["dead"][77]                            # dead column was zeroed before quantising


def test_g1_linear(g1):
    y = O.linear_ref(g1["x"], g1["w_deq"])
    np.testing.assert_allclose(y, g1["y32"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["a", "b"])
def test_g2_llama_width_sha(g2, name):
    N, K, seed = (int(v) for v in g2[f"{name}_shape"])
    import torch
    g = torch.Generator().manual_seed(seed)
    W16 = (torch.randn(N, K, generator=g) * 0.02).half().numpy()
    p = O.mxq_quantize(W16)
    p["w_deq"] = p["w_deq32"].astype(np.float16)
    s2, s4 = O.mxq_scales(p)
    p["scale2"], p["scale4"] = s2, s4
    for k in PTQ_KEYS + ("w_deq", "scale2", "scale4"):
        assert _sha(p[k]) == str(g2[f"{name}_sha_{k}"]), k
        ref = g2[f"{name}_rows16_32_{k}"]
        got = p[k][16:32] if p[k].shape[0] == N else p[k][1:2]
        assert np.array_equal(got, ref), k


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_g3_fakequant_bit_exact(g3, dt, bits):
    key = f"{dt}_b{bits}"
    if dt == "bf16":
        w, ref = O.bf16_from_bits(g3[f"{key}_w"]), g3[f"{key}_out"]
        got = O.bf16_bits(O.fakequant_fwd(w, bits, dt))
        gin = O.bf16_bits(O.fakequant_bwd(O.bf16_from_bits(g3[f"{key}_gout"]), w))
        assert np.array_equal(got, ref)
        assert np.array_equal(gin, g3[f"{key}_gin"])
    else:
        w = g3[f"{key}_w"].astype(np.float32)
        got = O.fakequant_fwd(w, bits, dt).astype(g3[f"{key}_out"].dtype)
        ref = g3[f"{key}_out"]
        same = (got == ref) | (np.isnan(got) & np.isnan(ref))      # fp16 constant group -> NaN (H5)
        assert same.all()
        gin = O.fakequant_bwd(g3[f"{key}_gout"], w)
        assert np.array_equal(gin, g3[f"{key}_gin"])
    if dt == "fp16":
        assert np.isnan(g3[f"{key}_out"][6, 0:16].astype(np.float32)).all()


@pytest.mark.parametrize("K", [4096, 11008])
def test_g3_fakequant_llama_width_bf16(g3, K):
    w = O.bf16_from_bits(g3[f"bf16_K{K}_w"])
    got = O.bf16_bits(O.fakequant_fwd(w, 2, "bf16"))
    assert np.array_equal(got, g3[f"bf16_K{K}_out"])


def test_g6_kat_proto_format(g6):
    """cuda_kernel/test_correct_gemv.py: every output == 4096."""
    N, K = int(g6["N"]), int(g6["K"])
    sub = 64                                           # a slice of rows is enough on CPU
    full = lambda shape, v, dt: np.full(shape, v, dt)
    y = O.gemv_mxq_proto_ref(
        full((1, K), g6["x"], np.float16),
        full((sub, 256), g6["weight_2b"], np.uint32), full((sub, 64), g6["weight_4b"], np.uint32),
        full((sub, 32), g6["zeros_and_scales_1st"], np.uint32),
        full((sub // 4, 256), g6["scales_2nd"], np.float16),
        full((sub // 4, 32), g6["zeros_2nd"], np.uint32),
        full((sub,), g6["scales_4b"], np.float16), full((sub // 8,), g6["zeros_4b"], np.uint32))
    assert np.all(y.astype(np.int32) == int(g6["expected"]))


@pytest.mark.parametrize("layout,pre", [("w2g16", "w2"), ("w4row", "w4")])
def test_g8_uniform_arms(layout, pre):
    """Uniform W2 (group 16) / W4 (per row) arms of the config-5 sweep vs the reference's Quantizer."""
    from tests.conftest import load_golden
    g8 = load_golden("g8_uniform.npz")
    p = O.uniform_quantize(g8["W"], layout)
    for k in ("codes", "sc", "zero", "qs", "qz"):
        assert np.array_equal(p[k], g8[f"{pre}_{k}"]), k
    assert np.array_equal(p["w_deq32"].astype(np.float16).view(np.uint16), g8[f"{pre}_wdeq"].view(np.uint16))


# ----------------------------------------------------------------------------------------
# activation / KV fake quantisers (oracle/act_quant.py) against the reference's outputs
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("q", ["sym", "asym"])
@pytest.mark.parametrize("case", ["w2d", "a3d", "a3d_long", "s4d"])
def test_g7_activation_quantizers(q, case):
    import torch
    from oracle.act_quant import AsymQuantizer, SymQuantizer
    from tests.conftest import load_golden
    g7 = load_golden("g7_act_quantizers.npz")
    Q = SymQuantizer if q == "sym" else AsymQuantizer
    clip = torch.tensor([-2.0, 2.0])
    for bits in (4, 16):
        for lw in (0, 1):
            key = f"{case}_{q}_b{bits}_{lw}"
            x = torch.from_numpy(g7[key + "_x"]).requires_grad_()
            y = Q.apply(x, clip, bits, bool(lw))
            y.backward(torch.from_numpy(g7[key + "_gy"]))
            assert np.array_equal(y.detach().numpy(), g7[key + "_y"]), key
            assert np.array_equal(x.grad.numpy(), g7[key + "_gx"]), key


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_g9_activation_quantizers_16bit(dt):
    import torch
    from oracle.act_quant import AsymQuantizer, SymQuantizer
    from tests.conftest import load_golden
    g9 = load_golden("g9_act16.npz")
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    clip = torch.tensor([-2.0, 2.0])
    keys = sorted(k[:-2] for k in g9.files if k.startswith(dt) and k.endswith("_x"))
    assert len(keys) == 60
    for key in keys:
        _, _, qn, bits, lw = key.rsplit("_", 4)[-5:] if False else (None, None, *key.split("_")[-3:])
        Q = SymQuantizer if qn == "sym" else AsymQuantizer
        x = torch.from_numpy(g9[key + "_x"].view(np.int16)).view(tdt)
        y = Q.apply(x, clip, int(bits[1:]), bool(int(lw)))
        assert np.array_equal(y.view(torch.int16).numpy().view(np.uint16), g9[key + "_y"]), key


def test_compact_metadata_stays_inside_the_gemm_budget():
    """CPU statement of the compact mode's cost: rounding the 2-bit zero-points to fp16 leaves the
    codes alone and moves y = x . W'^T by well under the 1e-3 budget against the exact reference weight."""
    rng = np.random.default_rng(3)
    W = (rng.standard_normal((256, 1024)) * 0.02).astype(np.float16)
    p = O.mxq_quantize(W)
    c = O.mxq_compact_params(p)
    assert all(np.array_equal(p[k], c[k]) for k in ("codes2", "codes4", "sc2", "sc4")) and not np.array_equal(p["zero2"], c["zero2"])
    x = rng.standard_normal((64, 1024)).astype(np.float16)
    y_exact = O.linear_ref(x, O.mxq_dequant(p).astype(np.float16))
    y_comp = O.linear_ref(x, O.mxq_dequant(c).astype(np.float16))
    err_fro = np.linalg.norm(y_comp - y_exact) / np.linalg.norm(y_exact)
    err_max = np.abs(y_comp - y_exact).max() / np.abs(y_exact).max()
    assert err_fro < 6e-4 and err_max < 8e-4, (err_fro, err_max)


def test_gemm_awq_operand_restatement_nibble_order():
    """oracle.gemm_awq_*: the interleaved nibble order of the reference GEMM's operands (dequantize.cuh:35-51: the
    conversion returns nibbles (0, 4, 1, 5, 2, 6, 3, 7) as elements 0..7) -- a known-answer word and pack / unpack
    round trips.  (Parity unpinned for this entry: the reference never builds the kernel and ships no vector for it.)"""
    import numpy as np
    from oracle import mxq_oracle as O
    word = np.array([[0x76543210]], np.uint32).view(np.int32)          # nibble i holds the value i
    assert O.gemm_awq_unpack(word).tolist() == [[0, 4, 1, 5, 2, 6, 3, 7]]
    rng = np.random.default_rng(0)
    q = rng.integers(0, 16, size=(64, 128))
    assert np.array_equal(O.gemm_awq_unpack(O.gemm_awq_pack(q)), q)
    s = (rng.random((2, 128)) * 0.01 + 0.001).astype(np.float16)
    z = rng.integers(0, 16, size=(2, 128))
    w = O.gemm_awq_weight(O.gemm_awq_pack(q), s, O.gemm_awq_pack(z), 32)
    k, n = 40, 77
    want = np.float16((np.float32(q[k, n]) - np.float32(z[1, n])) * np.float32(s[1, n]))
    assert w.shape == (64, 128) and w[k, n] == want
