"""GPU parity tests: every call goes through the C-ABI (libmxq_hip.so) and is compared with
the CPU oracle (oracle/mxq_oracle.py, pinned to the reference by tests/test_oracle_golden.py)
and with the committed golden vectors.  Run on the MI355X box:  pytest tests -m gpu

Bars: integer unpack bit-exact; dequant tile bit-exact fp16; GEMM
max|dy|/max|y| <= 1e-3 and Frobenius <= 1e-3 against x16 . W_deq16^T in fp32; fake-quant and
STE backward bit-exact in fp32 / bf16 / fp16."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import mxq_oracle as O

pytestmark = pytest.mark.gpu

KEYS = ("codes2", "sc2", "zero2", "qs2", "qz2", "codes4", "sc4", "zero4", "qs4", "qz4")
REL_TOL = 1e-3      # BASELINE.json north_star: "within 1e-3 relative on the fp16 GEMM result"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from mxq_amd import _lib
    _lib.load()        # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _to_dev(p, dev):
    return {k: torch.from_numpy(np.ascontiguousarray(p[k])).to(dev) for k in KEYS}


def _check_gemm(y, yref, what=""):
    y = y.astype(np.float32)
    emax = np.abs(y - yref).max() / np.abs(yref).max()
    efro = np.linalg.norm(y - yref) / np.linalg.norm(yref)
    assert emax <= REL_TOL and efro <= REL_TOL, f"{what}: max-rel {emax:.2e}, fro-rel {efro:.2e}"
    return emax, efro


# ----------------------------------------------------------------------------------------
# quantise + pack + unpack + dequant
# ----------------------------------------------------------------------------------------
def test_g1_quantize_pack_unpack_bit_exact(dev, g1):
    from mxq_amd import packing
    p = packing.quantize_pack(torch.from_numpy(g1["W"]).to(dev), torch.from_numpy(g1["dead"]).to(dev))
    got = packing.unpack(p)
    for k in KEYS:
        assert np.array_equal(got[k].cpu().numpy(), g1[k]), k
    w = packing.dequant(p).cpu().numpy()
    assert np.array_equal(w.view(np.uint16), g1["w_deq"].view(np.uint16))


def test_g1_pack_codes_roundtrip(dev, g1):
    from mxq_amd import packing
    p = packing.pack_codes(_to_dev(g1, dev), 64, 256)
    got = packing.unpack(p)
    for k in KEYS:
        assert np.array_equal(got[k].cpu().numpy(), g1[k]), k
    w = packing.dequant(p).cpu().numpy()
    assert np.array_equal(w.view(np.uint16), g1["w_deq"].view(np.uint16))
    # the kernel-packed and the codes-packed buffers are the same bytes
    p2 = packing.quantize_pack(torch.from_numpy(g1["W"]).to(dev), torch.from_numpy(g1["dead"]).to(dev))
    assert torch.equal(p.qweight, p2.qweight) and torch.equal(p.rowmeta, p2.rowmeta)


@pytest.mark.parametrize("seed,N,K,zscale,sscale", [(1, 32, 320, 1e-3, 3e-4), (2, 16, 704, 37.0, 0.02), (3, 64, 1024, 1e4, 5.0),
                                                    (4, 48, 64, 1.0, 1e-6)])
def test_arbitrary_params_pack_unpack_dequant_bit_exact(dev, seed, N, K, zscale, sscale):
    """Parameter sets no quantiser would produce (random codes, scale codes, zero-points of any magnitude,
    a zero scale, -0.0): the pack / unpack kernels are the identity on them and the LUT / v_perm dequant
    kernel equals the oracle's fp32 formula rounded once to fp16, overflow to inf included.  Same generator
    as the CPU property test of the host-compiled helpers (tests/test_format_host_emu.py)."""
    from mxq_amd import packing
    rng = np.random.default_rng(seed)
    nc, rb = K // 64, N // 16
    G = 3 * nc
    p = dict(
        codes2=rng.integers(0, 4, (N, 48 * nc), dtype=np.uint8), sc2=rng.integers(0, 16, (N, G), dtype=np.uint8),
        zero2=(rng.standard_normal((N, G)) * zscale).astype(np.float32),
        qs2=(np.abs(rng.standard_normal((rb, G))) * sscale).astype(np.float32),
        qz2=(rng.standard_normal((rb, G)) * 6).astype(np.float32),
        codes4=rng.integers(0, 16, (N, 16 * nc), dtype=np.uint8), sc4=rng.integers(0, 16, (N,), dtype=np.uint8),
        zero4=(rng.standard_normal((N,)) * zscale).astype(np.float32),
        qs4=(np.abs(rng.standard_normal((rb,))) * sscale).astype(np.float32),
        qz4=(rng.standard_normal((rb,)) * 6).astype(np.float32), N=N, K=K)
    p["zero2"][0, 0] = 0.0
    p["zero2"][-1, -1] = -0.0
    p["qs2"][0, 0] = 0.0
    pk = packing.pack_codes(_to_dev(p, dev), N, K)
    got = packing.unpack(pk)
    for k in KEYS:
        a, b = got[k].cpu().numpy(), p[k]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                              b.view(np.uint32) if b.dtype == np.float32 else b), k
    with np.errstate(over="ignore", invalid="ignore"):
        w16 = O.mxq_dequant(p).astype(np.float16)
    assert np.array_equal(packing.dequant(pk).cpu().numpy().view(np.uint16), w16.view(np.uint16))


@pytest.mark.parametrize("name", ["a", "b"])
def test_g2_llama_width_sha(dev, g2, name):
    from mxq_amd import packing
    N, K, seed = (int(v) for v in g2[f"{name}_shape"])
    g = torch.Generator().manual_seed(seed)
    W16 = (torch.randn(N, K, generator=g) * 0.02).half()
    p = packing.quantize_pack(W16.to(dev))
    got = {k: v.cpu().numpy() for k, v in packing.unpack(p).items()}
    got["w_deq"] = packing.dequant(p).cpu().numpy()
    for k in KEYS + ("w_deq",):
        assert _sha(got[k]) == str(g2[f"{name}_sha_{k}"]), k


@pytest.mark.parametrize("N,K,dt", [(32, 64, torch.float16), (48, 320, torch.float16), (16, 704, torch.bfloat16),
                                    (128, 1024, torch.float32), (4096, 4096, torch.float16)])
def test_quantize_random_vs_oracle(dev, N, K, dt):
    """Ragged chunk counts (K = 320, 704: tile padding), all three input dtypes, full size."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(N + K)
    W = (torch.randn(N, K, generator=g) * 0.02).to(dt)
    W[0, :16] = 0.5
    ref = O.mxq_quantize(W.float().numpy())
    p = packing.quantize_pack(W.to(dev))
    got = packing.unpack(p)
    for k in KEYS:
        assert np.array_equal(got[k].cpu().numpy(), ref[k]), k
    w = packing.dequant(p).cpu().numpy()
    assert np.array_equal(w.view(np.uint16), ref["w_deq32"].astype(np.float16).view(np.uint16))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float32])
def test_quantize_pack_division_screen_adversarial(dev, dt):
    """The quantise kernels replace x / s by x * rcp(s) behind a rounding-boundary screen (csrc/pack.hip quant_key) and
    fall back to the division per wave: inputs built to sit ON and NEXT TO the boundaries -- values on a k/2 grid of the
    group's own scale (exact ties), groups riding on large offsets (|x / s| ~ 1e3: the bound grows with it), tiny and
    huge ranges, constant groups, one-ulp neighbours of ties -- must give the oracle's codes bit for bit."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(7)
    N, K = 64, 1024
    W = torch.empty(N, K)
    for n in range(N):
        for c in range(K // 16):
            kind = (n + 3 * c) % 8
            step = float(2.0 ** torch.randint(-12, 3, (1,), generator=g).item())
            kk = torch.randint(0, 7, (16,), generator=g).float() * 0.5            # multiples of half a step: ties
            if kind == 0:
                v = kk * step
            elif kind == 1:
                v = 1000.0 * step + kk * step                                       # large offset
            elif kind == 2:
                v = -517.0 * step + kk * step
            elif kind == 3:
                v = torch.full((16,), 0.37 * step)                                  # constant group
            elif kind == 4:
                v = torch.randn(16, generator=g) * 1e-6
            elif kind == 5:
                v = torch.randn(16, generator=g) * 3e3                              # (finite in fp16)
            elif kind == 6:
                v = kk * step
                v = torch.nextafter(v, torch.full_like(v, 1e9)) if dt == torch.float32 else v * (1 + 2.0 ** -9)
            else:
                v = torch.randn(16, generator=g) * step
            W[n, c * 16:(c + 1) * 16] = v
    W = W.to(dt)
    ref = O.mxq_quantize(W.float().numpy())
    p = packing.quantize_pack(W.to(dev))
    got = packing.unpack(p)
    for k in KEYS:
        assert np.array_equal(got[k].cpu().numpy(), ref[k]), k
    for layout in ("w2g16", "w4row"):
        refu = O.uniform_quantize(W.float().numpy(), layout)
        gotu = packing.expand_uniform(packing.quantize_pack_uniform(W.to(dev), layout))[1]
        for k in ("codes", "sc", "zero", "qs", "qz"):
            assert np.array_equal(gotu[k].cpu().numpy().reshape(refu[k].shape), refu[k]), (layout, k)


def test_quantize_pack_nonfinite_and_huge_scale(dev):
    """ADVICE r2 on the division screen (csrc/pack.hip): (a) scales beyond 2^126, where v_rcp_f32 flushes its result
    to zero, must still give the oracle's codes (such groups are sent to the IEEE division); (b) a NaN weight gets
    code 0 and leaves every other code and parameter of the tensor untouched (min / max ignore it, its key is dropped
    by v_min_f32: the fast path and the division path agree on 0)."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(11)
    W = torch.randn(32, 128, generator=g) * 0.02
    W[:16] = (torch.rand(16, 128, generator=g) - 0.5) * 3.2e38          # s0 = range / 3 ~ 1e38 > 2^126
    f = (0.85 + 0.01 * torch.arange(16.0))                              # the rows' ranges differ: the 4-bit coded scale
    W[:16, :16] *= f[:, None] * 0.99                                    # of group 0 stays ~1e38 > 2^126 (equal ranges
    W[:16, 0] = -1.6e38 * f                                             # would collapse it to 1.0, quantizer.py:90-92)
    W[:16, 1] = 1.6e38 * f
    ref = O.mxq_quantize(W.numpy())
    s_grp0 = ref["qs2"][0, 0] * (ref["sc2"][:16, 0].astype(np.float32) - ref["qz2"][0, 0])
    assert (s_grp0 > 2.0 ** 126).all()                                  # the case the test is about
    got = packing.unpack(packing.quantize_pack(W.to(dev)))
    for k in KEYS:
        assert np.array_equal(got[k].cpu().numpy(), ref[k]), k
    Wn = torch.randn(32, 128, generator=g) * 0.02
    Wn[21, 3] = Wn[21, 4]                                               # a duplicate: removing it changes no statistic
    base = packing.unpack(packing.quantize_pack(Wn.to(dev)))
    Wn[21, 3] = float("nan")
    nan = packing.unpack(packing.quantize_pack(Wn.to(dev)))
    assert int(nan["codes2"][21, 3]) == 0
    for k in KEYS:
        a, b = base[k].clone(), nan[k].clone()
        if k == "codes2":
            a[21, 3] = b[21, 3] = 0
        assert torch.equal(a, b), k


def test_mxqgpt_driver_api(dev, g1):
    """MXQGPT(layer).add_batch / fasterquant / free as nas_quant calls them (prune.py:385-414)."""
    from mxq_amd.lib.mxqgpt import MXQGPT
    from mxq_amd.quant_linear import QuantLinear
    lin = torch.nn.Linear(256, 64, bias=False).to(dev).half()
    lin.weight.data = torch.from_numpy(g1["W"]).to(dev)
    gpt = MXQGPT(lin)
    gpt.add_batch(torch.from_numpy(g1["x"]).to(dev), None)
    gpt.fasterquant(percdamp=0.01, blocksize=16)
    assert lin.weight.dtype == torch.float16
    assert np.array_equal(lin.weight.data.cpu().numpy().view(np.uint16), g1["w_deq"].view(np.uint16))
    q = gpt.quantizer(1, 2)
    assert np.array_equal(q.codes().cpu().numpy().astype(np.uint8), g1["codes2"][:, 48 + 32:48 + 48])
    assert np.array_equal(q.scale.reshape(-1).cpu().numpy(), g1["scale2"][:, 5])
    assert np.array_equal(gpt.quantizer_4b.scale.reshape(-1).cpu().numpy(), g1["scale4"])
    ql = QuantLinear.from_packed(gpt.packed)
    gpt.free()
    x = torch.from_numpy(g1["x"]).to(dev)
    _check_gemm(ql(x).cpu().numpy(), g1["y32"], "QuantLinear on G1")


def test_quantizer_reference_api_g8(dev):
    """lib.quantizer.Quantizer used exactly as the reference's loop uses it (mxqgpt.py:417-428, :433-436):
    configure -> find_params(W1, weight=True) -> quantize / quantize_dequantize, bit-exact against the reference's own
    Quantizer outputs in golden G8 (2-bit group by group, 4-bit per row), including the constant group / row."""
    from mxq_amd.lib.quantizer import Quantizer
    from tests.conftest import load_golden
    g8 = load_golden("g8_uniform.npz")
    Wf = torch.from_numpy(g8["W"]).to(dev).float()
    N, K = Wf.shape
    for g in range(K // 16):
        blk = Wf[:, 16 * g:16 * g + 16].clone()
        q = Quantizer()
        q.configure(bits=2, perchannel=True, sym=False, qq_scale_bits=4)
        q.find_params(blk, weight=True)
        assert q.scale.shape == (N, 1) and q.zero.shape == (N, 1) and int(q.maxq) == 3 and q.ready()
        assert np.array_equal(q.quantize(blk).cpu().numpy().astype(np.uint8), g8["w2_codes"][:, 16 * g:16 * g + 16])
        assert np.array_equal(q.quantize_dequantize(blk).half().cpu().numpy().view(np.uint16),
                              g8["w2_wdeq"][:, 16 * g:16 * g + 16].view(np.uint16))
        assert np.array_equal(q.quant_scale.reshape(-1).cpu().numpy().astype(np.uint8), g8["w2_sc"][:, g])
        assert np.array_equal(q.zero.reshape(-1).cpu().numpy(), g8["w2_zero"][:, g])
        assert np.array_equal(q.qq_scale.scale.reshape(-1).cpu().numpy(), g8["w2_qs"][:, g])
        assert np.array_equal(q.qq_scale.zero.reshape(-1).cpu().numpy(), g8["w2_qz"][:, g])
        # stored parameters applied to OTHER data: the reference's formula on the device (quantizer.py:14-20)
        other = (blk * 1.7 + 0.003).contiguous()
        ref = torch.clamp(torch.round(other.cpu() / q.scale.cpu().clamp_min(1e-9) + q.zero.cpu()), 0, 3)
        assert torch.equal(q.quantize(other).cpu(), ref)
        assert torch.equal(q.dequantize(q.quantize(other)).cpu(), q.scale.cpu() * (ref - q.zero.cpu()))
    q = Quantizer()
    q.configure(bits=4, perchannel=True, sym=False, qq_scale_bits=4)
    q.find_params(Wf, weight=True)
    assert int(q.maxq) == 15
    assert np.array_equal(q.quantize(Wf).cpu().numpy().astype(np.uint8), g8["w4_codes"])
    assert np.array_equal(q.quantize_dequantize(Wf).half().cpu().numpy().view(np.uint16), g8["w4_wdeq"].view(np.uint16))
    assert np.array_equal(q.zero.cpu().numpy(), g8["w4_zero"])
    assert np.array_equal(q.quant_scale.reshape(-1, 1).cpu().numpy().astype(np.uint8), g8["w4_sc"])
    # a 4-bit arm whose width is not a multiple of the kernel's 64-column block (K/4 of a ragged K): padded internally
    sub = Wf[:, :80].contiguous()
    q2 = Quantizer()
    q2.configure(bits=4, perchannel=True, sym=False, qq_scale_bits=4)
    q2.find_params(sub, weight=True)
    r = O.uniform_quantize(np.concatenate([sub.cpu().numpy(), np.repeat(sub.cpu().numpy()[:, -1:], 48, 1)], 1).astype(np.float32), "w4row")
    assert np.array_equal(q2.quantize(sub).cpu().numpy().astype(np.uint8), r["codes"][:, :80])
    # unsupported configurations refuse instead of approximating
    with pytest.raises(NotImplementedError):
        Quantizer().configure(bits=3, perchannel=True, sym=False, qq_scale_bits=4)
    with pytest.raises(NotImplementedError):
        Quantizer().configure(bits=2, perchannel=True, sym=True, qq_scale_bits=4)
    with pytest.raises(ValueError):
        q.find_params(Wf.cpu(), weight=True)


def test_quantizer_source_cache_survives_recycled_address(dev):
    """find_params on a temporary that is freed, then quantize() of ANOTHER temporary of the same shape
    (the caching allocator hands out the same address, version 0, same dtype) must apply the formula to the tensor
    passed in (quantizer.py:5-20), never return the first tensor's cached codes."""
    from mxq_amd.lib.quantizer import Quantizer
    g = torch.Generator().manual_seed(5)
    A = (torch.randn(64, 16, generator=g) * 0.02).to(dev)
    B = (torch.randn(64, 16, generator=g) * 0.05).to(dev)
    q = Quantizer()
    q.configure(bits=2, perchannel=True, sym=False, qq_scale_bits=4)
    tmp = A.clone()
    addr = tmp.data_ptr()
    q.find_params(tmp, weight=True)
    codes_a = q.quantize(tmp).clone()
    del tmp
    for _ in range(4):                       # without the held reference the very next clone reuses A's block
        other = B.clone()
        want = torch.clamp(torch.round(other / q.scale.clamp_min(1e-9) + q.zero), 0, 3)
        assert torch.equal(q.quantize(other), want)
        assert other.data_ptr() != addr      # the source is held alive by the Quantizer
        del other
    assert not torch.equal(codes_a, want)    # (the two data sets really differ)
    # an identical view of the held source still hits the kernel's codes; a modified source does not
    src = q._src_ref
    assert torch.equal(q.quantize(src[:]), codes_a)
    src.mul_(1.5)
    assert torch.equal(q.quantize(src), torch.clamp(torch.round(src / q.scale.clamp_min(1e-9) + q.zero), 0, 3))


def test_quantizer_reproduces_fasterquant_loop_g1(dev, g1):
    """The reference's fasterquant loop written against lib.quantizer.Quantizer (3 two-bit groups per chunk + one
    4-bit Quantizer over the gathered last-16 columns, mxqgpt.py:404-443) gives G1's codes and fp16 weight."""
    from mxq_amd.lib.quantizer import Quantizer
    W = torch.from_numpy(g1["W"]).to(dev).float()
    W[:, torch.from_numpy(g1["dead"].astype(bool)).to(dev)] = 0        # mxqgpt.py:401-403 (the planted dead column)
    N, K = W.shape
    out = torch.zeros_like(W)
    W4 = torch.cat([W[:, 64 * c + 48:64 * c + 64] for c in range(K // 64)], dim=1).contiguous()
    for c in range(K // 64):
        for g in range(3):
            W1 = W[:, 64 * c + 16 * g:64 * c + 16 * g + 16].contiguous()
            q = Quantizer()
            q.configure(2, perchannel=True, sym=False, qq_scale_bits=4, round_zero=False)
            q.find_params(W1, weight=True)
            out[:, 64 * c + 16 * g:64 * c + 16 * g + 16] = q.quantize_dequantize(W1)
            assert np.array_equal(q.quantize(W1).cpu().numpy().astype(np.uint8), g1["codes2"][:, 48 * c + 16 * g:48 * c + 16 * g + 16])
    q4 = Quantizer()
    q4.configure(4, perchannel=True, sym=False, qq_scale_bits=4, round_zero=False)
    q4.find_params(W4, weight=True)
    d4 = q4.quantize_dequantize(W4)
    for c in range(K // 64):
        out[:, 64 * c + 48:64 * c + 64] = d4[:, 16 * c:16 * c + 16]
    assert np.array_equal(q4.quantize(W4).cpu().numpy().astype(np.uint8), g1["codes4"])
    assert np.array_equal(out.half().cpu().numpy().view(np.uint16), g1["w_deq"].view(np.uint16))


# ----------------------------------------------------------------------------------------
# dequant-GEMM / GEMV
# ----------------------------------------------------------------------------------------
GEMM_KERNELS = ["gemm", "gemm1", "gemm8", "gemm9", "midm", "gemm8h", "gemm8h_split", "gemm8h_slices", "gemm8q_split", "gemm8q_slices", "gemm8n_split", "gemm8n_slices"]      # packing.GEMM_PATHS: every kernel the product library ships


def _packed_case(dev, N, K, seed):
    from mxq_amd import packing
    g = torch.Generator().manual_seed(seed)
    W16 = (torch.randn(N, K, generator=g) * 0.02).half()
    ref = O.mxq_quantize(W16.numpy())
    return packing.quantize_pack(W16.to(dev)), ref["w_deq32"].astype(np.float16), g


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (128, 256, 512), (200, 192, 320), (5, 64, 128), (77, 16, 704),
                                   (256, 4096, 1024), (300, 144, 192), (512, 128, 128)])
@pytest.mark.parametrize("path", GEMM_KERNELS)
def test_gemm_vs_oracle(dev, M, N, K, path):
    """Ragged M (200, 5, 77), N below / not a multiple of the tile, ragged K; every shipped kernel
    (128x128 two-stage, 256x128 wave-specialised with and without its stream-K split) and the automatic choice."""
    from mxq_amd import packing
    p, w16, g = _packed_case(dev, N, K, M * 7 + N)
    x = torch.randn(M, K, generator=g).half()
    y = packing.linear(x.to(dev), p, path=path).cpu().numpy()
    _check_gemm(y, O.linear_ref(x.numpy(), w16), f"{path} {M}x{N}x{K}")


def test_config1_workload_m128_4096_vs_oracle(dev):
    """BASELINE configs[0] exactly: one 4096 x 4096 MXQ Linear, batch 1 x seq 128 (weights seed 0, activations seed 7, as
    bench.py's cpu_baseline leg): every kernel that can run it, and the automatic dispatch, against the ORACLE's
    weight (numpy restatement of fasterquant, pinned to the reference by the golden vectors) -- <= 1e-3 in both norms."""
    from mxq_amd import packing
    N = K = 4096
    M = 128
    W16 = (torch.randn(N, K, generator=torch.Generator().manual_seed(0)) * 0.02).half()
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(7)).half()
    ref = O.mxq_quantize(W16.numpy())
    p = packing.quantize_pack(W16.to(dev))
    got = packing.unpack(p)
    for k in KEYS:
        assert np.array_equal(got[k].cpu().numpy(), ref[k]), k             # the integer unpack, bit for bit
    yref = O.linear_ref(x.numpy(), ref["w_deq32"].astype(np.float16))
    for path in ["auto"] + GEMM_KERNELS:
        _check_gemm(packing.linear(x.to(dev), p, path=path).cpu().numpy(), yref, f"configs[0] {path}")


def test_gemm_integer_exact_layout(dev):
    """Small-integer weights and activations make every partial sum exact in fp32/fp16, so
    any fragment / swizzle / k-ordering mistake shows up as an exact mismatch (asymmetric
    data: cdna guide section 3 'A=I-check with ASYMMETRIC B')."""
    from mxq_amd import packing
    N, K, M = 256, 256, 160
    rng = np.random.default_rng(5)
    p = O.mxq_quantize(np.zeros((N, K), np.float16))
    p["codes2"] = rng.integers(0, 4, (N, K // 64 * 48), dtype=np.uint8)
    p["codes4"] = rng.integers(0, 16, (N, K // 4), dtype=np.uint8)
    p["sc2"][:] = 1; p["qs2"][:] = 1.0; p["qz2"][:] = 0.0; p["zero2"][:] = 1.0     # w = q - 1
    p["sc4"][:] = 1; p["qs4"][:] = 1.0; p["qz4"][:] = 0.0; p["zero4"][:] = 7.0     # w = q - 7
    w = O.mxq_dequant(p)
    assert np.array_equal(w, np.round(w))
    x = rng.integers(-2, 3, (M, K)).astype(np.float16)
    pk = packing.pack_codes(_to_dev(p, dev), N, K)
    yref = x.astype(np.float32) @ w.T
    assert np.abs(yref).max() < 2048
    for path, rows in [(k, M) for k in GEMM_KERNELS] + [("gemv", 3)]:
        y = packing.linear(torch.from_numpy(x[:rows]).to(dev), pk, path=path).cpu().numpy().astype(np.float32)
        assert np.array_equal(y, yref[:rows]), path


@pytest.mark.parametrize("M,N,K", [(256, 2048, 4096),     # 16 tiles < 256 CUs: every K-step is stream-K, 16 contributors per tile
                                   (300, 2048, 4096),     # ragged M edge in the partial slots
                                   (512, 1024, 8192),     # units straddle two tiles
                                   (1024, 4224, 2048),    # 132 tiles: XCDs 0-3 hold 17 tail tiles, 4-7 hold 16
                                   (100, 1152, 8192),     # 9 tiles: XCD 0 holds two, the others one; half-empty 256-row tile
                                   (2048, 5120, 1024)])   # 320 tiles = one full round + a 64-tile tail
@pytest.mark.parametrize("sk", ["gemm9", "gemm8h_split", "gemm8q_split", "gemm8n_split"])
def test_gemm_stream_k_tail(dev, M, N, K, sk):
    """csrc/gemm8.hip: tiles beyond the last full round of CUs are split along K over all CUs and
    reduced through the workspace.  Checks (1) against the oracle matmul, (2) that the workspace
    counters are left zeroed, (3) run-to-run bit-determinism, (4) agreement with the 128x128-tile
    kernel (no split) to summation-order rounding."""
    from mxq_amd import packing
    p, w16, g = _packed_case(dev, N, K, M + N + K)
    x = torch.randn(M, K, generator=g).half()
    xd = x.to(dev)
    y = packing.linear(xd, p, path=sk)
    _check_gemm(y.cpu().numpy(), O.linear_ref(x.numpy(), w16), f"{sk} {M}x{N}x{K}")
    ws = packing.gemm_workspace(xd.device)
    assert int(ws[:65504].view(torch.int32).abs().sum().item()) == 0 and int(ws[65520:65536].view(torch.int32).abs().sum().item()) == 0
    for _ in range(3):
        assert torch.equal(packing.linear(xd, p, path=sk), y)
    y5 = packing.linear(xd, p, path="gemm1")
    assert ((y.float() - y5.float()).abs().max() / y5.float().abs().max()).item() <= 1e-3
    # the default dispatch is the same kernel, splitting only where it pays: equal up to summation order
    yd = packing.linear(xd, p, path="gemm")
    assert ((y.float() - yd.float()).abs().max() / y5.float().abs().max()).item() <= 1e-3
    assert torch.equal(packing.linear(xd, p, path="gemm8"), yd)
    if sk == "gemm8h_split":      # the 128-token build of the same kernel, splitting only where it pays
        yh = packing.linear(xd, p, path="gemm8h")
        assert ((y.float() - yh.float()).abs().max() / y5.float().abs().max()).item() <= 1e-3
        assert int(ws[:65504].view(torch.int32).abs().sum().item()) == 0 and int(ws[65520:65536].view(torch.int32).abs().sum().item()) == 0


@pytest.mark.parametrize("M,N,K", [(640, 11008, 4096),     # 258 tiles: a 2-tile tail on 256 CUs (Llama gate/up at 640-768 tokens)
                                   (768, 11008, 4096),
                                   (1536, 11008, 4096),    # 516 tiles: two full rounds + a 4-tile tail
                                   (256, 34176, 4096),     # 267 tiles: an 11-tile tail, XCDs 0-2 hold two tail tiles, 3-7 one
                                   (2048, 11008, 4096)])   # 688 tiles: the 176-tile tail the dispatch splits since round 4
def test_gemm_small_tail_stream_k_through_the_dispatch(dev, M, N, K):
    """The PRODUCT dispatch's stream-K branches (csrc/gemm8.hip launch8: a tail of a few tiles split over fewer units per
    XCD; the large tail of gate/up at 2048 tokens) -- reached by `auto` / `gemm8`, not only by the forced-split test
    path: against the fp32 product on the dequantised weight, twice with identical bits, counters left zero."""
    from mxq_amd import packing
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
    x = torch.randn(M, K, generator=g, device=dev).half()
    wd = packing.dequant(p)
    ws = packing.gemm_workspace(torch.device(dev))
    for path in ("gemm8", "auto"):
        y = packing.linear(x, p, path=path)
        worst = 0.0
        for n0 in range(0, N, 8192):                   # fp32 reference in column slabs
            r = x.float() @ wd[n0:n0 + 8192].float().t()
            worst = max(worst, ((y[:, n0:n0 + 8192].float() - r).abs().max() / r.abs().max()).item())
        assert worst <= REL_TOL, (path, worst)
        assert torch.equal(packing.linear(x, p, path=path), y), path
        assert int(ws[:65504].view(torch.int32).abs().sum().item()) == 0 and int(ws[65520:65536].view(torch.int32).abs().sum().item()) == 0, path
    # the split changes the summation order of the tail tiles only: whole-tile launch within rounding
    yw = packing.linear(x, p, path="whole")
    assert ((y.float() - yw.float()).abs().max() / yw.float().abs().max()).item() <= REL_TOL


@pytest.mark.parametrize("sk", ["gemm9", "gemm8h_split", "gemm8q_split", "gemm8n_split"])
def test_stream_k_partition_fuzz(dev, sk):
    """Random (tokens, out, in) shapes through the forced stream-K schedule against the single-tile kernel:
    exercises unit ranges that start / end anywhere inside tiles, XCDs with unequal tail lengths, units with
    one, two and many segments, ragged M and N edges."""
    from mxq_amd import packing
    rng = np.random.default_rng(2026)
    g = torch.Generator(device=dev).manual_seed(9)
    for _ in range(24):
        M = int(rng.integers(5, 1500))
        N = 16 * int(rng.integers(1, 200))
        K = 64 * int(rng.choice([2, 3, 8, 16, 33, 64, 100, 172]))
        W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
        p = packing.quantize_pack(W)
        x = torch.randn(M, K, generator=g, device=dev).half()
        y = packing.linear(x, p, path=sk).float()
        r = (x.float() @ packing.dequant(p).float().t())
        err = ((y - r).abs().max() / r.abs().max()).item()
        assert err <= REL_TOL, (M, N, K, err)
        assert torch.equal(packing.linear(x, p, path=sk).float(), y), (M, N, K)
    ws = packing.gemm_workspace(torch.device(dev))
    assert int(ws[:65504].view(torch.int32).abs().sum().item()) == 0 and int(ws[65520:65536].view(torch.int32).abs().sum().item()) == 0


@pytest.mark.parametrize("M,N,K", [(192, 11008, 4096),     # 2 x 86 = 172 tiles of 128 x 128: Llama gate/up, 129-256 tokens
                                   (128, 11008, 4096),     # 86 tiles: 3 CUs per tile
                                   (384, 4096, 4096),      # 96 tiles
                                   (512, 4096, 11008),     # 128 tiles, 172 K-steps each
                                   (300, 4224, 2048),      # 99 tiles, ragged token edge, XCDs hold 13 / 12 tiles
                                   (129, 4096, 1024),      # 64 tiles: the last launch in slices mode ...
                                   (129, 4224, 1024),      # ... 66: the first in stream-K mode
                                   (128, 4096, 4096),      # BASELINE configs[0]'s shape: 32 tiles, 8 slices each
                                   (256, 4096, 11008),     # 64 tiles x 4 slices of 43 K-steps (long K: the 128 x 128 tile)
                                   (256, 4096, 4096),      # short K: 128 tiles of 128 x 64, stream-K
                                   (200, 2048, 6144),      # ... 64 tiles of 128 x 64, slices, ragged token edge
                                   (100, 2048, 1024),      # 16 tiles, one ragged; S = 16 clamped to 4 (4 K-steps per slice)
                                   (65, 1040, 256),        # 9 tiles, ragged both ways, K too short to slice: whole tiles
                                   (48, 11008, 4096),      # <= 64 tokens: 86 tiles of 64 x 128 -> the 64-token build, stream-K
                                   (64, 4096, 4096),       # ... 32 tiles: mid-M kernel (mixed), 64-token build in slices mode (uniform)
                                   (21, 11008, 1024),      # just above the skinny kernel's limit for the 45-M-element weights
                                   (256, 11264, 1024),     # 176 tiles: stream-K mode's upper edge ...
                                   (256, 11392, 1024)])    # ... and 178: the 256-token tile again
@pytest.mark.parametrize("layout", ["mixed", "mixedc", "w2g16", "w4row"])
def test_half_height_tile_through_the_dispatch(dev, M, N, K, layout):
    """capi.hip gemm8h_mode / gemm8q_mode: beyond 64 tokens, launches of up to 176 tiles of 128 x 128 run the fused kernel's
    128-token build (gemm8h.hip) -- up to 64 tiles in slices mode (K slices + combine launch), 65..176 in one launch with
    stream-K over the idle CUs; up to 64 tokens the 64-token build (gemm8q.hip) where it wins -- every weight layout, through the product dispatch: against the fp32 product on the bit-exact dequantised
    weight, twice with identical bits, counters left zero; for the mixed layout the dispatch's result IS the explicit path's."""
    from mxq_amd import packing
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
    if layout in ("mixed", "mixedc"):
        p = packing.quantize_pack(W)
        if layout == "mixedc":
            p = packing.compact(p)
    else:
        p = packing.quantize_pack_uniform(W, layout)
    x = torch.randn(M, K, generator=g, device=dev).half()
    wd = packing.dequant(p) if layout in ("mixed", "mixedc") else packing.expand_uniform(p, codes=False)[0]
    ws = packing.gemm_workspace(torch.device(dev))
    y = packing.linear_layout(x, p, path="auto")
    r = x.float() @ wd.float().t()
    assert ((y.float() - r).abs().max() / r.abs().max()).item() <= REL_TOL
    assert torch.equal(packing.linear_layout(x, p, path="auto"), y)
    assert int(ws[:65504].view(torch.int32).abs().sum().item()) == 0 and int(ws[65520:65536].view(torch.int32).abs().sum().item()) == 0
    tiles = -(-M // 128) * -(-N // 128)
    if layout == "mixed" and M <= 64:
        if 65 <= -(-N // 128) <= 176 and M > 20:
            yq = packing.linear(x, p, path="gemm8q_split")      # the same kernel with its tail ALWAYS split: the dispatch
            if K >= 4096:                                        # splits where that pays (it does from ~32 K-steps per tile on)
                assert torch.equal(yq, y)
            assert ((y.float() - yq.float()).abs().max() / r.abs().max()).item() <= REL_TOL
    elif layout == "mixed" and tiles <= 64 and K <= 6144:       # short K: the 128 x 64 tile (half the partial-tile bytes)
        t64 = -(-M // 128) * -(-N // 64)
        yn = packing.linear(x, p, path="gemm8n_slices" if t64 <= 64 else "gemm8n_split")
        if t64 <= 64 or K >= 4096:                               # (stream-K: the dispatch splits only where that pays)
            assert torch.equal(yn, y)
        assert ((y.float() - yn.float()).abs().max() / r.abs().max()).item() <= REL_TOL
    elif layout == "mixed" and tiles <= 176:
        assert torch.equal(packing.linear(x, p, path="gemm8h_slices" if tiles <= 64 else "gemm8h"), y)
        ym = packing.linear(x, p, path="midm")           # the neighbour it replaced here: equal up to summation order
        assert ((y.float() - ym.float()).abs().max() / r.abs().max()).item() <= REL_TOL


def test_linear_auto_c_entry_is_correct_on_every_path_it_dispatches_to(dev):
    """include/mxq_hip.h: mxq_linear_f16_auto -- ONE C call for the whole dispatch.  From mxq_hoist_min_tokens() tokens on
    and given a scratch buffer it runs the hoisted-dequant mode (bit-identical to mxq_linear_f16_hoisted and to the fused
    kernel); without a scratch, or below the threshold, the _ws dispatch; every layout."""
    from mxq_amd import _lib, packing
    lib = _lib.load()
    assert lib.mxq_hoist_min_tokens() == packing.HOIST_MIN_TOKENS
    g = torch.Generator(device=dev).manual_seed(4)
    N, K = 512, 1024
    W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
    ws = packing.gemm_workspace(torch.device(dev))
    scratch = torch.empty(lib.mxq_hoist_scratch_bytes(N, K), dtype=torch.uint8, device=dev)
    cases = [(packing.quantize_pack(W), 0), (packing.quantize_pack(W, compact_meta=True), 3),
             (packing.quantize_pack_uniform(W, "w2g16"), 1), (packing.quantize_pack_uniform(W, "w4row"), 2)]
    for M in (4096, 4100, 100, 3):
        x = torch.randn(M, K, generator=g, device=dev).half()
        for p, layout in cases:
            args = (x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr())
            y = torch.empty(M, N, dtype=torch.float16, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.mxq_linear_f16_auto(*args, y.data_ptr(), M, N, K, layout, ws.data_ptr(), ws.numel(),
                                               scratch.data_ptr(), scratch.numel(), st), "auto")
            y2 = torch.empty_like(y)
            _lib.check(lib.mxq_linear_f16_auto(*args, y2.data_ptr(), M, N, K, layout, ws.data_ptr(), ws.numel(), None, 0, st), "auto")
            ref = x.float() @ (packing.dequant(p) if layout in (0, 3) else packing.expand_uniform(p, codes=False)[0]).float().t()
            for got in (y, y2):
                assert ((got.float() - ref).abs().max() / ref.abs().max()).item() <= REL_TOL, (M, layout)
            if M >= 4096:
                yh = torch.empty_like(y)
                _lib.check(lib.mxq_linear_f16_hoisted(*args, yh.data_ptr(), M, N, K, layout, scratch.data_ptr(),
                                                      scratch.numel(), st), "hoisted")
                assert torch.equal(y, yh) and torch.equal(y, y2), (M, layout)     # hoisted == fused, bit for bit
    # a scratch that is too small never fails: the fused kernel runs
    p, layout = cases[0]
    x = torch.randn(4096, K, generator=g, device=dev).half()
    y = torch.empty(4096, N, dtype=torch.float16, device=dev)
    assert lib.mxq_linear_f16_auto(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), y.data_ptr(), 4096, N, K, 0,
                                   ws.data_ptr(), ws.numel(), scratch.data_ptr(), 1024,
                                   torch.cuda.current_stream().cuda_stream) == 0
    assert torch.equal(y, packing.linear(x, p, path="fused"))


@pytest.mark.parametrize("M", [3, 40, 64, 128, 256, 300])
def test_workspace_free_dispatch_every_layout(dev, M):
    """mxq_linear_f16_layout_ws with workspace == NULL (and mxq_linear_f16): ONE workspace-free schedule for every layout
    (capi.hip linear_noworkspace; ADVICE r4: compact metadata at 128 tokens used to take another road than exact metadata)
    -- correct against the fp32 product on the kernel-dequantised weight at token counts either side of every threshold."""
    from mxq_amd import _lib, packing
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(M)
    N, K = 384, 1024
    W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
    x = torch.randn(M, K, generator=g, device=dev).half()
    st = torch.cuda.current_stream().cuda_stream
    for p, layout in [(packing.quantize_pack(W), 0), (packing.quantize_pack(W, compact_meta=True), 3),
                      (packing.quantize_pack_uniform(W, "w2g16"), 1), (packing.quantize_pack_uniform(W, "w4row"), 2)]:
        ref = x.float() @ (packing.dequant(p) if layout in (0, 3) else packing.expand_uniform(p, codes=False)[0]).float().t()
        y = torch.empty(M, N, dtype=torch.float16, device=dev)
        _lib.check(lib.mxq_linear_f16_layout_ws(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), y.data_ptr(), M, N, K,
                                                layout, None, 0, st), "layout_ws without a workspace")
        assert ((y.float() - ref).abs().max() / ref.abs().max()).item() <= REL_TOL, (M, layout)
        if layout == 0:
            y2 = torch.empty_like(y)
            _lib.check(lib.mxq_linear_f16(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), y2.data_ptr(), M, N, K, st), "linear")
            assert torch.equal(y, y2)
    assert lib.mxq_linear_workspace_need(M, N, K, 0, 1) <= lib.mxq_gemm_workspace_bytes()


@pytest.mark.parametrize("how", ["poll", "next_call"])
@pytest.mark.parametrize("M,N,K,build", [(512, 1024, 8192, "gemm8"),     # owner protocol (units straddle two tiles)
                                          (256, 2048, 4096, "gemm8"),     # 16 contributors per tile: all-contributors reduction
                                          (384, 4096, 4096, "gemm8h")])   # the 128-token build
def test_stream_k_wait_expiry_is_visible(dev, M, N, K, build, how):
    """A stream-K wait that gives up must not pass for a result.  Fault injection through the
    PROFILING library (`mxq_prof_<build>_skwithhold_f16` never counts unit 0's parked pieces and gives up after 4096 polls):
    the launch ends, the workspace's status words are set, the starved tiles are NaN.  "poll": `packing.workspace_status`
    raises and re-zeroes the head.  "next_call" (round 6): NOBODY polls -- the kernel has also written its
    status into the workspace's pinned host mailbox, and the next product call on that workspace raises by itself.  Either
    way the launch after that is correct."""
    import ctypes
    import os
    from mxq_amd import _lib, packing
    prof = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmxq_hip_prof.so")
    if not os.path.exists(prof):
        pytest.skip("libmxq_hip_prof.so not built (make -C mxq_amd/csrc prof)")
    plib = ctypes.CDLL(prof)
    fn = getattr(plib, f"mxq_prof_{build}_skwithhold_f16")
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    p, w16, g = _packed_case(dev, N, K, M + N + K + 1)
    x = torch.randn(M, K, generator=g).half()
    xd = x.to(dev)
    path = "gemm9" if build == "gemm8" else "gemm8h_split"
    good = packing.linear(xd, p, path=path)
    packing.workspace_status(xd.device)                         # nothing flagged by a healthy launch
    ws = packing.gemm_workspace(xd.device)
    y = torch.zeros(M, N, dtype=torch.float16, device=dev)
    rc = fn(xd.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), y.data_ptr(), M, N, K, ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    st = (ctypes.c_int * 4)()
    assert _lib.load().mxq_workspace_status(ws.data_ptr(), ws.numel(), ctypes.cast(st, ctypes.c_void_p), None) == 0
    assert st[0] in (1, 2) and st[2] >= 0, list(st)
    bad = torch.isnan(y.float())
    assert bad.any(), "a starved tile must be poisoned"
    ok = ~bad
    assert torch.equal(y[ok], good[ok]) or ((y[ok].float() - good[ok].float()).abs().max() / good.float().abs().max()).item() <= 1e-3
    with pytest.raises(RuntimeError, match="stream-K wait expired"):
        if how == "poll":
            packing.workspace_status(xd.device)
        else:
            packing.linear(xd, p, path=path)                    # the product's own next call: no status query anywhere
    assert int(ws[:65504].view(torch.int32).abs().sum().item()) == 0          # head re-zeroed (its last bytes: mailbox address, status)
    assert int(ws[65520:65536].view(torch.int32).abs().sum().item()) == 0
    again = packing.linear(xd, p, path=path)
    assert torch.equal(again, good)
    packing.workspace_status(xd.device)


def test_stream_k_launch_beside_an_lds_heavy_kernel(dev):
    """a stream-K launch whose workgroups CANNOT all be resident at once.  Workgroups of an unrelated
    kernel (profiling library: `mxq_prof_occupy`, 100 / 20 KB of LDS each, 3 ms, every wave leaves on time) sit on some or on
    all CUs on a second stream while a gate/up-shaped launch (2048 x 11008 x 4096: 176 tail tiles split over all 256
    workgroups, 144 KB of LDS each) starts: part of its workgroups are dispatched only when others -- or the occupier -- leave,
    and owners wait for parked pieces meanwhile.  The result must be the undisturbed launch's bit for bit, the status 0.
    (Workgroups are dispatched in index order and an owner waits only for lower-numbered units: the protocol needs the
    launch's workgroups to be dispatched EVENTUALLY, not all at once; the all-contributors reduction of the second shape waits
    in both directions and relies on the foreign kernel ending -- every wait is bounded either way.)"""
    import ctypes
    import os
    import time
    from mxq_amd import _lib, packing
    prof = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmxq_hip_prof.so")
    if not os.path.exists(prof):
        pytest.skip("libmxq_hip_prof.so not built (make -C mxq_amd/csrc prof)")
    plib = ctypes.CDLL(prof)
    plib.mxq_prof_occupy.restype = ctypes.c_int
    plib.mxq_prof_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # gate/up at 2048 tokens: the owner protocol (an owner waits only for LOWER-numbered units, which are dispatched first);
    # 256 tokens x 2048 channels: 16 contributors per tile, the all-contributors reduction (any contributor may wait for any other)
    for M, N, K in ((2048, 11008, 4096), (256, 2048, 4096)):
        g = torch.Generator(device=dev).manual_seed(11)
        p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
        x = torch.randn(M, K, generator=g, device=dev).half()
        whole = packing.linear(x, p, path="whole")
        path = "auto" if M == 2048 else "gemm9"                 # the product entry (stream-K tail at this shape) / the tail always split
        calm = packing.linear(x, p, path=path)
        e0.record(); packing.linear(x, p, path=path); e1.record()
        torch.cuda.synchronize()
        calm_ms = e0.elapsed_time(e1)
        packing.workspace_status(dev)
        # 64 CUs taken (the launch's last 64 workgroups start when the first ones leave: about twice the calm time) / every CU
        # taken (nothing starts before the occupier leaves) / LDS left on every CU but not enough for a 144-KB workgroup
        for lds_kb, grid, held in ((100, 64, 1.4), (100, 256, 5.0), (20, 1024, 5.0)):
            assert plib.mxq_prof_occupy(grid, lds_kb * 1024, 3000, sink.data_ptr(), side.cuda_stream) == 0
            time.sleep(0.0005)                                  # the occupier is on the chip before the GEMM is launched
            e0.record()
            y = packing.linear(x, p, path=path)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            assert torch.equal(y, calm), (M, lds_kb, grid)
            assert ((y.float() - whole.float()).abs().max() / whole.float().abs().max()).item() <= REL_TOL
            packing.workspace_status(dev)                       # status 0: no wait gave up
            if M == 2048:
                assert ms > held * calm_ms, (f"the occupier ({grid} x {lds_kb} KB) did not hold the launch back ({ms:.3f} vs {calm_ms:.3f} ms): "
                                             "the test proves nothing")


def test_stream_k_gemm_in_graphs_and_on_two_streams(dev):
    """The stream-K GEMM leaves its workspace counters zeroed, so a captured launch can be replayed; and
    launches on different streams use different workspaces, so they may overlap."""
    from mxq_amd import packing
    p, w16, g = _packed_case(dev, 2048, 4096, 5)
    x = torch.randn(512, 4096, generator=g).half().to(dev)
    ref = packing.linear(x, p, path="gemm9")                      # warm-up: allocates this stream's workspace
    out = torch.empty_like(ref)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        packing.linear(x, p, out=out, path="gemm9")
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for s in (s1, s2):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            outs.append([packing.linear(x, p, path="gemm9") for _ in range(8)])
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for lst in outs for o in lst)
    keys = {k for k in packing._WORKSPACES if k[0] == x.device.index}
    assert len(keys) >= 3                                          # default stream + two side streams
    # every captured graph owns its workspace (keyed by the capture sequence's id): ONE memset of the counter head is
    # recorded per graph, and whatever the buffer holds before a replay does not matter
    ckeys = [k for k in packing._WORKSPACES if k[0] == x.device.index and len(k) >= 3 and k[1] == "capture"]
    assert len(ckeys) >= 1
    for k in ckeys:
        packing._WORKSPACES[k][:65536].fill_(7)
    for _ in range(2):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("M,N,K", [(128, 4096, 4096),      # slices mode: two launches (K slices, combine) per call
                                   (192, 11008, 4096),     # the 128-token build, stream-K in one launch
                                   (48, 11008, 4096)])     # the 64-token build, stream-K
def test_small_tile_dispatch_in_a_graph(dev, M, N, K):
    """The dispatch's small-tile paths under hipGraph capture: three calls per graph (the workspace's slabs / counters are
    reused from call to call inside the graph), replayed with the workspace overwritten in between -- equal to the eager
    result every time."""
    from mxq_amd import packing
    g = torch.Generator(device=dev).manual_seed(M + N)
    p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half())
    xs = [torch.randn(M, K, generator=g, device=dev).half() for _ in range(3)]
    refs = [packing.linear(x, p, path="auto") for x in xs]
    outs = [torch.empty_like(r) for r in refs]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for x, o in zip(xs, outs):
            packing.linear(x, p, out=o, path="auto")
    ckeys = [k for k in packing._WORKSPACES if len(k) >= 3 and k[1] == "capture"]
    for it in range(3):
        for o in outs:
            o.zero_()
        if it:
            packing._WORKSPACES[ckeys[-1]].fill_(it)      # whatever the buffer holds before a replay must not matter
        graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(o, r) for o, r in zip(outs, refs)), it


def test_two_stream_k_graphs_replayed_concurrently(dev):
    """Two graphs with stream-K launches (partial tiles + counters in the workspace), replayed AT THE SAME TIME on two
    streams, many times: each graph has its own workspace, so neither corrupts the other's partial sums (round 3 shared
    one buffer per device among all captured launches)."""
    from mxq_amd import packing
    pa, _, g = _packed_case(dev, 2048, 4096, 21)
    pb, _, _ = _packed_case(dev, 1152, 8192, 22)
    xa = torch.randn(512, 4096, generator=g).half().to(dev)
    xb = torch.randn(100, 8192, generator=g).half().to(dev)
    refa, refb = packing.linear(xa, pa, path="gemm9"), packing.linear(xb, pb, path="gemm9")
    outa, outb = torch.empty_like(refa), torch.empty_like(refb)
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        for _ in range(4):
            packing.linear(xa, pa, out=outa, path="gemm9")
    with torch.cuda.graph(gb):
        for _ in range(4):
            packing.linear(xb, pb, out=outb, path="gemm9")
    ckeys = [k for k in packing._WORKSPACES if len(k) >= 3 and k[1] == "capture"]
    assert len({packing._WORKSPACES[k].data_ptr() for k in ckeys}) >= 2
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(10):
        outa.zero_(); outb.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            ga.replay()
        with torch.cuda.stream(s2):
            gb.replay()
        torch.cuda.synchronize()
        assert torch.equal(outa, refa) and torch.equal(outb, refb)


def test_linear_empty_and_nonfinite_inputs(dev):
    """Empty token dimension returns an empty result without a launch; a NaN / inf activation poisons exactly
    its own token row, on the GEMV, the single-tile GEMM and the stream-K GEMM alike (no cross-row leakage
    through the shared accumulators or the stream-K workspace)."""
    from mxq_amd import packing
    p, w16, g = _packed_case(dev, 256, 4096, 11)
    assert packing.linear(torch.empty(0, 4096, dtype=torch.float16, device=dev), p).shape == (0, 256)
    assert packing.linear(torch.empty(2, 0, 4096, dtype=torch.float16, device=dev), p).shape == (2, 0, 256)
    for M, path in ((3, "gemv"), (100, "gemm1"), (300, "gemm9"), (700, "gemm8"), (100, "midm"), (150, "auto"), (600, "midm")):
        x = torch.randn(M, 4096, generator=g).half()
        clean = packing.linear(x.to(dev), p, path=path)
        x[1, 7] = float("nan")
        x[M - 1, 4000] = float("inf")
        y = packing.linear(x.to(dev), p, path=path)
        assert torch.isnan(y[1]).all()
        assert not torch.isfinite(y[M - 1]).any()
        keep = [i for i in range(M) if i not in (1, M - 1)]
        assert torch.equal(y[keep], clean[keep]) and torch.isfinite(y[keep]).all()


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K", [(64, 256), (256, 704), (4096, 4096)])
def test_gemv_vs_oracle(dev, M, N, K):
    from mxq_amd import packing
    p, w16, g = _packed_case(dev, N, K, M + N + K)
    x = torch.randn(M, K, generator=g).half()
    y = packing.linear(x.to(dev), p, path="gemv").cpu().numpy()
    _check_gemm(y, O.linear_ref(x.numpy(), w16), f"gemv {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(1, 16, 64), (4, 16, 64), (1, 48, 192), (3, 6144, 64), (2, 32, 1088), (4, 4096, 11008),
                                   (1, 22016, 4096), (4, 12288, 4096)])
@pytest.mark.parametrize("compact", [False, True])
def test_gemv_pipeline_edge_shapes(dev, M, N, K, compact):
    """Shapes that stress the GEMV's control structure rather than its arithmetic: one chunk (K = 64: every wave but the
    first walks only out-of-range tiles), ragged tiles (K = 192: 3 chunks; K = 1088: 17), more staging steps than the
    two that are hoisted (4 tokens x K = 11008 on 512 threads), every workgroup size (N = 16 .. 22016), both metadata
    modes.  Reference: fp32 product on the kernel-dequantised weight (bit-exact vs the oracle in other tests)."""
    from mxq_amd import packing
    g = torch.Generator(device=dev).manual_seed(M * 5 + N + K)
    p = packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half(), compact_meta=compact)
    wd = packing.dequant(p).float()
    x = torch.randn(M, K, generator=g, device=dev).half()
    y = packing.linear(x, p, path="gemv").float()
    ref = x.float() @ wd.t()
    assert ((y - ref).abs().max() / ref.abs().max()).item() <= REL_TOL
    x[M - 1, K - 1] = float("nan")          # a NaN stays in its own token's row
    y2 = packing.linear(x, p, path="gemv").float()
    assert torch.isnan(y2[M - 1]).all() and (M == 1 or torch.equal(y2[: M - 1], y[: M - 1]))


@pytest.mark.parametrize("V", [32000, 1000, 7])
def test_lmhead_argmax_matches_the_torch_head(dev, V):
    """csrc/decode_ops.hip: final RMSNorm + fp16 lm_head + greedy argmax in one launch pair against the torch ops it
    replaces (same roundings: fp16(x * inv) * g, fp32 dot -> fp16 logit, lowest index on ties).  The two may only
    differ where the top two fp16 logits are adjacent (another summation order): then either is accepted."""
    from mxq_amd import _lib
    K = 4096
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(V)
    w = (torch.randn(V, K, generator=g, device=dev) * 0.02).half()
    gw = (1.0 + 0.1 * torch.randn(K, generator=g, device=dev)).half()
    part = torch.empty(2 * 1024, dtype=torch.float32, device=dev)
    tok = torch.zeros(1, dtype=torch.int64, device=dev)
    for it in range(6):
        h = torch.randn(1, K, generator=g, device=dev).half() * (0.5 + it)
        if it == 5:                                       # an exact tie: two identical rows -> the lower index
            w[V - 1] = w[V // 2]
        _lib.check(lib.mxq_lmhead_argmax_f16(h.data_ptr(), gw.data_ptr(), 1e-5, w.data_ptr(), V, K, part.data_ptr(), 1024,
                                             tok.data_ptr(), torch.cuda.current_stream().cuda_stream), "mxq_lmhead_argmax_f16")
        hf = h.float()
        xn = (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + 1e-5)).half() * gw
        logits = (xn.float() @ w.float().t()).half()[0]
        want = int(logits.argmax())
        got = int(tok.item())
        assert 0 <= got < V
        if got != want:
            top = logits.float().topk(2).values
            assert logits[got].float() >= top[1] and (top[0] - top[1]) <= top[0].abs() * 2 ** -9, (got, want)
        if it == 5:
            assert got != V - 1


@pytest.mark.parametrize("pos", [0, 1, 15, 16, 63, 64, 65, 130, 255])
def test_attn_decode_kernel_across_the_prefetched_rows(dev, pos):
    """csrc/decode_ops.hip loads the first 64 K / V rows of the cache speculatively together with the position and
    walks the rest with ordinary loads; the new key / value take part from LDS.  Positions on both sides of every
    boundary (0, 16-row V groups, 64, deep contexts, the cache's last row) against a torch restatement."""
    from mxq_amd import _lib
    heads, hd, ctx = 4, 128, 256
    g = torch.Generator(device=dev).manual_seed(pos)
    qkv = torch.randn(3 * heads * hd, generator=g, device=dev).half()
    kc = torch.randn(heads, ctx, hd, generator=g, device=dev).half()
    vc = torch.randn(heads, ctx, hd, generator=g, device=dev).half()
    kc0, vc0 = kc.clone(), vc.clone()
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    ang = torch.arange(ctx, device=dev).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    posd = torch.tensor([pos], dtype=torch.int64, device=dev)
    out = torch.empty(heads * hd, dtype=torch.float16, device=dev)
    lib = _lib.load()
    _lib.check(lib.mxq_attn_decode_f16(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), posd.data_ptr(), cos.data_ptr(),
                                       sin.data_ptr(), out.data_ptr(), heads, hd, ctx,
                                       torch.cuda.current_stream().cuda_stream), "mxq_attn_decode_f16")
    q, k, v = (qkv[j * heads * hd:(j + 1) * heads * hd].view(heads, hd).float() for j in range(3))

    def rope(t):
        t1, t2 = t[:, : hd // 2], t[:, hd // 2:]
        return torch.cat([t1 * cos[pos] - t2 * sin[pos], t2 * cos[pos] + t1 * sin[pos]], -1)
    qr, kr = rope(q).half().float(), rope(k).half()
    kref, vref = kc0.clone(), vc0.clone()
    kref[:, pos], vref[:, pos] = kr, v.half()
    assert torch.equal(kc, kref) and torch.equal(vc, vref)            # exactly one row appended, nothing else touched
    att = (qr[:, None, :] @ kref[:, : pos + 1].float().transpose(1, 2) / hd ** 0.5).half().float().softmax(-1)
    want = (att.half().float() @ vref[:, : pos + 1].float()).reshape(-1)
    assert ((out.float() - want).abs().max() / want.abs().max()).item() < 4e-3
    # round 5: the position's rotary row gathered once per token (mxq_rope_row_f32) + the kernel variant that takes the row
    # instead of the tables: same arithmetic, identical bits -- output and both caches
    row = torch.empty(hd, dtype=torch.float32, device=dev)
    out2 = torch.empty_like(out)
    kc2, vc2 = kc0.clone(), vc0.clone()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.mxq_rope_row_f32(posd.data_ptr(), cos.data_ptr(), sin.data_ptr(), row.data_ptr(), hd // 2, ctx, st), "rope_row")
    assert torch.equal(row[: hd // 2], cos[pos]) and torch.equal(row[hd // 2:], sin[pos])
    _lib.check(lib.mxq_attn_decode_row_f16(qkv.data_ptr(), kc2.data_ptr(), vc2.data_ptr(), posd.data_ptr(), row.data_ptr(),
                                           out2.data_ptr(), heads, hd, ctx, st), "mxq_attn_decode_row_f16")
    assert torch.equal(out2, out) and torch.equal(kc2, kc) and torch.equal(vc2, vc)


@pytest.mark.parametrize("pos,splits,heads,ctx", [(p, s, 4, 1024) for s in (8, 3) for p in (0, 63, 127, 128, 129, 191, 192, 300, 511, 512, 777, 1023)]
                         + [(p, 16, 32, 2048) for p in (128, 1100, 1920, 2047)])      # the shipped long-cache configuration: 16 splits, 32 heads
def test_attn_decode_split_over_keys(dev, pos, splits, heads, ctx):
    """Long contexts: a head's keys split over up to `splits` workgroups, partial softmax parts merged by the head's last
    arriver (csrc/decode_ops.hip attn_decode_split_kernel).  Up to 128 keys the result is the one-workgroup kernel's bit for
    bit; beyond, against a torch restatement (fp32 softmax over fp16-rounded scores).  Exactly one cache row appended,
    the arrival counters left zeroed, launch after launch on the same workspace."""
    from mxq_amd import _lib
    hd = 128
    g = torch.Generator(device=dev).manual_seed(pos * 7 + splits)
    qkv = torch.randn(3 * heads * hd, generator=g, device=dev).half()
    kc0 = torch.randn(heads, ctx, hd, generator=g, device=dev).half()
    vc0 = torch.randn(heads, ctx, hd, generator=g, device=dev).half()
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    ang = torch.arange(ctx, device=dev).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    posd = torch.tensor([pos], dtype=torch.int64, device=dev)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    row = torch.empty(hd, dtype=torch.float32, device=dev)
    _lib.check(lib.mxq_rope_row_f32(posd.data_ptr(), cos.data_ptr(), sin.data_ptr(), row.data_ptr(), hd // 2, ctx, st), "rope_row")
    ws = torch.zeros(lib.mxq_attn_split_workspace_bytes(heads, splits), dtype=torch.uint8, device=dev)
    outs = []
    for rep in range(2):                      # twice on the same workspace: the counters must come back zeroed
        kc, vc = kc0.clone(), vc0.clone()
        out = torch.empty(heads * hd, dtype=torch.float16, device=dev)
        _lib.check(lib.mxq_attn_decode_split_f16(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(), posd.data_ptr(), row.data_ptr(),
                                                 out.data_ptr(), heads, hd, ctx, splits, ws.data_ptr(), st), "attn split")
        torch.cuda.synchronize()
        assert int(ws[:1024].view(torch.int32).abs().sum().item()) == 0
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    q, k, v = (qkv[j * heads * hd:(j + 1) * heads * hd].view(heads, hd).float() for j in range(3))

    def rope(t):
        t1, t2 = t[:, : hd // 2], t[:, hd // 2:]
        return torch.cat([t1 * cos[pos] - t2 * sin[pos], t2 * cos[pos] + t1 * sin[pos]], -1)
    qr, kr = rope(q).half().float(), rope(k).half()
    kref, vref = kc0.clone(), vc0.clone()
    kref[:, pos], vref[:, pos] = kr, v.half()
    assert torch.equal(kc, kref) and torch.equal(vc, vref)
    if pos + 1 <= 128:                         # one workgroup per head: the other kernel's result bit for bit
        one = torch.empty_like(out)
        kc1, vc1 = kc0.clone(), vc0.clone()
        _lib.check(lib.mxq_attn_decode_row_f16(qkv.data_ptr(), kc1.data_ptr(), vc1.data_ptr(), posd.data_ptr(), row.data_ptr(),
                                               one.data_ptr(), heads, hd, ctx, st), "attn row")
        assert torch.equal(out, one)
    scores = (qr[:, None, :] @ kref[:, : pos + 1].float().transpose(1, 2) / hd ** 0.5).half().float()
    want = (scores.softmax(-1) @ vref[:, : pos + 1].float()).reshape(-1)
    assert ((out.float() - want).abs().max() / want.abs().max()).item() < 4e-3


@pytest.mark.parametrize("M", [1, 5, 8, 16, 17, 32, 33, 48, 64])
@pytest.mark.parametrize("N,K", [(64, 256), (256, 704), (4096, 4096), (11008, 4096), (4096, 11008)])
@pytest.mark.parametrize("compact", [False, True])
def test_skinny_mfma_vs_oracle(dev, M, N, K, compact):
    """csrc/skinny.hip (4 < M <= 32, also callable down to 1 token): against the oracle's weight, exact and compact
    metadata, ragged K (704 = 11 chunks over 16 waves), the three Llama shapes; the automatic dispatch picks it for
    5..32 tokens; rows agree with the GEMV / GEMM paths to accumulation order."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(M * 3 + N + K)
    W16 = (torch.randn(N, K, generator=g) * 0.02).half()
    x = torch.randn(M, K, generator=g).half()
    if N * K <= 256 * 704:
        ref = O.mxq_quantize(W16.numpy())
        w16 = O.mxq_dequant(O.mxq_compact_params(ref) if compact else ref).astype(np.float16)
    else:                                   # Llama shapes: the dequant kernel's weight (bit-exact vs the oracle in other tests)
        w16 = None
    p = packing.quantize_pack(W16.to(dev), compact_meta=compact)
    y = packing.linear(x.to(dev), p, path="skinny")
    if w16 is None:
        w16 = packing.dequant(p).cpu().numpy()
    _check_gemm(y.cpu().numpy(), O.linear_ref(x.numpy(), w16), f"skinny {M}x{N}x{K} compact={compact}")
    # the automatic dispatch takes the skinny kernel up to 40 tokens, 20 for weights beyond 24 M elements (capi.hip:
    # skinny_max_tokens -- beyond that the mid-M split-K kernel is the faster one), and agrees with it bit for bit there
    if 4 < M <= (20 if N * K > (24 << 20) else 40):
        assert torch.equal(packing.linear(x.to(dev), p), y), "auto dispatch should be the skinny kernel here"
    other = packing.linear(x.to(dev), p, path="gemv" if M <= 4 else "gemm").float()
    assert ((other - y.float()).abs().max() / other.abs().max()).item() <= REL_TOL


def test_skinny_integer_exact_and_nonfinite(dev):
    """Small-integer weights / activations: every partial sum is exact, so a wrong lane -> (row, k) mapping of the
    register-built MFMA operands shows as an exact mismatch; and a NaN activation poisons its own token only."""
    from mxq_amd import packing
    N, K, M = 64, 512, 19
    rng = np.random.default_rng(11)
    p = O.mxq_quantize(np.zeros((N, K), np.float16))
    p["codes2"] = rng.integers(0, 4, (N, K // 64 * 48), dtype=np.uint8)
    p["codes4"] = rng.integers(0, 16, (N, K // 4), dtype=np.uint8)
    p["sc2"][:] = 1; p["qs2"][:] = 1.0; p["qz2"][:] = 0.0; p["zero2"][:] = 1.0     # w = q - 1
    p["sc4"][:] = 1; p["qs4"][:] = 1.0; p["qz4"][:] = 0.0; p["zero4"][:] = 7.0     # w = q - 7
    w = O.mxq_dequant(p)
    x = rng.integers(-2, 3, (M, K)).astype(np.float16)
    pk = packing.pack_codes(_to_dev(p, dev), N, K)
    yref = x.astype(np.float32) @ w.T
    for pp in (pk, packing.compact(pk)):                     # integer zero-points are exact in fp16 too
        y = packing.linear(torch.from_numpy(x).to(dev), pp, path="skinny").cpu().numpy().astype(np.float32)
        assert np.array_equal(y, yref)
    xt = torch.from_numpy(x).to(dev)
    xt[3, 100] = float("nan")
    y = packing.linear(xt, pk, path="skinny")
    assert torch.isnan(y[3]).all() and torch.isfinite(y[[i for i in range(M) if i != 3]]).all()
    with pytest.raises(ValueError):
        packing.linear(torch.zeros(65, K, dtype=torch.float16, device=dev), pk, path="skinny")


@pytest.mark.parametrize("M", [21, 41, 49, 64, 100, 128, 200, 256, 512, 1000])   # 21 / 41: just above the dispatch's skinny limits
@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008), (208, 2176), (4096, 192)])
def test_midm_split_k_llama_shapes(dev, M, N, K):
    """The mid-M split-K kernel (csrc/midm.hip; reference analogue: the split_k_iters launcher,
    gemm_cuda_gen.cu:429-475) at the Llama shapes and token counts it serves -- one and two token tiles, ragged M,
    K slices of equal and unequal length (172 chunks), a ragged last channel tile (N = 208), an odd chunk count
    (K = 192: the second chunk of the last double-step lies beyond K) -- against the fp32 product on the bit-exact
    dequant kernel's weight, and the automatic dispatch against the same."""
    from mxq_amd import packing
    g = torch.Generator(device="cpu").manual_seed(N + K + M)
    p = packing.quantize_pack((torch.randn(N, K, generator=g) * 0.02).half().to(dev))
    wd = packing.dequant(p).float()
    x = torch.randn(M, K, generator=g).half().to(dev)
    yref = x.float() @ wd.t()
    for path in ("midm", "auto"):
        y = packing.linear(x, p, path=path).float()
        assert ((y - yref).abs().max() / yref.abs().max()).item() <= REL_TOL, path
        assert ((y - yref).norm() / yref.norm()).item() <= REL_TOL, path
    # deterministic: the slabs are summed in slice order, whatever the arrival order
    assert torch.equal(packing.linear(x, p, path="midm"), packing.linear(x, p, path="midm"))
    # compact metadata (fp16 zero-points): the same kernel behind mxq_linear_f16_layout_ws, against ITS dequantised weight
    pc = packing.compact(p)
    yc_ref = x.float() @ packing.dequant(pc).float().t()
    yc = packing.linear(x, pc, path="auto").float()
    assert ((yc - yc_ref).abs().max() / yc_ref.abs().max()).item() <= REL_TOL
    assert ((yc - yc_ref).norm() / yc_ref.norm()).item() <= REL_TOL


@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_full_size_properties(dev, N, K):
    """BASELINE config 2 shapes (M = 2048).  The oracle is too slow here, so use
    size-independent properties: (1) GEMM == fp32 matmul against the bit-exact dequant
    kernel's output, (2) GEMV rows agree with GEMM rows, (3) linearity in x."""
    from mxq_amd import packing
    g = torch.Generator(device="cpu").manual_seed(N ^ K)
    W = (torch.randn(N, K, generator=g) * 0.02).half().to(dev)
    p = packing.quantize_pack(W)
    wd = packing.dequant(p)
    M = 2048
    x = torch.randn(M, K, generator=g).half().to(dev)
    y = packing.linear(x, p, path="gemm").float()
    yref = x.float() @ wd.float().t()
    assert ((y - yref).abs().max() / yref.abs().max()).item() <= REL_TOL
    assert ((y - yref).norm() / yref.norm()).item() <= REL_TOL
    yv = packing.linear(x[:4], p, path="gemv").float()
    assert ((yv - yref[:4]).abs().max() / yref[:4].abs().max()).item() <= REL_TOL
    x2 = torch.randn(M, K, generator=g).half().to(dev) * 0.5
    y2 = packing.linear(x2, p, path="gemm").float()
    y12 = packing.linear((x.float() + x2.float()).half(), p, path="gemm").float()
    assert ((y12 - (y + y2)).norm() / y12.norm()).item() <= 2e-3
    # unpack -> pack is the identity on the packed bytes (checksum of checksums)
    pk = packing.pack_codes(packing.unpack(p), N, K)
    assert torch.equal(pk.qweight, p.qweight) and torch.equal(pk.rowmeta, p.rowmeta)


@pytest.mark.parametrize("layout,N,K", [("mixed", 4096, 4096), ("mixed", 11008, 4096), ("mixed", 4096, 11008),
                                        ("w2g16", 4096, 4096), ("w2g16", 4096, 11008),
                                        ("w4row", 4096, 4096), ("w4row", 11008, 4096)])
def test_config5_full_size_m32768(dev, layout, N, K):
    """BASELINE configs[4]: batch 8 x seq 4096 = 32768 tokens, per weight layout (mixed 2/4, uniform W2 group 16,
    uniform W4 per row).  Size-independent properties at the full size: (1) the GEMM equals the fp32 product on the
    dequant kernel's weight (itself bit-exact against the oracle at the sizes the oracle can run) on every token
    row, max-norm and Frobenius <= 1e-3; (2) 16 sampled rows against the ORACLE's weight (numpy restatement);
    (3) the first and the last 2048 tokens agree bit for bit with the same rows run as their own whole-tile launch (a
    tile's result does not depend on where the persistent loop runs it), and to <= 1e-3 with the dispatch's launch."""
    from mxq_amd import packing
    M = 32768
    g = torch.Generator(device="cpu").manual_seed(N * 3 + K + len(layout))
    W16 = (torch.randn(N, K, generator=g) * 0.02).half()
    W = W16.to(dev)
    if layout == "mixed":
        p = packing.quantize_pack(W)
        wd = packing.dequant(p)
    else:
        p = packing.quantize_pack_uniform(W, layout)
        wd = packing.expand_uniform(p, codes=False)[0]
    x = torch.randn(M, K, generator=g).half().to(dev)
    y = packing.linear_layout(x, p)
    worst_max = worst_fro = 0.0
    for r0 in range(0, M, 4096):                       # fp32 reference in slabs (keeps the fp32 copies small)
        ref = x[r0:r0 + 4096].float() @ wd.float().t()
        got = y[r0:r0 + 4096].float()
        worst_max = max(worst_max, ((got - ref).abs().max() / ref.abs().max()).item())
        worst_fro = max(worst_fro, ((got - ref).norm() / ref.norm()).item())
    assert worst_max <= REL_TOL and worst_fro <= REL_TOL, (layout, N, K, worst_max, worst_fro)
    rows = [0, 255, 256, 4095, 16384, 20000, 32767, 31999, 12345, 777, 1024, 8191, 8192, 30000, 2047, 2048]
    w_or = (O.mxq_quantize(W16.numpy())["w_deq32"] if layout == "mixed"
            else O.uniform_quantize(W16.numpy(), layout)["w_deq32"]).astype(np.float16)
    assert np.array_equal(wd.cpu().numpy().view(np.uint16), w_or.view(np.uint16)), "dequant kernel vs oracle at full size"
    _check_gemm(y[rows].cpu().numpy(), O.linear_ref(x[rows].cpu().numpy(), w_or), f"{layout} M=32768 rows vs oracle")
    for r0 in (0, M - 2048):
        xs = x[r0:r0 + 2048].contiguous()
        alone = packing.linear_layout(xs, p, path="whole")          # whole tiles: the same sums in the same order
        assert torch.equal(alone, y[r0:r0 + 2048]), (layout, r0)
        # the dispatch may split the tail of a 2048-token launch along K (gate/up does since round 4): other summation order
        disp = packing.linear_layout(xs, p)
        assert ((disp.float() - alone.float()).abs().max() / alone.float().abs().max()).item() <= REL_TOL, (layout, r0)


@pytest.mark.parametrize("layout", ["mixed", "mixedc", "w2g16", "w4row"])
@pytest.mark.parametrize("M,N,K", [(300, 144, 192), (2048, 4096, 4096), (8192, 1024, 11008), (4100, 4224, 1024)])
def test_hoisted_dequant_mode_is_bit_identical_to_the_fused_gemm(dev, layout, M, N, K):
    """mxq_linear_f16_hoisted (dequant kernel once into a scratch buffer + the MFMA kernel on fp16 tiles) multiplies
    the same fp16 weights in the same order as the fused kernel: the outputs are bit-identical, on every layout,
    with ragged M / N edges; "auto" takes the hoisted mode from HOIST_MIN_TOKENS tokens on."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(M + N + K)
    W = (torch.randn(N, K, generator=g) * 0.02).half().to(dev)
    x = torch.randn(M, K, generator=g).half().to(dev)
    if layout in ("mixed", "mixedc"):
        p = packing.quantize_pack(W, compact_meta=layout == "mixedc")
        wd = packing.dequant(p)
    else:
        p = packing.quantize_pack_uniform(W, layout)
        wd = packing.expand_uniform(p, codes=False)[0]
    fused = packing.linear_layout(x, p, path="fused" if layout.startswith("mixed") else "gemm") if M < packing.HOIST_MIN_TOKENS \
        else None
    hoisted = packing.linear_layout(x, p, path="hoist")
    ref = x.float() @ wd.float().t()
    assert ((hoisted.float() - ref).abs().max() / ref.abs().max()).item() <= REL_TOL
    if fused is not None and M % 256 == 0 and (M // 256) * ((N + 127) // 128) % 256 == 0:
        assert torch.equal(hoisted, fused)                  # no stream-K split involved: same summation order
    elif fused is not None:
        assert ((hoisted.float() - fused.float()).abs().max() / ref.abs().max()).item() <= REL_TOL
    if M >= packing.HOIST_MIN_TOKENS:
        assert torch.equal(packing.linear_layout(x, p, path="auto"), hoisted)


@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (300, 272, 256), (1000, 784, 4096), (8192, 4096, 256),
                                   (515, 512, 11008), (2304, 11008, 512), (12288, 1024, 384)])
def test_dense256_kernel_is_bit_identical_to_the_256x128_one(dev, M, N, K):
    """mxq_dense_f16 (the reference's nn.Linear on the fake-quant fp16 weight, mxq_quant/main.py:85): the 256 x 256-tile
    quadrant-phase kernel (csrc/dense256.hip) multiplies the same fp16 operands in the same K order as the 256 x 128
    kernel's dense instantiation -- identical bits -- on one tile, ragged M / N edges, Llama's K-tile counts (64, 172),
    several tiles per persistent workgroup (512-688 tiles on 256 CUs: the unit pipeline runs across tile boundaries),
    twice in a row (no state left behind)."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(M * 3 + N + K)
    w16 = (torch.randn(N, K, generator=g) * 0.02).half().to(dev)
    x = torch.randn(M, K, generator=g).half().to(dev)
    a = packing.linear_dense(x, w16, variant="dense128")
    b = packing.linear_dense(x, w16, variant="dense256")
    rows = torch.arange(0, M, max(1, M // 512), device=dev)
    ref = x[rows].float() @ w16.float().t()
    assert ((b[rows].float() - ref).abs().max() / ref.abs().max()).item() <= REL_TOL
    assert torch.equal(a, b)
    assert torch.equal(packing.linear_dense(x, w16, variant="dense256"), b)
    assert torch.equal(packing.linear_dense(x, w16), b)                       # "auto": whichever kernel, the same bits


def test_dense256_rejects_an_odd_k_tile_count(dev):
    from mxq_amd import packing
    x = torch.zeros(256, 192, dtype=torch.float16, device=dev)
    w16 = torch.zeros(256, 192, dtype=torch.float16, device=dev)
    with pytest.raises(ValueError):
        packing.linear_dense(x, w16, variant="dense256")
    assert packing.linear_dense(x, w16).abs().max().item() == 0.0               # auto falls back to the 256 x 128 kernel


# ----------------------------------------------------------------------------------------
# compact metadata mode (format v2: fp16 zero-points, 3.75 bit/weight)
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K", [(64, 256), (256, 704), (1024, 4096)])
def test_compact_pack_unpack_dequant(dev, N, K):
    """Compact blocks: the integer unpack is bit-exact on the codes / scale codes / (qs, qz) of the EXACT quantiser;
    the zero-points come back as fp16(zero); the dequant kernel is bit-exact against the oracle's restatement of
    the compact parameter set; 3.75 bit/weight + rowmeta."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(N + K)
    W16 = (torch.randn(N, K, generator=g) * 0.02).half()
    ref = O.mxq_quantize(W16.numpy())
    cref = O.mxq_compact_params(ref)
    pe = packing.quantize_pack(W16.to(dev))
    pc = packing.compact(pe)
    assert pc.compact and pc.qweight.numel() * 4 == (N // 16) * (K // 64) * 480 and packing.compact(pc) is pc
    assert abs(pc.bits_per_weight() - (3.75 + 128.0 / K)) < 1e-6
    got = {k: v.cpu().numpy() for k, v in packing.unpack(pc).items()}
    for k in packing.PARAM_KEYS:
        assert np.array_equal(got[k], cref[k]), k
    for k in ("codes2", "codes4", "sc2", "sc4", "qs2", "qz2", "qs4", "qz4", "zero4"):
        assert np.array_equal(got[k], ref[k]), k                         # untouched by the compaction
    wc = packing.dequant(pc).cpu().numpy()
    assert np.array_equal(wc.view(np.uint16), O.mxq_dequant(cref).astype(np.float16).view(np.uint16))
    # quantize_pack(compact_meta=True) is the same thing in one call
    assert torch.equal(packing.quantize_pack(W16.to(dev), compact_meta=True).qweight, pc.qweight)


@pytest.mark.parametrize("M,N,K", [(1, 64, 256), (3, 256, 704), (4, 4096, 4096), (128, 256, 512), (300, 144, 192),
                                   (2048, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008), (1, 4096, 11008)])
def test_compact_linear_within_budget_of_exact_reference(dev, M, N, K):
    """GEMM / GEMV on compact blocks against the EXACT reference weight (what the reference's Python path produces):
    <= 1e-3 in max-norm and Frobenius, the bar of north_star; and against the compact weight itself, where only the
    accumulation differs."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(M + N + K)
    W = (torch.randn(N, K, generator=g) * 0.02).half().to(dev)
    pe = packing.quantize_pack(W)
    pc = packing.compact(pe)
    x = torch.randn(M, K, generator=g).half().to(dev)
    y = packing.linear(x, pc).float()
    exact = x.float() @ packing.dequant(pe).float().t()
    own = x.float() @ packing.dequant(pc).float().t()
    for ref, what in ((exact, "exact reference"), (own, "compact weight")):
        err_max = ((y - ref).abs().max() / ref.abs().max()).item()
        err_fro = ((y - ref).norm() / ref.norm()).item()
        assert err_max <= REL_TOL and err_fro <= REL_TOL, (what, M, N, K, err_max, err_fro)
    if M <= 4:
        assert torch.equal(packing.linear(x, pc, path="gemv").float(), y)


def test_compact_checkpoint_and_fused_decode(dev, tmp_path):
    """Compact QuantLinears round-trip through the packed checkpoint (fmt version 2), and the decode stage's fused
    GEMVs (RMSNorm / SwiGLU / residual) run on compact blocks."""
    from mxq_amd import checkpoint, packing
    from mxq_amd.llama_decode import DecodeStage
    from mxq_amd.quant_linear import QuantLinear
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(256, 128, bias=False), torch.nn.Linear(128, 64, bias=True)).to(dev).half()
    names = checkpoint.pack_model(model, skip=(), compact=True)
    assert len(names) == 2 and all(isinstance(m, QuantLinear) and m.compact for m in model)
    x = torch.randn(5, 256, device=dev).half()
    y = model(x)
    checkpoint.save_packed(model, str(tmp_path))
    fresh = torch.nn.Sequential(torch.nn.Linear(256, 128, bias=False), torch.nn.Linear(128, 64, bias=True)).to(dev).half()
    checkpoint.load_packed(fresh, str(tmp_path))
    assert all(m.compact and int(m.fmt[0]) == 2 for m in fresh) and torch.equal(fresh(x), y)
    st_c = DecodeStage(range(1), dev, max_ctx=8, hidden=256, inter=704, heads=2, vocab=64, compact=True)
    st_e = DecodeStage(range(1), dev, max_ctx=8, hidden=256, inter=704, heads=2, vocab=64)
    assert st_c.packed_bytes() < 0.86 * st_e.packed_bytes()
    h = torch.randn(1, 256, device=dev).half()
    a, b = st_c.step(h).float(), st_e.step(h).float()
    assert ((a - b).abs().max() / b.abs().max()).item() < 2e-2          # fp16 activations end to end


def test_quantlinear_module(dev):
    from mxq_amd.quant_linear import QuantLinear
    lin = torch.nn.Linear(512, 256, bias=True).to(dev).half()
    ql = QuantLinear.from_linear(lin)
    x = torch.randn(2, 9, 512, device=dev).half()
    y = ql(x)
    assert y.shape == (2, 9, 256) and y.dtype == torch.float16
    yref = torch.nn.functional.linear(x.float(), ql.dequantize().float(), lin.bias.float())
    assert ((y.float() - yref).abs().max() / yref.abs().max()).item() <= REL_TOL
    sd = ql.state_dict()
    ql2 = QuantLinear(512, 256, bias=True, device=dev)
    ql2.load_state_dict(sd)
    assert torch.equal(ql2(x), y)
    with pytest.raises(ValueError):
        ql(x.float())
    with pytest.raises(ValueError):
        ql(torch.randn(2, 100, device=dev).half())
    # fused q | k | v style module: same numbers as the parts, one launch
    parts = [QuantLinear.from_linear(torch.nn.Linear(512, n, bias=True).to(dev).half()) for n in (256, 128, 64)]
    fused = QuantLinear.fuse(parts)
    assert fused.out_features == 448
    for xx in (x, torch.randn(300, 512, device=dev).half()):       # GEMV path and GEMM path
        assert torch.equal(fused(xx), torch.cat([p(xx) for p in parts], dim=-1))


class _MlpBlock(torch.nn.Module):
    """Residual block with four Linears (two of them behind a nested name), fp16."""
    def __init__(self, h, inter):
        super().__init__()
        self.attn = torch.nn.Module()
        self.attn.v_proj = torch.nn.Linear(h, h, bias=False)
        self.attn.o_proj = torch.nn.Linear(h, h, bias=True)
        self.up = torch.nn.Linear(h, inter, bias=False)
        self.down = torch.nn.Linear(inter, h, bias=False)

    def forward(self, x, scale=1.0):
        x = x + self.attn.o_proj(self.attn.v_proj(x)) * scale
        return (x + self.down(torch.nn.functional.silu(self.up(x))) * scale,)      # tuple, like a decoder layer


def test_quantize_sequential_and_packed_checkpoint(dev, tmp_path):
    """The nas_quant layer loop (prune.py:368-420) on two tiny blocks: every Linear's fake-quant weight is
    bit-equal to the oracle's quantisation of the ORIGINAL weight with the layer's dead input channels
    (diag(H) == 0) zeroed; with pack=True the model runs on QuantLinear modules, and the packed checkpoint
    reloads into a fresh model that reproduces its outputs bit for bit."""
    import copy
    from mxq_amd import checkpoint
    from mxq_amd.lib.prune import check_sparsity_linear, find_layers, quantize_sequential
    from mxq_amd.quant_linear import QuantLinear
    torch.manual_seed(3)
    h, inter, ns, seq = 128, 192, 3, 24
    layers = torch.nn.ModuleList([_MlpBlock(h, inter), _MlpBlock(h, inter)]).to(dev).half()
    with torch.no_grad():
        for m in layers.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(0.3)
    orig = {f"{i}.{n}": lin.weight.detach().cpu().numpy().copy() for i, l in enumerate(layers) for n, lin in find_layers(l).items()}
    inps = torch.randn(ns, seq, h, device=dev).half()
    inps[:, :, 5] = 0        # an input channel no calibration sample activates: dead for layer 0's v_proj and up
    inps[:, :, 77] = 0
    a, b = copy.deepcopy(layers), copy.deepcopy(layers)
    packed = quantize_sequential(a, inps.clone(), {"scale": 0.5})
    assert list(packed) == [f"{i}.{n}" for i in (0, 1) for n in ("attn.v_proj", "attn.o_proj", "up", "down")]
    for key, w0 in orig.items():
        i, n = key.split(".", 1)
        dead = np.zeros(w0.shape[1], bool)
        if i == "0" and n in ("attn.v_proj",):
            dead[[5, 77]] = True        # ("up" sees x + attn(x): the residual keeps channels 5/77 zero only if attn adds 0 there)
        ref = O.mxq_quantize(w0, dead)["w_deq32"].astype(np.float16)
        got = find_layers(a[int(i)])[n].weight.detach().cpu().numpy()
        if n == "up" and i == "0":
            continue                    # dead-ness depends on the block's own arithmetic; covered through v_proj
        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), key
    assert 0.0 < check_sparsity_linear(a) < 0.2
    # packed variant: same quantisation decisions, QuantLinear modules in place
    packed_b = quantize_sequential(b, inps.clone(), {"scale": 0.5}, pack=True)
    for key in packed:
        assert torch.equal(packed[key].qweight, packed_b[key].qweight), key
    assert all(isinstance(m, QuantLinear) for l in b for m in find_layers(l, layers=[QuantLinear]).values())
    assert len(find_layers(b[0], layers=[QuantLinear])) == 4 and not find_layers(b[0])
    x = torch.randn(2, seq, h, device=dev).half()
    ya = a[1](a[0](x, 0.5)[0], 0.5)[0]
    yb = b[1](b[0](x, 0.5)[0], 0.5)[0]
    assert ((ya.float() - yb.float()).abs().max() / ya.float().abs().max()).item() <= 2e-3
    model = torch.nn.ModuleDict({"layers": b})
    d = checkpoint.save_packed(model, str(tmp_path / "ck"))
    fresh = torch.nn.ModuleDict({"layers": torch.nn.ModuleList([_MlpBlock(h, inter), _MlpBlock(h, inter)]).to(dev).half()})
    checkpoint.load_packed(fresh, d)
    yf = fresh["layers"][1](fresh["layers"][0](x, 0.5)[0], 0.5)[0]
    assert torch.equal(yf, yb)
    import os
    packed_bytes = os.path.getsize(os.path.join(d, checkpoint.WEIGHTS_NAME))
    fp16_bytes = sum(w.size * 2 for w in orig.values())
    assert packed_bytes < 0.45 * fp16_bytes        # ~4.5-5 bit/weight at these small K, vs 16
    # round-to-nearest packing of a whole module tree
    c = copy.deepcopy(layers)
    names = checkpoint.pack_model(torch.nn.ModuleDict({"layers": c, "lm_head": torch.nn.Linear(h, 64).to(dev).half()}))
    assert len(names) == 8 and not any("lm_head" in n for n in names)


# ----------------------------------------------------------------------------------------
# the reference extension's entry points (mxq_inference_engine)
# ----------------------------------------------------------------------------------------
def test_kat_test_correct_gemv(dev, g6):
    """cuda_kernel/test_correct_gemv.py restated: constants -> every output == 4096."""
    import mxq_inference_engine as eng
    N, K = int(g6["N"]), int(g6["K"])
    full = lambda shape, v, dt: torch.from_numpy(np.full(shape, v, dt)).to(dev)
    i32 = lambda v: np.array(v, np.uint32).view(np.int32)
    C = eng.gemv_mxq_forward_cuda(
        full((1, K), g6["x"], np.float16), full((N, 256), i32(g6["weight_2b"]), np.int32),
        full((N, 64), i32(g6["weight_4b"]), np.int32), full((N, 32), i32(g6["zeros_and_scales_1st"]), np.int32),
        full((N // 4, 256), g6["scales_2nd"], np.float16), full((N // 4, 32), i32(g6["zeros_2nd"]), np.int32),
        full((N,), g6["scales_4b"], np.float16), full((N // 8,), i32(g6["zeros_4b"]), np.int32), 16)
    torch.cuda.synchronize()
    assert C.shape == (1, N) and torch.all(C.int() == int(g6["expected"]))
    with pytest.raises(ValueError):
        eng.gemv_mxq_forward_cuda(
            full((1, K), 1, np.float16), full((N, 256), 0, np.int32), full((N, 64), 0, np.int32),
            full((N, 32), 0, np.int32), full((N // 4, 256), 1, np.float16), full((N // 4, 32), 0, np.int32),
            full((N,), 1, np.float16), full((N // 8,), 0, np.int32), 32)   # unsupported group size raises


@pytest.mark.parametrize("B", [1, 2, 5])     # 5: two batch blocks of the in-kernel batch loop, the second one ragged
def test_proto_and_awq_gemv_random_vs_oracle(dev, B):
    import mxq_inference_engine as eng
    rng = np.random.default_rng(3 + B)
    OC, IC = 64, 4096
    ri = lambda shape: rng.integers(0, 2 ** 32, shape, dtype=np.uint64).astype(np.uint32)
    x = (rng.standard_normal((B, IC)) * 0.5).astype(np.float16)
    ops = dict(weight=ri((OC, 256)), weight_last=ri((OC, 64)), zs=ri((OC, 32)),
               s2=(rng.standard_normal((OC // 4, 256)) * 0.01).astype(np.float16), z2=ri((OC // 4, 32)),
               s4=(rng.standard_normal((OC,)) * 0.01).astype(np.float16), z4=ri((OC // 8,)))
    yref = O.gemv_mxq_proto_ref(x, ops["weight"], ops["weight_last"], ops["zs"], ops["s2"], ops["z2"], ops["s4"],
                                ops["z4"])
    t = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev)
    y = eng.gemv_mxq_forward_cuda(t(x), t(ops["weight"]), t(ops["weight_last"]), t(ops["zs"]), t(ops["s2"]),
                                  t(ops["z2"]), t(ops["s4"]), t(ops["z4"]), 16).cpu().numpy()
    _check_gemm(y, yref, "proto gemv")
    for G in (32, 64, 128):
        kern = ri((OC, IC // 8))
        zw = ((IC // G + 7) // 8 + 3) // 4 * 4
        sc = (rng.standard_normal((OC, zw * 8)) * 0.01).astype(np.float16)
        zz = ri((OC, zw))
        yref = O.gemv_awq_ref(x, kern, sc, zz, G)
        y = eng.gemv_forward_cuda(t(x), t(kern), t(sc), t(zz), G).cpu().numpy()
        _check_gemm(y, yref, f"awq gemv g{G}")


def test_reference_timing_script_operands_vs_oracle(dev):
    """The operands of the reference's timing harness, cuda_kernel/test_mxq_gemv.py:35-73 (M = 1, N = K = 4096), seeded
    instead of unseeded: `gemv_forward_cuda(A, B, scales, zeros, 128)` and
    `gemv_mxq_forward_cuda(A, B, B, scales_1nd, scales_2nd, zeros_2nd, scales_4b, zeros_4b, 16)` -- the SAME tensor `B`
    [N, K/16] as `kernel` and as `kernel_last` (the kernel reads its first N * K/64 words, row stride K/64,
    gemv_mxq_cuda.cu:56), `scales_2nd` as [N/4, K/16] read with a row stride of 192 (:61), and `zeros_2nd` as [N/4, 16]
    although the kernel's row stride is 32 (:59, :70: the reference reads out of bounds; here the operand is zero-padded,
    mxq_inference_engine/__init__.py).  The script checks no values; this test checks them against the oracle given
    exactly what the kernel is specified to read."""
    import mxq_inference_engine as eng
    M, N, K = 1, 4096, 4096
    g = torch.Generator().manual_seed(20240607)
    A = torch.randn((M, K), generator=g).half()
    # -- awq_4bit leg (script lines 35-54)
    B8 = torch.randint(-1000000000, 1000000000, (N, K // 8), generator=g, dtype=torch.int32)
    scales = torch.randn((N, K // 128), generator=g).half()
    zeros = torch.ones((N, K // 128 // 8), dtype=torch.int32)
    C = eng.gemv_forward_cuda(A.to(dev), B8.to(dev), scales.to(dev), zeros.to(dev), 128)
    assert C.shape == (M, N) and C.dtype == torch.float16
    _check_gemm(C.cpu().numpy(), O.gemv_awq_ref(A.numpy(), B8.numpy(), scales.numpy(), zeros.numpy(), 128), "script awq leg")
    # -- mxq_2.8bit leg (script lines 63-82)
    B16 = torch.randint(-1000000000, 1000000000, (N, K // 16), generator=g, dtype=torch.int32)
    scales_1nd = torch.ones((N, K // 16 // 16 * 2), dtype=torch.int32)
    scales_2nd = torch.randn((N // 4, K // 16), generator=g).half()
    zeros_2nd = torch.ones((N // 4, K // 16 // 16), dtype=torch.int32)
    scales_4b = torch.randn(N, generator=g).half()
    zeros_4b = torch.ones(N // 8, dtype=torch.int32)
    Bd = B16.to(dev)
    C = eng.gemv_mxq_forward_cuda(A.to(dev), Bd, Bd, scales_1nd.to(dev), scales_2nd.to(dev), zeros_2nd.to(dev),
                                  scales_4b.to(dev), zeros_4b.to(dev), 16)
    assert C.shape == (M, N) and C.dtype == torch.float16
    last = B16.numpy().reshape(-1)[: N * (K // 64)].reshape(N, K // 64)         # what a row stride of K/64 sees of B
    z2 = np.zeros((N // 4) * 32, np.int32)
    z2[: zeros_2nd.numel()] = zeros_2nd.numpy().reshape(-1)                       # the padded operand
    want = O.gemv_mxq_proto_ref(A.numpy(), B16.numpy(), last, scales_1nd.numpy(), scales_2nd.numpy(), z2.reshape(N // 4, 32),
                                scales_4b.numpy(), zeros_4b.numpy())
    _check_gemm(C.cpu().numpy(), want, "script mxq leg")


# ----------------------------------------------------------------------------------------
# QAT: MXAsymQuantizer / QuantizeLinear
# ----------------------------------------------------------------------------------------
def _t(a, dt, dev):
    if dt == "bf16":
        return torch.from_numpy(a.view(np.int16)).view(torch.bfloat16).to(dev)
    return torch.from_numpy(a).to(dev)


def _bits(t):
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16 or t.dtype == torch.float16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy().view(np.uint32)


def _same(a_bits, ref):
    ref_bits = ref.view(np.uint16) if ref.dtype in (np.uint16, np.float16) else ref.view(np.uint32)
    if ref.dtype in (np.float16, np.float32):          # NaN payloads may differ; NaN-ness must not
        nan_ref = np.isnan(ref)
        got = a_bits.view(ref.dtype)
        return bool(np.all((a_bits == ref_bits) | (nan_ref & np.isnan(got))))
    return bool(np.array_equal(a_bits, ref_bits))


@pytest.mark.parametrize("bits", [2, 3, 4])
@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
def test_g3_fakequant_fwd_bwd_bit_exact(dev, g3, dt, bits):
    from mxq_amd.utils_quant import MXAsymQuantizer
    key = f"{dt}_b{bits}"
    w = _t(g3[f"{key}_w"], dt, dev).requires_grad_()
    out = MXAsymQuantizer.apply(w, torch.tensor([-2.0, 2.0]), bits, False)
    out.backward(_t(g3[f"{key}_gout"], dt, dev))
    assert _same(_bits(out), g3[f"{key}_out"]), "forward"
    assert _same(_bits(w.grad), g3[f"{key}_gin"]), "backward"


@pytest.mark.parametrize("K", [4096, 11008])
def test_g3_fakequant_llama_width(dev, g3, K):
    from mxq_amd.utils_quant import mx_fake_quant
    out = mx_fake_quant(_t(g3[f"bf16_K{K}_w"], "bf16", dev), 2)
    assert np.array_equal(_bits(out), g3[f"bf16_K{K}_out"])


@pytest.mark.parametrize("shape", [(4096, 4096), (11008, 4096), (4096, 11008)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
def test_ste_backward_full_size(dev, shape, dt):
    """BASELINE configs[3] at the decoder block's real weight shapes: MXAsymQuantizer.backward (utils_quant.py:464-475) is
    grad_in = grad_out except 0 where w >= 2 or w <= -2 -- checked EXACTLY over the whole tensor against the same
    rule written with torch ops on the device (planted values at, just inside and beyond +-2.0, NaN / inf gradients pass
    through untouched), through the autograd Function the trainer calls."""
    from mxq_amd.utils_quant import MXAsymQuantizer
    g = torch.Generator().manual_seed(shape[0] + shape[1])
    w = (torch.randn(*shape, generator=g) * 0.9).to(dt)
    flat = w.view(-1)
    edge = torch.tensor([2.0, -2.0, 1.9921875, -1.9921875, 2.015625, -2.015625, 0.0, 65504.0 if dt != torch.bfloat16 else 3e38])
    flat[:: flat.numel() // 64][: edge.numel()] = edge.to(dt)
    flat[5] = 0.5
    flat[7] = -0.5                                                    # inside the clip range: their gradients pass
    go = torch.randn(*shape, generator=g).to(dt)
    go.view(-1)[5] = float("nan")
    go.view(-1)[7] = float("inf")
    wd = w.to(dev).requires_grad_()
    out = MXAsymQuantizer.apply(wd, torch.tensor([-2.0, 2.0]), 2, False)
    out.backward(go.to(dev))
    want = go.to(dev).clone()
    want[(wd.detach() >= 2.0) | (wd.detach() <= -2.0)] = 0
    got = wd.grad
    same = (got.view(torch.int16 if dt != torch.float32 else torch.int32) == want.view(torch.int16 if dt != torch.float32 else torch.int32))
    assert bool(same.all()), int((~same).sum())
    assert int((want == 0).sum()) > 0 and torch.isnan(got.view(-1)[5]) and torch.isinf(got.view(-1)[7])


@pytest.mark.parametrize("shape,dt", [((4096, 4096), "bf16"), ((1000, 11008), "bf16"), ((512, 4096), "fp32"),
                                      ((1001, 8192), "bf16"),      # row split over 2 waves, odd row count
                                      ((37, 16384), "bf16"),       # 4 waves per row at the register kernel's limit
                                      ((64, 12352), "bf16")])      # 193 chunks: not splittable -> two-pass kernel
def test_fakequant_full_size_properties(dev, shape, dt):
    """Config-4 shapes: idempotence (fake-quant of a fake-quant weight in fp32 is the
    identity up to one rounding), per-group level count, and agreement of a row sample
    with the oracle."""
    from mxq_amd.utils_quant import mx_fake_quant
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    g = torch.Generator().manual_seed(shape[0])
    w = (torch.randn(*shape, generator=g) * 0.02).to(tdt)
    out = mx_fake_quant(w.to(dev), 2)
    rows = [0, 1, shape[0] // 2, shape[0] - 1]
    ref = O.fakequant_fwd(w[rows].float().numpy(), 2, dt)
    got = out[rows].float().cpu().numpy()
    assert np.array_equal(got, ref)
    grp = out.float().reshape(shape[0], shape[1] // 64, 64)[:, :, :48].reshape(shape[0], -1, 16)
    sample = grp[:: max(1, shape[0] // 64)].cpu().numpy()
    levels = np.array([[len(np.unique(r)) for r in blk] for blk in sample])
    assert levels.max() <= 4                     # a 2-bit group has at most 4 distinct values


@pytest.mark.parametrize("shape", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_fakequant_whole_tensor_bf16_vs_oracle(dev, shape):
    """BASELINE configs[3] at full size: the WHOLE bf16 weight through MXAsymQuantizer's forward kernel, every element
    against the oracle (LLM-QAT/models/utils_quant.py:316-462 restated in oracle/mxq_oracle.py: one rounding per op),
    bit for bit.  The oracle is row-independent: it runs in bands of 512 rows (a few seconds per shape)."""
    from mxq_amd.utils_quant import mx_fake_quant
    g = torch.Generator().manual_seed(shape[0] * 3 + shape[1])
    w = (torch.randn(*shape, generator=g) * 0.02).bfloat16()
    w[5, 100] = 2.5                                   # an outlier group and an all-equal group
    w[7, 64:80] = 0.0123
    out = mx_fake_quant(w.to(dev), 2).float().cpu().numpy()
    wf = w.float().numpy()
    for r0 in range(0, shape[0], 512):
        ref = O.fakequant_fwd(wf[r0:r0 + 512], 2, "bf16")
        assert np.array_equal(out[r0:r0 + 512], ref), f"rows {r0}..{r0 + 511}"


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_g4_quantizelinear_fwd_bwd(dev, g4, dt):
    """QuantizeLinear(256, 64, w_bits=2, a_bits=16): weight fake-quant (HIP) is bit-exact, so
    y / dx / dW match the reference up to the GEMM's accumulation order."""
    from mxq_amd.utils_quant import QuantizeLinear
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    lin = QuantizeLinear(256, 64, w_bits=2, a_bits=16).to(dev).to(tdt)
    lin.weight.data = _t(g4[f"{dt}_w"], dt, dev)
    x = _t(g4[f"{dt}_x"], dt, dev).requires_grad_()
    y = lin(x)
    y.backward(_t(g4[f"{dt}_gy"], dt, dev))
    conv = (lambda a: O.bf16_from_bits(a)) if dt == "bf16" else (lambda a: a)
    tol = 2e-2 if dt == "bf16" else 1e-5
    for got, name in ((y, "y"), (x.grad, "dx"), (lin.weight.grad, "dw")):
        ref = conv(g4[f"{dt}_{name}"])
        err = np.abs(got.detach().float().cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err <= tol, (name, err)


# ----------------------------------------------------------------------------------------
# SymQuantizer / AsymQuantizer (activation / KV fake quant, csrc/actquant.hip)
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("q", ["sym", "asym"])
def test_g7_activation_quantizers_hip(dev, q):
    """fp32, forward and STE backward, every branch of the reference: 2-D groups, 3-D per token with the
    uncovered tokens of the long sequence, 4-D per (batch, head), layerwise."""
    from mxq_amd.utils_quant import AsymQuantizer, SymQuantizer
    from tests.conftest import load_golden
    g7 = load_golden("g7_act_quantizers.npz")
    Q = SymQuantizer if q == "sym" else AsymQuantizer
    clip = torch.tensor([-2.0, 2.0])
    for case in ("w2d", "a3d", "a3d_long", "s4d"):
        for bits in (4, 16):
            for lw in (0, 1):
                key = f"{case}_{q}_b{bits}_{lw}"
                x = torch.from_numpy(g7[key + "_x"]).to(dev).requires_grad_()
                y = Q.apply(x, clip, bits, bool(lw))
                y.backward(torch.from_numpy(g7[key + "_gy"]).to(dev))
                assert np.array_equal(y.detach().cpu().numpy(), g7[key + "_y"]), key
                assert np.array_equal(x.grad.cpu().numpy(), g7[key + "_gx"]), key


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_g9_activation_quantizers_16bit_hip(dev, dt):
    from mxq_amd.utils_quant import AsymQuantizer, SymQuantizer
    from tests.conftest import load_golden
    g9 = load_golden("g9_act16.npz")
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    clip = torch.tensor([-2.0, 2.0])
    keys = sorted(k[:-2] for k in g9.files if k.startswith(dt) and k.endswith("_x"))
    assert len(keys) == 60
    for key in keys:
        qn, bits, lw = key.split("_")[-3:]
        Q = SymQuantizer if qn == "sym" else AsymQuantizer
        x = torch.from_numpy(g9[key + "_x"].view(np.int16)).view(tdt).to(dev)
        y = Q.apply(x, clip, int(bits[1:]), bool(int(lw)))
        assert np.array_equal(y.cpu().view(torch.int16).numpy().view(np.uint16), g9[key + "_y"]), key


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,lw", [((2, 512, 4096), False),      # activations of a Llama block: one range per token (row kernel)
                                       ((1, 1100, 1024), False),     # row kernel with the reference's un-ranged tail tokens (1024 < 1100)
                                       ((4, 64, 4096), False),       # too few rows for the row kernel: two-pass path
                                       ((1, 32, 256, 128), False),   # KV cache: one range per head, 32 K elements each
                                       ((64, 11008), False),         # 2-D groups at Llama's MLP width
                                       ((3, 100, 4096), True)])      # layerwise: a single 1.2 M-element segment
def test_activation_quantizers_random_vs_oracle(dev, shape, lw, dt):
    """Bit-exact against the CPU restatement (itself pinned to G7 / G9) at realistic sizes, with NaN / inf
    free random data plus an all-zero token and a constant token."""
    from mxq_amd.utils_quant import AsymQuantizer, SymQuantizer
    from oracle import act_quant as OA
    g = torch.Generator().manual_seed(len(shape) * 31 + shape[-1])
    x = (torch.randn(*shape, generator=g) * 1.5).to(dt)
    x.reshape(-1, shape[-1])[1] = 0
    x.reshape(-1, shape[-1])[2] = 0.75
    clip = torch.tensor([-2.0, 2.0])
    for Qh, Qo, bits in ((SymQuantizer, OA.SymQuantizer, 16), (SymQuantizer, OA.SymQuantizer, 4),
                         (AsymQuantizer, OA.AsymQuantizer, 8)):
        y = Qh.apply(x.to(dev), clip, bits, lw).cpu()
        ref = Qo.apply(x, clip, bits, lw)
        it = torch.int32 if dt == torch.float32 else torch.int16
        assert torch.equal(y.view(it), ref.view(it)), (Qh.__name__, bits)


def test_activation_quantizer_backward_and_errors(dev):
    from mxq_amd.utils_quant import SymQuantizer
    x = (torch.randn(2, 16, 256, device=dev) * 1.5).requires_grad_()
    go = torch.randn(2, 16, 256, device=dev)
    SymQuantizer.apply(x, torch.tensor([-2.0, 2.0]), 8, False).backward(go)
    ref = go.clone(); ref[x.ge(2.0)] = 0; ref[x.le(-2.0)] = 0
    assert torch.equal(x.grad, ref)
    with pytest.raises(IndexError):
        SymQuantizer.apply(torch.zeros(64, device=dev), torch.tensor([-2.0, 2.0]), 8, False)
    with pytest.raises(ValueError):
        SymQuantizer.apply(torch.zeros(4, 30, device=dev), torch.tensor([-2.0, 2.0]), 8, False)


# ----------------------------------------------------------------------------------------
# the reference's PTQ flow on a real HF module tree (mxq_quant/main.py --prune_method mxq, lib/prune.py:338-420)
# ----------------------------------------------------------------------------------------
def test_hf_llama_ptq_flow_fake_quant_vs_packed(dev, tmp_path):
    """A tiny ``transformers`` LlamaForCausalLM (random init: no checkpoint offline) goes through the reference's
    flow -- calibration inputs recorded at the first decoder layer, then layer-by-layer MXQGPT quantisation -- once
    leaving fp16 fake-quant weights in ``nn.Linear`` (what the reference does) and once swapping in packed
    ``QuantLinear`` modules; the two models must produce the same logits (the packed kernels compute on the very
    weights the fake-quant model holds), and the packed checkpoint must reload into a fresh model bit for bit."""
    transformers = pytest.importorskip("transformers")
    from mxq_amd import checkpoint
    from mxq_amd.lib.prune import find_layers, prepare_calibration_input, quantize_sequential
    from mxq_amd.quant_linear import QuantLinear
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=704, num_attention_heads=2, num_key_value_heads=2,
                                   num_hidden_layers=2, vocab_size=128, max_position_embeddings=64)
    cfg._attn_implementation = "eager"

    def fresh():
        torch.manual_seed(11)
        m = transformers.LlamaForCausalLM(cfg).half().to(dev).eval()
        m.seqlen = 32
        return m
    g = torch.Generator().manual_seed(5)
    calib = [(torch.randint(0, 128, (1, 32), generator=g),) for _ in range(4)]
    probe = torch.randint(0, 128, (2, 32), generator=g).to(dev)
    models = []
    for pack in (False, True):
        m = fresh()
        with torch.no_grad():
            inps, _outs, _am, _pid, kw = prepare_calibration_input(m, calib, dev, nsamples=4, return_kwargs=True)
            packed = quantize_sequential(m.model.layers, inps, kw, pack=pack)
        assert len(packed) == 2 * 7
        models.append(m)
    fq, pk = models
    assert all(isinstance(l, QuantLinear) for layer in pk.model.layers for l in find_layers(layer, layers=[QuantLinear]).values())
    assert not find_layers(pk.model.layers[0]) and len(find_layers(fq.model.layers[0])) == 7
    # the packed modules hold exactly the fake-quant weights of the first model's first layer
    # (later layers see slightly different calibration activations: packed kernels vs fp16 matmul on the way)
    for name, lin in find_layers(fq.model.layers[0]).items():
        ql = find_layers(pk.model.layers[0], layers=[QuantLinear])[name]
        assert torch.equal(ql.dequantize(), lin.weight.data), name
    with torch.no_grad():
        a = fq(probe).logits.float()
        b = pk(probe).logits.float()
    assert ((a - b).abs().max() / a.abs().max()).item() < 3e-2          # fp16 model end to end, two GEMM implementations
    # one decoded token on top of a KV cache: the packed model's Linears now see 2 rows (batch 2 x 1 token: GEMV path)
    with torch.no_grad():
        pa = fq(probe[:, :16], use_cache=True)
        pb = pk(probe[:, :16], use_cache=True)
        a1 = fq(probe[:, 16:17], past_key_values=pa.past_key_values, use_cache=True).logits.float()
        b1 = pk(probe[:, 16:17], past_key_values=pb.past_key_values, use_cache=True).logits.float()
    assert ((a1 - b1).abs().max() / a1.abs().max()).item() < 3e-2
    checkpoint.save_packed(pk, str(tmp_path))
    again = checkpoint.load_packed(fresh(), str(tmp_path))
    with torch.no_grad():
        assert torch.equal(again(probe).logits, pk(probe).logits)


# ----------------------------------------------------------------------------------------
# decode stage (config 3 harness): fused q/k/v and gate/up GEMVs vs a dense fp32 restatement
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("heads,fused", [(4, False), (2, True), (2, False), (32, True), (32, False)])
def test_decode_stage_matches_dense_reference(dev, heads, fused):
    """fused = RMSNorm / SwiGLU / residual folded into the GEMV launches + the RoPE / cache /
    attention kernel (head_dim 128); unfused = torch ops around plain GEMVs."""
    from mxq_amd import packing
    from mxq_amd.llama_decode import DecodeStage
    # heads == 32: the full Llama-2-7B layer shape (hidden 4096, intermediate 11008, 32 heads of 128), 2 layers
    hidden, inter, ctx = (4096, 11008, 32) if heads == 32 else (256, 704, 32)
    st = DecodeStage(range(2), dev, max_ctx=ctx, hidden=hidden, inter=inter, heads=heads, vocab=64, fused=fused)
    assert st.fused == (fused and hidden // heads == 128)
    Wd = [[packing.dequant(p).float() for p in ws] for ws in st.w]
    hd = hidden // heads
    kc = torch.zeros(2, heads, ctx, hd, device=dev)
    vc = torch.zeros_like(kc)

    def rms(x):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5)

    def rope(t, pos):
        c, s = st.cos[pos], st.sin[pos]
        t1, t2 = t[:, : hd // 2], t[:, hd // 2:]
        return torch.cat([t1 * c - t2 * s, t2 * c + t1 * s], -1)

    def ref_step(h, pos):
        for i, (qkv, o, gu, down) in enumerate(Wd):
            y = rms(h) @ qkv.t()
            q, k, v = (y[0, j * hidden:(j + 1) * hidden].view(heads, hd) for j in range(3))
            q, k = rope(q, pos), rope(k, pos)
            kc[i][:, pos], vc[i][:, pos] = k, v
            att = (q[:, None, :] @ kc[i][:, : pos + 1].transpose(1, 2) / hd ** 0.5).softmax(-1)
            h = h + (att @ vc[i][:, : pos + 1]).reshape(1, hidden) @ o.t()
            g = rms(h) @ gu.t()
            h = h + (torch.nn.functional.silu(g[:, :inter]) * g[:, inter:]) @ down.t()
        return h

    g = torch.Generator(device=dev).manual_seed(4)
    st.capture()
    st.reset()
    for pos in range(5):
        h0 = torch.randn(1, hidden, generator=g, device=dev).half()
        got = st.step_graph(h0).float().clone()
        want = ref_step(h0.float(), pos)
        assert ((got - want).abs().max() / want.abs().max()).item() < 2e-2, pos     # fp16 activations end to end
    assert int(st.pos.item()) == 5


@pytest.mark.parametrize("hidden,inter", [(256, 704), (4096, 11008), (512, 1408)])
@pytest.mark.parametrize("compact", [False, True])
def test_swiglu_in_the_producer_is_bit_identical(dev, hidden, inter, compact):
    """Round 5: gate | up with the SwiGLU in its final reduction (mxq_gemv_swiglu_f16: the row blocks (i, i + I / 16) paired
    in one workgroup, the activation written in the kernels' staged order with its group sums) + the Linear on that
    staged row (mxq_gemv_staged_f16) against the two launches they replace (RMSNorm -> gate | up, then SwiGLU staging ->
    down + residual): the staged row is the permutation of fp16(silu(gate)) * up, the sums are its sequential fp32 group
    sums, and the layer output is identical bit for bit."""
    from mxq_amd import packing
    g = torch.Generator(device=dev).manual_seed(hidden + inter + int(compact))
    mk = lambda N, K: packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half(), compact_meta=compact)
    gu = packing.concat_packed([mk(inter, hidden), mk(inter, hidden)])
    down = mk(hidden, inter)
    norm_w = (1.0 + 0.1 * torch.randn(hidden, generator=g, device=dev)).half()
    for it in range(3):
        h = (torch.randn(1, hidden, generator=g, device=dev) * (1 + it)).half()
        y_gu = packing.linear_fused(h, gu, 1, norm_w)
        want = packing.linear_fused(y_gu, down, 2, residual=h)
        act, act_sum = packing.linear_swiglu(h, gu, norm_w)
        gate, up = y_gu[0, :inter].float(), y_gu[0, inter:]
        ref_act = (gate / (1.0 + torch.exp(-gate))).half() * up                     # fp16 multiply, as the staging does
        perm = torch.tensor([0, 2, 1, 3, 4, 6, 5, 7], device=dev)
        unstaged = act[0].view(-1, 8)[:, perm].reshape(-1)                           # the permutation is its own inverse
        close = (unstaged.float() - ref_act.float()).abs() <= 2e-3 * ref_act.float().abs() + 1e-6   # (__expf vs torch.exp)
        assert bool(close.all())
        sums = unstaged.float().view(-1, 16)
        seq = torch.zeros(inter // 16, device=dev)
        for j in range(16):
            seq = seq + sums[:, j]                                                   # sequential fp32 sum in element order
        assert torch.equal(act_sum, seq)
        got = packing.linear_staged(act, act_sum, down, residual=h)
        assert torch.equal(got, want), (it, (got.float() - want.float()).abs().max().item())


def test_decode_gemv_launches_full_size_vs_oracle(dev):
    """Every GEMV launch of ONE full-size decode layer (hidden 4096, intermediate 11008), fused prologue / residual included,
    against the ORACLE: weights quantised by oracle/mxq_oracle.py (the kernel's packed form is checked bit-equal to it
    first), y = O.linear_ref on the prologue's fp16 input, <= 1e-3 of the output scale (VERDICT r4 weak #1b: the stage test
    compares with a torch restatement at 2e-2)."""
    from mxq_amd import packing
    H, I = 4096, 11008
    rng = np.random.default_rng(11)
    g = torch.Generator().manual_seed(12)

    def oracle_weight(N, K, seed):
        W16 = (torch.randn(N, K, generator=torch.Generator().manual_seed(seed)) * 0.02).half()
        ref = O.mxq_quantize(W16.numpy())
        p = packing.quantize_pack(W16.to(dev))
        w16 = ref["w_deq32"].astype(np.float16)
        assert np.array_equal(packing.dequant(p).cpu().numpy().view(np.uint16), w16.view(np.uint16))
        return p, w16
    norm_w = (1.0 + 0.1 * torch.randn(H, generator=g)).half()
    h = torch.randn(1, H, generator=g).half()
    res = torch.randn(1, H, generator=g).half()

    def check(y, ref, what):
        err = np.abs(y.float().cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err <= REL_TOL, (what, err)

    def rms_in(x16):            # the RMSNorm prologue: x <- fp16(x * norm_w); the row's rsqrt(mean x^2 + eps) multiplies the outputs
        xf = x16.float().numpy()
        xin = (xf * norm_w.float().numpy()).astype(np.float16)
        return xin, np.float32(1.0) / np.sqrt((xf.astype(np.float64) ** 2).mean() + 1e-5).astype(np.float32)
    # q|k|v (RMSNorm prologue) and gate|up (RMSNorm prologue): concatenated weights, as the decode stage launches them
    for name, parts, seed in (("qkv", [(H, H)] * 3, 100), ("gate_up", [(I, H)] * 2, 200)):
        ps, ws = zip(*[oracle_weight(N, K, seed + i) for i, (N, K) in enumerate(parts)])
        xin, scale = rms_in(h)
        ref = np.concatenate([O.linear_ref(xin, w) for w in ws], axis=1) * scale
        y = packing.linear_fused(h.to(dev), packing.concat_packed(ps), 1, norm_w.to(dev))
        check(y, ref.astype(np.float16).astype(np.float32), name)
        if name == "gate_up":
            gu16 = y.cpu()
    # o_proj: plain input + residual
    p, w = oracle_weight(H, H, 300)
    a = torch.randn(1, H, generator=g).half()
    ref = (res.float().numpy() + O.linear_ref(a.numpy(), w).astype(np.float16).astype(np.float32))
    check(packing.linear_fused(a.to(dev), p, 0, residual=res.to(dev)), ref, "o_proj")
    # down_proj: SwiGLU prologue on the kernel's own gate|up output + residual
    p, w = oracle_weight(H, I, 400)
    gf, uf = gu16[:, :I].float().numpy(), gu16[:, I:].float().numpy()
    act = ((gf / (1.0 + np.exp(-gf))).astype(np.float16).astype(np.float32) * uf).astype(np.float16)
    ref = res.float().numpy() + O.linear_ref(act, w).astype(np.float16).astype(np.float32)
    check(packing.linear_fused(gu16.to(dev), p, 2, residual=res.to(dev)), ref, "down_proj")


@pytest.mark.parametrize("M,IC,OC,G,S", [(16, 4096, 4096, 128, 8), (128, 4096, 4096, 128, 4), (2048, 4096, 4096, 64, 1),
                                         (16, 4096, 11008, 128, 8), (128, 4096, 11008, 32, 2), (2048, 4096, 11008, 128, 1),
                                         (1, 256, 64, 32, 1), (130, 320, 192, 64, 3),
                                         # round 6, one case per schedule of the rebuilt kernel (csrc/capi.hip mxq_gemm_awq_f16):
                                         (2048, 11008, 4096, 128, 1),   # 256-token tiles, 172 K-steps (not a multiple of the burst of 3)
                                         (1000, 4096, 11008, 128, 1),   # 256-token tiles, ragged token tile, stream-K tail (344 tiles)
                                         (512, 4096, 4096, 128, 1),     # 128-token tiles, 128 of them: stream-K, the tail always split
                                         (300, 11008, 4096, 64, 1),     # 128-token tiles, 96 of them, 172 K-steps, ragged tokens
                                         (64, 4096, 11008, 128, 1),     # 64-token tiles, 86 of them: stream-K
                                         (48, 11008, 4096, 32, 1),      # 64-token tiles, 32 of them: K slices + combine launch
                                         # <= 32 tokens: the streaming kernel (csrc/skinny_awq.hip)
                                         (32, 4096, 4096, 64, 1),       # two token blocks, 8 K slices per channel block
                                         (20, 11008, 4096, 128, 1),     # ragged tokens, 172 K-steps over 8 slices (the last one shorter)
                                         (7, 320, 192, 32, 1)])         # half a channel block at the edge, 5 K-steps, G = 32
def test_gemm_forward_cuda_reference_operands_vs_oracle(dev, M, IC, OC, G, S):
    """`gemm_forward_cuda(in_feats, kernel, scaling_factors, zeros, split_k_iters)` on the reference's
    operand format (gemm_cuda.h:3-4; K-major int32 words of 8 interleaved nibbles, group-G scales [IC/G, OC], packed
    integer zeros) against the oracle's restatement: <= 1e-3 (max-norm and Frobenius) of the fp32-accumulated product on
    the fp16 weight fp16((q - z) * s).  Parity UNPINNED: the reference never compiles this kernel."""
    import mxq_inference_engine as eng
    rng = np.random.default_rng(M + IC + OC + G)
    q = rng.integers(0, 16, size=(IC, OC))
    z = rng.integers(0, 16, size=(IC // G, OC))
    s = (rng.random((IC // G, OC)) * 0.004 + 0.001).astype(np.float16)
    x = (rng.standard_normal((M, IC))).astype(np.float16)
    kern, zw = O.gemm_awq_pack(q), O.gemm_awq_pack(z)
    y = eng.gemm_forward_cuda(torch.from_numpy(x).to(dev), torch.from_numpy(kern).to(dev), torch.from_numpy(s).to(dev),
                              torch.from_numpy(zw).to(dev), S)
    assert y.shape == (M, OC) and y.dtype == torch.float16
    _check_gemm(y.cpu().numpy(), O.gemm_awq_ref(x, kern, s, zw, G), f"awq gemm {M}x{IC}x{OC} G{G} S{S}")
    # deterministic (every sum in a fixed order, whatever the K schedule -- slices, stream-K, last arriver), and independent of
    # the launcher's split_k_iters argument, which this implementation only validates
    y2 = eng.gemm_forward_cuda(torch.from_numpy(x).to(dev), torch.from_numpy(kern).to(dev), torch.from_numpy(s).to(dev),
                               torch.from_numpy(zw).to(dev), S + 3)
    assert torch.equal(y, y2)
    from mxq_amd import packing
    packing.workspace_status(dev)


def test_gemm_awq_c_entry_without_a_workspace(dev):
    """`mxq_gemm_awq_f16` with `workspace == NULL` (a direct C caller): nothing may split K across workgroups -- the streaming
    kernel takes the call when the whole K range fits its LDS (IC = 1024), else the tile kernel runs on whole tiles (IC = 4096);
    same results as with a workspace, to accumulation order."""
    from mxq_amd import _lib, packing
    lib = _lib.load()
    rng = np.random.default_rng(5)
    st = torch.cuda.current_stream().cuda_stream
    for M, IC, OC, G in ((16, 1024, 256, 64), (16, 4096, 256, 128), (100, 1024, 256, 64)):
        q, z = rng.integers(0, 16, size=(IC, OC)), rng.integers(0, 16, size=(IC // G, OC))
        s = (rng.random((IC // G, OC)) * 0.004 + 0.001).astype(np.float16)
        x = rng.standard_normal((M, IC)).astype(np.float16)
        t = lambda a: torch.from_numpy(a).to(dev)
        kern, zw, xd, sd = t(O.gemm_awq_pack(q)), t(O.gemm_awq_pack(z)), t(x), t(s)
        y = torch.empty(M, OC, dtype=torch.float16, device=dev)
        _lib.check(lib.mxq_gemm_awq_f16(xd.data_ptr(), kern.data_ptr(), sd.data_ptr(), zw.data_ptr(), y.data_ptr(), M, IC, OC, G,
                                        None, 0, st), "mxq_gemm_awq_f16 without a workspace")
        _check_gemm(y.cpu().numpy(), O.gemm_awq_ref(x, kern.cpu().numpy(), s, zw.cpu().numpy(), G), f"awq, no workspace {M}x{IC}x{OC}")


def test_gemm_forward_cuda_integer_exact_and_rejections(dev):
    """Operand mapping exactly (small integers: every product and sum exact in fp16 / fp32) and the launcher's rejections
    (gemm_cuda_gen.cu:447-454) as ValueError with its messages."""
    import mxq_inference_engine as eng
    rng = np.random.default_rng(3)
    M, IC, OC, G = 20, 128, 128, 32
    q = rng.integers(0, 16, size=(IC, OC)); z = rng.integers(0, 16, size=(IC // G, OC))
    s = rng.integers(1, 3, size=(IC // G, OC)).astype(np.float16)
    x = rng.integers(-2, 3, size=(M, IC)).astype(np.float16)
    kern, zw = O.gemm_awq_pack(q), O.gemm_awq_pack(z)
    t = lambda a: torch.from_numpy(a).to(dev)
    for S in (1, 2):
        y = eng.gemm_forward_cuda(t(x), t(kern), t(s), t(zw), S).float().cpu().numpy()
        assert np.array_equal(y, O.gemm_awq_ref(x, kern, s, zw, G)), S
    with pytest.raises(ValueError, match="cta_N = 64"):
        eng.gemm_forward_cuda(t(x), t(kern[:, :12].copy()), t(s[:, :96].copy()), t(zw[:, :12].copy()), 1)
    with pytest.raises(ValueError, match="multiple of 32"):
        eng.gemm_forward_cuda(t(x), t(kern), t(np.ones((8, OC), np.float16)), t(np.zeros((8, OC // 8), np.int32)), 1)
    with pytest.raises(ValueError, match="Group size"):      # OC % group_size: 192 % 128
        eng.gemm_forward_cuda(t(x), t(np.zeros((IC, 24), np.int32)), t(np.ones((1, 192), np.float16)), t(np.zeros((1, 24), np.int32)), 1)
    with pytest.raises(TypeError):
        eng.gemm_forward_cuda(t(x).float(), t(kern), t(s), t(zw), 1)


def test_decode_token_graph_matches_eager_loop(dev):
    """The one-graph-per-token loop (embedding -> layers -> head -> argmax captured together) must generate
    exactly the tokens of the step-by-step loop through LayerPipeline.decode on the same stage."""
    from mxq_amd.llama_decode import DecodeStage
    from mxq_amd.pipeline import LayerPipeline
    st = DecodeStage(range(2), dev, max_ctx=64, first=True, last=True, hidden=256, inter=704, heads=2, vocab=512)
    tbuf = torch.zeros(1, dtype=torch.int64, device=dev)
    hbuf = torch.zeros(1, 256, device=dev, dtype=torch.float16)
    pipe = LayerPipeline(0, 1)

    def stage_fn(h, step):
        out = st.step(h)
        st.advance()
        return out
    st.reset()
    eager = pipe.decode(3, 12, st.embed_token, stage_fn, st.head, hbuf, tbuf)
    st.capture_token_loop(tbuf)
    st.reset()
    graphed = st.decode_tokens(tbuf, 3, 12)
    assert graphed == eager and len(set(graphed)) > 1
    assert int(st.pos.item()) == 12


@pytest.mark.parametrize("max_ctx,splits", [(256, 8), (1024, 16)])
def test_decode_stage_long_context_graphs_match_eager_across_the_split_switch(dev, max_ctx, splits):
    """the configuration that ships for long contexts -- `attn_splits` 8 / 16, a short and a long captured
    graph switched by `_pick` at 128 keys, ONE split workspace shared by all layers -- decodes 150 tokens (positions 0..149:
    the one-workgroup launch up to 128 keys, the split launch beyond) three ways on the same stage weights: the eager
    `step` + `advance` loop, the per-stage graph (`capture` / `step_graph`) and the one-graph-per-token loop
    (`capture_token_loop` / `decode_tokens`).  Same kernels in the same order: the ids must be identical."""
    from mxq_amd.llama_decode import DecodeStage
    n = 150

    def mk():
        st = DecodeStage(range(2), dev, max_ctx=max_ctx, first=True, last=True, hidden=256, inter=704, heads=2, vocab=512)
        assert st.fused and st.attn_splits == splits
        return st
    st = mk()
    tok = torch.full((1,), 3, dtype=torch.int64, device=dev)
    eager, used_long = [], []
    for _ in range(n):
        h = st.step(st.embed_token(tok))
        used_long.append(st._long_ctx)
        st.advance()
        tok = st.head(h).reshape(-1)[:1].clone()
        eager.append(int(tok.item()))
    assert used_long == [i + 1 > 128 for i in range(n)]            # the split launch really ran from the 129th key on
    st2 = mk().capture()
    assert st2._graph_long is not None
    tok = torch.full((1,), 3, dtype=torch.int64, device=dev)
    staged = []
    for _ in range(n):
        h = st2.step_graph(st2.embed_token(tok))
        tok = st2.head(h).reshape(-1)[:1].clone()
        staged.append(int(tok.item()))
    st3 = mk()
    tbuf = torch.zeros(1, dtype=torch.int64, device=dev)
    st3.capture_token_loop(tbuf)
    assert st3._tgraph_long is not None
    st3.reset()
    looped = st3.decode_tokens(tbuf, 3, n)
    first_bad = next((i for i, (a, b, c) in enumerate(zip(eager, staged, looped)) if not a == b == c), None)
    assert first_bad is None, (first_bad, eager[first_bad], staged[first_bad], looped[first_bad])
    assert len(set(eager)) > 1 and int(st3.pos.item()) == n
    assert int(st3._attn_ws[:1024].view(torch.int32).abs().sum().item()) == 0      # arrival counters left zeroed


def test_g5_decoder_block_fwd_bwd(dev, g5):
    """A Llama decoder layer built from this repo's QuantizeLinear (w_bits=2, a_bits=16) must
    reproduce the reference's LlamaDecoderLayer (LLM-QAT/models/modeling_llama_quant.py:414-469)
    output, input gradient and all seven weight gradients on the golden block G5.  The layer
    scaffolding (RMSNorm, rotary, attention) is restated here with plain torch ops; only the
    quantised Linears are the product code."""
    from mxq_amd.utils_quant import QuantizeLinear, SymQuantizer
    H, I, heads = 256, 704, 4
    hd = H // heads
    names = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj",
             "mlp.up_proj", "mlp.down_proj"]
    lin = {}
    for n in names:
        w = torch.from_numpy(g5[f"sd_{n}.weight"]).to(dev)
        m = QuantizeLinear(w.shape[1], w.shape[0], w_bits=2, a_bits=16).to(dev)
        m.weight.data = w.clone()
        lin[n] = m
    ln1 = torch.from_numpy(g5["sd_input_layernorm.weight"]).to(dev)
    ln2 = torch.from_numpy(g5["sd_post_attention_layernorm.weight"]).to(dev)
    inv_freq = torch.from_numpy(g5["sd_self_attn.rotary_emb.inv_freq"]).to(dev)
    x = torch.from_numpy(g5["x"]).to(dev).requires_grad_()
    pos = torch.from_numpy(g5["pos"]).to(dev)
    B, S, _ = x.shape

    def rms(h, w, eps=1e-6):
        return w * (h * torch.rsqrt(h.float().pow(2).mean(-1, keepdim=True) + eps))

    def rot_half(t):
        return torch.cat((-t[..., hd // 2:], t[..., : hd // 2]), dim=-1)

    freqs = torch.einsum("i,j->ij", torch.arange(64, device=dev).float(), inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos()[pos].unsqueeze(1), emb.sin()[pos].unsqueeze(1)          # [B, 1, S, hd]
    mask = torch.full((S, S), torch.finfo(torch.float32).min, device=dev).triu(1)[None, None]

    h = rms(x, ln1)
    q = lin["self_attn.q_proj"](h).view(B, S, heads, hd).transpose(1, 2)
    # KV fake-quant (kv_bits = 16 in G5): SymQuantizer with clip [-2, 2] (modeling_llama_quant.py:322-329)
    kv_clip = torch.tensor([-2.0, 2.0])
    k = SymQuantizer.apply(lin["self_attn.k_proj"](h), kv_clip, 16, False).view(B, S, heads, hd).transpose(1, 2)
    v = SymQuantizer.apply(lin["self_attn.v_proj"](h), kv_clip, 16, False).view(B, S, heads, hd).transpose(1, 2)
    q, k = q * cos + rot_half(q) * sin, k * cos + rot_half(k) * sin
    att = torch.matmul(q, k.transpose(2, 3)) / hd ** 0.5 + mask
    att = torch.max(att, torch.tensor(torch.finfo(att.dtype).min, device=dev))
    att = torch.softmax(att, dim=-1, dtype=torch.float32)
    a = torch.matmul(att, v).transpose(1, 2).reshape(B, S, H)
    h1 = x + lin["self_attn.o_proj"](a)
    h2 = rms(h1, ln2)
    y = h1 + lin["mlp.down_proj"](torch.nn.functional.silu(lin["mlp.gate_proj"](h2)) * lin["mlp.up_proj"](h2))
    y.backward(torch.from_numpy(g5["gy"]).to(dev))

    def close(got, ref, what, tol=2e-4):
        ref = torch.from_numpy(ref).to(dev)
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        assert err <= tol, (what, err)

    close(y.detach(), g5["y"], "y")
    close(x.grad, g5["dx"], "dx")
    for n in names:
        close(lin[n].weight.grad, g5[f"grad_{n}.weight"], n)


# ----------------------------------------------------------------------------------------
# uniform W2 / W4 layouts (config-5 sweep arms)
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout,pre", [("w2g16", "w2"), ("w4row", "w4")])
def test_g8_uniform_quantize_unpack_dequant(dev, layout, pre):
    from mxq_amd import packing
    from tests.conftest import load_golden
    g8 = load_golden("g8_uniform.npz")
    p = packing.quantize_pack_uniform(torch.from_numpy(g8["W"]).to(dev), layout)
    w16, got = packing.expand_uniform(p)
    for k in ("codes", "sc", "zero", "qs", "qz"):
        assert np.array_equal(got[k].cpu().numpy(), g8[f"{pre}_{k}"]), k
    assert np.array_equal(w16.cpu().numpy().view(np.uint16), g8[f"{pre}_wdeq"].view(np.uint16))


@pytest.mark.parametrize("layout", ["w2g16", "w4row"])
@pytest.mark.parametrize("M,N,K", [(1, 64, 256), (3, 256, 704), (4, 4096, 4096), (1, 4096, 11008)])
def test_uniform_gemv_vs_oracle(dev, layout, M, N, K):
    """The decode GEMV template on the uniform W2 / W4 layouts (BASELINE config 5 arms), against the oracle's
    dequantised weight; ragged K (704 = 11 chunks) exercises the partial last tile."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(N + K + M)
    W16 = (torch.randn(N, K, generator=g) * 0.02).half()
    ref = O.uniform_quantize(W16.numpy(), layout)
    p = packing.quantize_pack_uniform(W16.to(dev), layout)
    x = torch.randn(M, K, generator=g).half()
    y = packing.linear_layout(x.to(dev), p, path="gemv").cpu().numpy()
    _check_gemm(y, O.linear_ref(x.numpy(), ref["w_deq32"].astype(np.float16)), f"gemv {layout} {M}x{N}x{K}")
    ya = packing.linear_layout(x.to(dev), p, path="auto").cpu().numpy()
    assert np.array_equal(ya, y)
    with pytest.raises(ValueError):
        packing.linear_layout(torch.zeros(5, K, dtype=torch.float16, device=dev), p, path="gemv")


@pytest.mark.parametrize("layout", ["w2g16", "w4row"])
@pytest.mark.parametrize("M,N,K", [(5, 64, 128), (16, 256, 704), (17, 4096, 4096), (33, 16, 64), (48, 11008, 4096), (64, 144, 192)])
def test_uniform_skinny_vs_oracle(dev, layout, M, N, K):
    """csrc/skinny.hip on the uniform layouts (W2G16: both K slices of a chunk are 2-bit groups; W4ROW: one 4-bit code word
    per lane and slice): against the oracle's dequantised weight, 1 .. 4 token blocks, ragged K; and the library's
    dispatch takes it up to 48 tokens (the reference serves uniform W4 at any batch, gemv_cuda.cu:346-399)."""
    from mxq_amd import packing
    g = torch.Generator().manual_seed(N + K + M)
    W16 = (torch.randn(N, K, generator=g) * 0.02).half()
    ref = O.uniform_quantize(W16.numpy(), layout)
    p = packing.quantize_pack_uniform(W16.to(dev), layout)
    x = torch.randn(M, K, generator=g).half()
    y = packing.linear_layout(x.to(dev), p, path="skinny")
    _check_gemm(y.cpu().numpy(), O.linear_ref(x.numpy(), ref["w_deq32"].astype(np.float16)), f"skinny {layout} {M}x{N}x{K}")
    if M <= 48:
        assert torch.equal(packing.linear_layout(x.to(dev), p, path="auto"), y), "auto dispatch should be the skinny kernel here"
    else:
        _check_gemm(packing.linear_layout(x.to(dev), p, path="auto").cpu().numpy(),
                    O.linear_ref(x.numpy(), ref["w_deq32"].astype(np.float16)), f"auto {layout} {M}x{N}x{K}")


def test_uniform_skinny_integer_exact(dev):
    """Small-integer weights and activations: every product and sum exact, so a wrong lane -> (group, half / code word)
    mapping of the uniform layouts in the skinny kernel is an exact mismatch."""
    from mxq_amd import packing
    rng = np.random.default_rng(12)
    N, K, M = 64, 256, 24
    for layout in ("w2g16", "w4row"):
        W = rng.integers(0, 4 if layout == "w2g16" else 16, (N, K)).astype(np.float16)
        # every group / row spans the full code range, so the quantiser reproduces the integers exactly
        if layout == "w2g16":
            W.reshape(N, K // 16, 16)[:, :, 0] = 0; W.reshape(N, K // 16, 16)[:, :, 1] = 3
        else:
            W[:, 0] = 0; W[:, 1] = 15
        p = packing.quantize_pack_uniform(torch.from_numpy(W).to(dev), layout)
        w16, _ = packing.expand_uniform(p, codes=False)
        assert np.array_equal(w16.cpu().numpy(), W), layout
        x = rng.integers(-2, 3, (M, K)).astype(np.float16)
        yref = x.astype(np.float32) @ W.astype(np.float32).T
        y = packing.linear_layout(torch.from_numpy(x).to(dev), p, path="skinny").cpu().numpy().astype(np.float32)
        assert np.array_equal(y, yref), layout


@pytest.mark.parametrize("layout", ["w2g16", "w4row"])
@pytest.mark.parametrize("M,N,K", [(300, 144, 192), (512, 256, 1024), (64, 4096, 4096)])
def test_uniform_gemm_vs_oracle(dev, layout, M, N, K):
    from mxq_amd import packing
    g = torch.Generator().manual_seed(M + N + K)
    W = (torch.randn(N, K, generator=g) * 0.02).half()
    x = torch.randn(M, K, generator=g).half()
    ref = O.uniform_quantize(W.numpy(), layout)
    p = packing.quantize_pack_uniform(W.to(dev), layout)
    w16, got = packing.expand_uniform(p)
    assert np.array_equal(got["codes"].cpu().numpy(), ref["codes"])
    w_ref16 = ref["w_deq32"].astype(np.float16)
    assert np.array_equal(w16.cpu().numpy().view(np.uint16), w_ref16.view(np.uint16))
    y = packing.linear_layout(x.to(dev), p).cpu().numpy()
    _check_gemm(y, O.linear_ref(x.numpy(), w_ref16), f"{layout} {M}x{N}x{K}")
    if K >= 1024:   # rowmeta (16 B/row) is amortised over the row
        assert abs(p.bits_per_weight() - (4.5 if layout == "w2g16" else 4.0)) < 0.2
