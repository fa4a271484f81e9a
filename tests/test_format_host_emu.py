"""CPU checks of the packed format and the register-level dequant helpers: the SAME
inline functions the HIP kernels use (csrc/mxq_format.h, mxq_pack.h, mxq_dequant.h) are
compiled for the host with a software v_perm_b32 (tests/host_emu.cpp) and compared with
the oracle / golden vectors.  No GPU needed."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import mxq_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("codes2", "sc2", "zero2", "qs2", "qz2", "codes4", "sc4", "zero4", "qs4", "qz4")


@pytest.fixture(scope="session")
def emu():
    out = os.path.join(ROOT, "tests", "_build", "libhost_emu.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-mf16c", "-ffp-contract=off", "-shared", "-fPIC",
                           os.path.join(ROOT, "tests", "host_emu.cpp"), "-o", out])
    lib = ctypes.CDLL(out)
    lib.emu_qweight_dwords.restype = ctypes.c_long
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _pack(emu, p, N, K):
    arrs = [np.ascontiguousarray(p[k]) for k in KEYS]
    qw = np.zeros(emu.emu_qweight_dwords(N, K), np.uint32)
    rm = np.zeros((N, 4), np.float32)
    emu.emu_pack(*[_ptr(a) for a in arrs], _ptr(qw), _ptr(rm), N, K)
    return qw, rm


def _roundtrip_and_dequant(emu, p, N, K, w_ref16):
    qw, rm = _pack(emu, p, N, K)
    assert qw.nbytes == (N // 16) * (K // 64) * 576
    c2 = np.zeros_like(p["codes2"]); sc = np.zeros_like(p["sc2"]); z2 = np.zeros_like(p["zero2"])
    c4 = np.zeros_like(p["codes4"])
    emu.emu_unpack(_ptr(qw), _ptr(c2), _ptr(sc), _ptr(z2), _ptr(c4), N, K)
    assert np.array_equal(c2, p["codes2"]) and np.array_equal(c4, p["codes4"])      # integer unpack bit-exact
    assert np.array_equal(sc, p["sc2"]) and np.array_equal(z2.view(np.uint32), p["zero2"].view(np.uint32))
    out = np.zeros((N, K), np.uint16)
    emu.emu_dequant_f16(_ptr(qw), _ptr(rm), _ptr(out), N, K)
    assert np.array_equal(out, w_ref16.view(np.uint16))                               # LUT / perm dequant bit-exact


def test_g1_pack_unpack_dequant(emu, g1):
    p = {k: g1[k] for k in KEYS}
    _roundtrip_and_dequant(emu, p, 64, 256, g1["w_deq"])


@pytest.mark.parametrize("N,K,seed", [(32, 64, 0), (48, 320, 1), (16, 704, 2), (128, 1024, 3)])
def test_random_vs_oracle(emu, N, K, seed):
    """K = 320 / 704: chunk counts that are not a multiple of 4 (ragged GEMV tail)."""
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16)
    W[0, :16] = 0.5          # constant group
    p = O.mxq_quantize(W)
    _roundtrip_and_dequant(emu, p, N, K, p["w_deq32"].astype(np.float16))


def test_all_code_values_and_bit_positions(emu):
    """Every 2-bit / 4-bit code value in every element position, adversarial zero-points."""
    N, K = 16, 64
    p = O.mxq_quantize(np.zeros((N, K), np.float16))
    rng = np.random.default_rng(7)
    p["codes2"] = rng.integers(0, 4, (N, 48), dtype=np.uint8)
    p["codes4"] = rng.integers(0, 16, (N, 16), dtype=np.uint8)
    for k in range(16):
        p["codes2"][k, :] = 0; p["codes2"][k, k] = 3; p["codes2"][k, 16 + k] = 2; p["codes2"][k, 32 + k] = 1
        p["codes4"][k, :] = 0; p["codes4"][k, k] = 15
    p["sc2"] = rng.integers(0, 16, (N, 3), dtype=np.uint8)
    p["sc4"] = rng.integers(0, 16, (N,), dtype=np.uint8)
    p["zero2"] = (rng.standard_normal((N, 3)) * 3).astype(np.float32)
    p["zero4"] = (rng.standard_normal((N,)) * 9).astype(np.float32)
    p["qs2"] = np.abs(rng.standard_normal((1, 3))).astype(np.float32) * 1e-3
    p["qz2"] = -np.abs(rng.standard_normal((1, 3))).astype(np.float32) * 5
    p["qs4"] = np.array([3e-4], np.float32); p["qz4"] = np.array([-7.25], np.float32)
    _roundtrip_and_dequant(emu, p, N, K, O.mxq_dequant(p).astype(np.float16))


# ----------------------------------------------------------------------------------------
# property test: ANY parameter set (not only ones a quantiser would produce) survives pack -> unpack
# bit for bit and dequantises to the oracle's fp32 formula rounded once to fp16
# ----------------------------------------------------------------------------------------
from hypothesis import HealthCheck, given, settings, strategies as st   # noqa: E402


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2**32 - 1), nc=st.integers(1, 5), rb=st.integers(1, 3),
       zscale=st.sampled_from([1e-3, 1.0, 37.0, 1e4]), sscale=st.sampled_from([1e-6, 3e-4, 0.02, 5.0]))
def test_property_pack_unpack_dequant(emu, seed, nc, rb, zscale, sscale):
    N, K = 16 * rb, 64 * nc
    rng = np.random.default_rng(seed)
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
    # a few exact edge values: zero scale, zero zero-point, negative-zero
    p["zero2"][0, 0] = 0.0
    p["zero2"][-1, -1] = -0.0
    p["qs2"][0, 0] = 0.0
    with np.errstate(over="ignore", invalid="ignore"):
        w16 = O.mxq_dequant(p).astype(np.float16)       # overflow to inf is part of the contract (one rounding)
    _roundtrip_and_dequant(emu, p, N, K, w16)
