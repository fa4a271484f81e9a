"""The C-ABI library builds, loads, and exports every symbol include/mxq_hip.h declares.
No compute calls (CPU-only box)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from mxq_amd import _lib
    return _lib.load()


def _declared():
    text = open(os.path.join(ROOT, "include", "mxq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mxq_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from mxq_amd import _lib
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, f"{n} is declared in the header but not bound in mxq_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_size_helpers(lib):
    assert lib.mxq_version() >> 16 == 1
    assert lib.mxq_qweight_bytes(4096, 4096) == 256 * 64 * 576
    assert lib.mxq_qweight_bytes(4096, 11008) == 256 * 172 * 576
    assert lib.mxq_qweight_bytes(64, 320) == 4 * 5 * 576
    assert lib.mxq_qweight_bytes(100, 4096) == 0 and lib.mxq_qweight_bytes(4096, 100) == 0
    assert lib.mxq_rowmeta_bytes(4096) == 4096 * 16
    bits = 8.0 * (lib.mxq_qweight_bytes(4096, 4096) + lib.mxq_rowmeta_bytes(4096)) / 4096 ** 2
    assert 4.5 < bits < 4.55


def test_argument_validation_without_gpu(lib):
    """Rejected arguments return negative codes before anything touches the device."""
    assert lib.mxq_linear_f16(None, None, None, None, 1, 4096, 4096, None) == -2
    assert lib.mxq_linear_f16(16, 16, 16, 16, 1, 4095, 4096, None) == -1
    assert lib.mxq_linear_f16(16, 16, 16, 8, 1, 4096, 4096, None) == -4
    assert lib.mxq_gemv_f16(16, 16, 16, 16, 5, 4096, 4096, None) == -1
    assert lib.mxq_fakequant_fwd(16, 16, 4, 100, 2, 2, None) == -1
    assert lib.mxq_fakequant_fwd(16, 16, 4, 128, 2, 7, None) == -3
    assert lib.mxq_gemv_awq_f16(16, 16, 16, 16, 16, 1, 4096, 4096, 16, None) == -1
    assert lib.mxq_gemv_proto_f16(*([16] * 9), 1, 2048, 4096, 16, None) == -1
    assert lib.mxq_quantize_pack(16, 9, None, 16, 16, 64, 64, None) == -3
    assert lib.mxq_dense_f16(None, 16, 16, 8, 256, 128, 0, None) == -2          # nn.Linear on the fake-quant weight
    assert lib.mxq_dense_f16(16, 16, 16, 8, 250, 128, 0, None) == -1
    assert lib.mxq_dense_f16(16, 16, 8, 8, 256, 128, 0, None) == -4
    assert lib.mxq_dense_f16(16, 16, 16, 8, 256, 128, 3, None) == -1            # unknown kernel variant


def test_argument_validation_rejects_before_any_launch(lib):
    """Every entry point validates its arguments first and returns a negative MXQ_E_* code without
    touching the (fake) pointers or the device -- what the Python wrappers turn into ValueError, like the
    reference GEMM's std::invalid_argument (gemm_cuda_gen.cu:447-454).  No kernel is launched here."""
    E_SHAPE, E_NULL, E_DTYPE, E_ALIGN = -1, -2, -3, -4
    P, Q = 0x10000, 0x20008          # fake device addresses: 16-byte aligned / 8-byte aligned only
    # the Linear family: x, qweight, rowmeta, y, M, N, K, stream
    for fn in (lib.mxq_linear_f16, lib.mxq_gemm_f16, lib.mxq_gemv_f16):
        assert fn(None, P, P, P, 8, 64, 256, None) == E_NULL
        assert fn(P, P, P, P, 8, 60, 256, None) == E_SHAPE          # N % 16
        assert fn(P, P, P, P, 8, 64, 200, None) == E_SHAPE          # K % 64
        assert fn(P, P, P, P, 0, 64, 256, None) == E_SHAPE          # no tokens
        assert fn(P, Q, P, P, 2, 64, 256, None) == E_ALIGN
    assert lib.mxq_gemv_f16(P, P, P, P, 5, 64, 256, None) == E_SHAPE   # GEMV is for <= 4 tokens
    # unknown kernel variants are rejected, not dispatched (no profiling build is reachable through this ABI)
    for variant in (2, 3, 4, 5, 6, 7, 11, 15, 18, 64, 999, -1):
        assert lib.mxq_gemm_f16_ws(P, P, P, P, 8, 64, 256, variant, None, 0, None) == E_SHAPE
    assert lib.mxq_linear_f16_ws(P, P, P, P, 8, 64, 256, Q, 1 << 27, None) == E_ALIGN
    # the layout-aware dispatch (mid-M split-K kernel behind it): same checks, plus the layout code
    assert lib.mxq_linear_f16_layout_ws(P, P, P, P, 100, 64, 256, 7, P, 1 << 27, None) == E_SHAPE       # unknown layout
    assert lib.mxq_linear_f16_layout_ws(P, P, P, P, 100, 64, 256, 3, Q, 1 << 27, None) == E_ALIGN
    assert lib.mxq_linear_f16_layout_ws(P, None, P, P, 100, 64, 256, 0, P, 1 << 27, None) == E_NULL
    assert lib.mxq_gemm_f16_ws(P, P, P, P, 100, 60, 256, 10, P, 1 << 27, None) == E_SHAPE              # variant 10 = mid-M kernel: N % 16
    assert lib.mxq_gemm_f16_ws(P, P, P, None, 8, 64, 256, 0, None, 0, None) == E_NULL
    assert lib.mxq_gemv_fused_f16(P, P, P, P, 64, 256, 1, None, 1e-5, None, None) == E_NULL   # RMSNorm needs its weight
    assert lib.mxq_gemv_fused_f16(P, P, P, P, 64, 256, 7, None, 1e-5, None, None) == E_SHAPE
    # packing
    assert lib.mxq_quantize_pack(P, 9, None, P, P, 64, 256, None) == E_DTYPE
    assert lib.mxq_quantize_pack(P, 1, None, None, P, 64, 256, None) == E_NULL
    assert lib.mxq_dequant_f16(P, P, Q, 64, 256, None) == E_ALIGN
    assert lib.mxq_quantize_pack_layout(P, 1, P, P, 64, 256, 5, None) == E_SHAPE     # unknown layout
    # fake quant
    assert lib.mxq_fakequant_fwd(P, P, 16, 100, 2, 2, None) == E_SHAPE                # cols % 64
    assert lib.mxq_fakequant_fwd(P, P, 16, 128, 2, 7, None) == E_DTYPE
    assert lib.mxq_fakequant_fwd(P, None, 16, 128, 2, 2, None) == E_NULL
    assert lib.mxq_actquant_group_fwd(P, P, 4, 200, 48, 8, 1, 2, None) == E_SHAPE     # group / 8 not a power of two
    assert lib.mxq_actquant_group_fwd(P, P, 4, 204, 128, 8, 1, 2, None) == E_SHAPE    # cols % 8
    assert lib.mxq_actquant_fwd(P, P, None, 4, 256, 1, 1, 8, 1, 2, None) == E_NULL
    assert lib.mxq_actquant_fwd(P, P, P, 4, 250, 1, 1, 8, 1, 2, None) == E_SHAPE
    # the reference extension's formats
    assert lib.mxq_gemv_awq_f16(P, P, P, P, P, 1, 4096, 4096, 48, None) == E_SHAPE    # group size
    assert lib.mxq_gemv_awq_f16(P, Q, P, P, P, 1, 4096, 4096, 128, None) == E_ALIGN   # codes are read 16 bytes at a time
    assert lib.mxq_gemv_awq_f16(P, P, P, P, P, 1, 4104, 4096, 8, None) == E_SHAPE     # IC % 32 (rows of whole 16-B units)
    # the reference GEMM's operand format: the launcher's rejections (gemm_cuda_gen.cu:447-454) + this kernel's K-step
    assert lib.mxq_gemm_awq_f16(P, P, P, P, P, 16, 4096, 4000, 128, None, 0, None) == E_SHAPE     # OC % 64
    assert lib.mxq_gemm_awq_f16(P, P, P, P, P, 16, 4096, 4096, 48, None, 0, None) == E_SHAPE      # group size % 32
    assert lib.mxq_gemm_awq_f16(P, P, P, P, P, 16, 4096, 4160, 128, None, 0, None) == E_SHAPE     # OC % group size
    assert lib.mxq_gemm_awq_f16(P, P, P, P, P, 16, 4128, 4096, 32, None, 0, None) == E_SHAPE      # IC % 64
    assert lib.mxq_gemm_awq_f16(P, P, P, P, P, 16, 4096, 4096, 128, P, 4096, None) == E_SHAPE     # a workspace without its counter head
    assert lib.mxq_gemm_awq_f16(P, P, P, P, P, 16, 4096, 4096, 128, Q, 1 << 20, None) == E_ALIGN  # workspace alignment
    assert lib.mxq_gemm_awq_f16(P, P, None, P, P, 16, 4096, 4096, 128, None, 0, None) == E_NULL
    assert lib.mxq_gemm_awq_f16(Q, P, P, P, P, 16, 4096, 4096, 128, None, 0, None) == E_ALIGN
    assert not hasattr(lib, "mxq_prefetch")                                           # round-2 experiment: a record, not ABI
    assert lib.mxq_gemv_proto_f16(P, P, P, P, P, P, P, P, P, 1, 2048, 4096, 16, None) == E_SHAPE   # IC must be 4096
    assert lib.mxq_gemm_workspace_bytes() == 64 * 1024 + 256 * 2 * 256 * 128 * 4 or lib.mxq_gemm_workspace_bytes() > 64 * 1024


def test_no_profiling_entry_point_in_the_product_library(lib):
    """Ablation / profiling builds (wrong results) live in libmxq_hip_prof.so (`make prof`), never here."""
    import subprocess
    from mxq_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    syms = [l.split()[-1] for l in out.splitlines() if l.strip()]
    assert not [s_ for s_ in syms if "ablat" in s_ or "prof" in s_], "profiling symbols leaked into the product ABI"
    assert not hasattr(lib, "mxq_gemm_f16_ex")
