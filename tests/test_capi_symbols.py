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
