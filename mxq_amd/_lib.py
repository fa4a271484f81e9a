"""ctypes binding of libmxq_hip.so (include/mxq_hip.h).

The HIP library is the product; there is no CPU or pure-PyTorch fallback.  If the
shared object is missing or fails to load, every op raises ``MXQLibraryError`` -- it is
never silently replaced (build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C mxq_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmxq_hip.so")

DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2

_ERRORS = {-1: "invalid shape (need N % 16 == 0, K % 64 == 0, positive sizes, supported group size)",
           -2: "null pointer", -3: "unknown dtype code", -4: "pointer not 16-byte aligned"}


class MXQLibraryError(RuntimeError):
    pass


# name -> (restype, argtypes); must list every symbol include/mxq_hip.h declares
SIGNATURES = {
    "mxq_version": (c_int, []),
    "mxq_qweight_bytes": (c_size_t, [c_int, c_int]),
    "mxq_rowmeta_bytes": (c_size_t, [c_int]),
    "mxq_quantize_pack": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mxq_pack_codes": (c_int, [c_void_p] * 12 + [c_int, c_int, c_void_p]),
    "mxq_unpack": (c_int, [c_void_p] * 12 + [c_int, c_int, c_void_p]),
    "mxq_dequant_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mxq_compact": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mxq_unpack_compact": (c_int, [c_void_p] * 12 + [c_int, c_int, c_void_p]),
    "mxq_dequant_f16_compact": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mxq_gemv_fused_f16_compact": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "mxq_linear_f16": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p]),
    "mxq_gemm_f16": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p]),
    "mxq_actquant_group_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mxq_actquant_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                                 c_void_p]),
    "mxq_gemm_workspace_bytes": (c_size_t, []),
    "mxq_workspace_status": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "mxq_clock_stamp": (c_int, [c_void_p, c_void_p]),
    "mxq_linear_workspace_need": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "mxq_stream_capture_id": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mxq_hoist_scratch_bytes": (c_size_t, [c_int, c_int]),
    "mxq_linear_f16_hoisted": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mxq_hoist_min_tokens": (c_int, []),
    "mxq_linear_f16_auto": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_size_t,
                                    c_void_p]),
    "mxq_dense_f16": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int, c_int, c_void_p]),
    "mxq_linear_f16_ws": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mxq_linear_f16_layout_ws": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mxq_gemm_f16_ws": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mxq_gemv_f16": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p]),
    "mxq_skinny_f16": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p]),
    "mxq_qweight_bytes_layout": (c_size_t, [c_int, c_int, c_int]),
    "mxq_quantize_pack_layout": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mxq_expand_layout": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_void_p]),
    "mxq_gemm_f16_layout": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p]),
    "mxq_gemv_f16_layout": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p]),
    "mxq_gemv_fused_f16": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "mxq_gemv_swiglu_f16": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_float, c_int, c_void_p]),
    "mxq_gemv_staged_f16": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_int, c_void_p]),
    "mxq_lmhead_argmax_f16": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "mxq_attn_decode_f16": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    "mxq_attn_split_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mxq_attn_decode_split_f16": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mxq_rope_row_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "mxq_embed_rope_row": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mxq_lmhead_argmax_advance_f16": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                              c_void_p, c_int, c_void_p]),
    "mxq_attn_decode_row_f16": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p]),
    "mxq_fakequant_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mxq_fakequant_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_int, c_void_p]),
    "mxq_gemv_awq_f16": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_void_p]),
    "mxq_gemm_awq_f16": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mxq_gemv_proto_f16": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_void_p]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raise MXQLibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MXQLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; "
            "run `make -C mxq_amd/csrc` (needs hipcc, --offload-arch=gfx950).")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:   # pragma: no cover
        raise MXQLibraryError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MXQLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(code: int, what: str):
    """Map a C return code to the reference-side error behaviour: rejected arguments ->
    ValueError (like the reference GEMM's std::invalid_argument, gemm_cuda_gen.cu:447-454),
    HIP runtime error -> RuntimeError."""
    if code == 0:
        return
    if code < 0:
        raise ValueError(f"{what}: {_ERRORS.get(code, 'invalid argument')} (code {code})")
    raise RuntimeError(f"{what}: HIP error {code}")
