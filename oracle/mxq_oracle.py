"""CPU oracle for the MXQ hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the reference's Python arithmetic for the mixed 2/4-bit
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product path
(``mxq_amd``) never does and fails loudly when the HIP library is missing.

How far each restatement is pinned -- three groups, stated wherever the oracle is cited (README):

1. PINNED by fixtures produced by RUNNING the reference's own Python on CPU (``tests/golden/make_golden.py`` ->
   ``tests/golden/g1..g5, g7..g9*.npz``, held bit-exactly by ``tests/test_oracle_golden.py``): ``quantize`` / ``dequantize`` /
   ``find_params`` / ``mxq_quantize`` (the fasterquant layout) / ``linear_ref``, the uniform W2G16 / W4ROW arms (the
   reference's ``Quantizer``), ``mx_fake_quant`` / ``ste_clip_backward`` / ``QuantizeLinear`` / the decoder block, and the
   activation quantisers.  SURVEY.md 8 rows a1-a5, a9-a12 and the native-layout side of a6 / a7.
2. PINNED AT ONE POINT ONLY: ``gemv_mxq_proto_ref`` (the prototype operand format of ``gemv_mxq_forward_cuda``).  The
   reference holds exactly one known answer for it, the constant-operand KAT of ``cuda_kernel/test_correct_gemv.py:19-53``
   (every output == 4096, fixture g6).  Constant operands cannot see a column-mapping or field-order error (SURVEY.md 4 says
   so itself); beyond that point the function restates ``gemv_mxq_cuda.cu:39-208`` as read.
3. PARITY UNPINNED (restatements of the source as read; the reference holds no vector, test or packer, and its native code is
   CUDA + PTX with no ``nvcc`` here): ``gemv_awq_ref`` (operand format of ``gemv_forward_cuda``, ``gemv_cuda.cu:45-242`` --
   only ever called by the unchecked timing script ``test_mxq_gemv.py``) and ``gemm_awq_*`` (operand format of
   ``gemm_forward_cuda``, which the reference declares but never compiles, ``cuda_kernel/setup.py:37-41``; follows
   ``dequantize.cuh:15-78`` and ``gemm_cuda_gen.cu:124-141``).  GPU-vs-oracle agreement on these two says the kernel and the
   restatement read the source the same way, nothing more.

Reference files restated (paths relative to the upstream repo):

* ``mxq_quant/lib/quantizer.py:14-20``  quantize / dequantize
* ``mxq_quant/lib/quantizer.py:61-147`` Quantizer.find_params (asym, per-channel,
  ``qq_scale_bits=4`` second-order scale quantisation over 16 consecutive rows)
* ``mxq_quant/lib/mxqgpt.py:387-448``   MXQGPT.fasterquant (the 48x2b + 16x4b layout)
* ``LLM-QAT/models/utils_quant.py:316-475`` MXAsymQuantizer forward / backward
* ``mxq_quant/cuda_kernel/csrc/quantization/gemv_mxq_cuda.cu:39-208`` and
  ``gemv_cuda.cu:45-242`` operand formats of the two exported GEMV entry points
  (restated with the *intended* column mapping).
* ``mxq_quant/cuda_kernel/csrc/quantization/dequantize.cuh:15-78``, ``gemm_cuda_gen.cu:124-141,
  424-478`` operand format of ``gemm_forward_cuda`` (unpinned, group 3 above).

All float work is done in numpy float32, one IEEE operation per reference op,
so results are bit-identical to torch CPU float32.  ``np.rint`` is
round-half-to-even like ``torch.round``.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
EPS = F32(1e-9)  # quantizer.py:14 ``scale.clamp_min(eps)``

CHUNK = 64        # mxqgpt.py:411 ``for ii in range(0, self.columns, 64)``
N2B = 48          # mxqgpt.py:404-406 ``ratio_2b = 6/8`` -> int(64*0.75)
N4B = 16          # mxqgpt.py:406 ``num_4b = 64 - int(64*ratio_2b)``
GROUP = 16        # prune.py:409 ``blocksize=16``
QQ_GROUP = 16     # quantizer.py:40 ``qq_groupsize=16``


# --------------------------------------------------------------------------- #
# quantizer.py
# --------------------------------------------------------------------------- #
def quantize(x, scale, zero, maxq):
    """quantizer.py:14-16 -- q = clamp(round(x / max(scale, eps) + zero), 0, maxq)."""
    x = np.asarray(x, F32)
    q = np.rint(x / np.maximum(scale, EPS).astype(F32) + zero.astype(F32))
    return np.clip(q, F32(0), F32(maxq)).astype(F32)


def dequantize(q, scale, zero):
    """quantizer.py:19-20 -- scale * (q - zero), fp32."""
    return (scale.astype(F32) * (np.asarray(q, F32) - zero.astype(F32))).astype(F32)


def _minmax_params(x, maxq):
    """quantizer.py:81-99 for perchannel=True, sym=False, round_zero=False.

    x: [rows, cols] fp32.  Returns (scale0 [rows], zero [rows])."""
    x = np.asarray(x, F32)
    xmin = x.min(axis=1).astype(F32)
    xmax = x.max(axis=1).astype(F32)
    same = xmin == xmax                      # :90-92
    xmin = np.where(same, F32(-1), xmin).astype(F32)
    xmax = np.where(same, F32(+1), xmax).astype(F32)
    scale = ((xmax - xmin) / F32(maxq)).astype(F32)      # :94
    zero = ((-xmin) / scale).astype(F32)                 # :99 (float zero-point)
    return scale, zero


def find_params(x, bits, qq_scale_bits=4):
    """Quantizer.configure(bits, perchannel=True, sym=False, qq_scale_bits=4)
    + find_params(x, weight=True)  (quantizer.py:61-147).

    Returns a dict: scale [rows] (after second-order round trip), zero [rows],
    scale_code [rows] (0..15), qs [rows/16], qz [rows/16]."""
    maxq = 2 ** bits - 1
    scale0, zero = _minmax_params(x, maxq)
    rows = scale0.shape[0]
    assert rows % QQ_GROUP == 0, "second-order scale groups need rows % 16 == 0"
    sg = scale0.reshape(-1, QQ_GROUP)                    # :115 16 consecutive rows
    qs, qz = _minmax_params(sg, 2 ** qq_scale_bits - 1)  # :116-118 nested Quantizer
    code = quantize(sg, qs[:, None], qz[:, None], 2 ** qq_scale_bits - 1)   # :120
    scale = dequantize(code, qs[:, None], qz[:, None]).reshape(rows)       # :121
    return dict(scale=scale, zero=zero, scale_code=code.reshape(rows).astype(np.uint8),
                qs=qs, qz=qz, maxq=maxq)


# --------------------------------------------------------------------------- #
# mxqgpt.py  MXQGPT.fasterquant
# --------------------------------------------------------------------------- #
def mxq_quantize(W, dead=None):
    """MXQGPT.fasterquant(blocksize=16) (mxqgpt.py:387-448) on W [N, K].

    ``W`` is the layer weight (any float dtype; it is cast to fp32 like
    ``W.float()``, :394).  ``dead`` is the optional ``diag(H) == 0`` column mask
    (:401-403).  Returns the lossless parameterisation plus the fp32 fake-quant
    weight ``w_deq32`` (the value written back, before the cast of :448)."""
    W = np.array(W, dtype=F32, copy=True)
    N, K = W.shape
    assert K % CHUNK == 0 and N % QQ_GROUP == 0
    if dead is not None:
        W[:, np.asarray(dead, bool)] = 0
    nc = K // CHUNK
    codes2 = np.zeros((N, nc * 3, GROUP), np.uint8)
    sc2 = np.zeros((N, nc * 3), np.uint8)
    zero2 = np.zeros((N, nc * 3), F32)
    qs2 = np.zeros((N // QQ_GROUP, nc * 3), F32)
    qz2 = np.zeros((N // QQ_GROUP, nc * 3), F32)
    Wq = W.copy()
    W4 = np.zeros((N, nc * N4B), F32)                    # :409 (fp32 buffer)
    for c in range(nc):
        for g in range(3):                               # :417 2-bit groups
            lo = c * CHUNK + g * GROUP
            blk = W[:, lo:lo + GROUP]
            p = find_params(blk, 2)                      # :420-422
            q = quantize(blk, p["scale"][:, None], p["zero"][:, None], 3)
            Wq[:, lo:lo + GROUP] = dequantize(q, p["scale"][:, None], p["zero"][:, None])
            j = 3 * c + g
            codes2[:, j, :] = q.astype(np.uint8)
            sc2[:, j] = p["scale_code"]
            zero2[:, j] = p["zero"]
            qs2[:, j] = p["qs"]
            qz2[:, j] = p["qz"]
        W4[:, c * N4B:(c + 1) * N4B] = W[:, c * CHUNK + N2B:(c + 1) * CHUNK]   # :431
    p4 = find_params(W4, 4)                              # :433-435
    q4 = quantize(W4, p4["scale"][:, None], p4["zero"][:, None], 15)
    W4q = dequantize(q4, p4["scale"][:, None], p4["zero"][:, None])           # :436
    for c in range(nc):                                  # :438-443 scatter back
        Wq[:, c * CHUNK + N2B:(c + 1) * CHUNK] = W4q[:, c * N4B:(c + 1) * N4B]
    return dict(
        N=N, K=K,
        codes2=codes2.reshape(N, nc * N2B),   # [N, 3K/4], group-major within chunk
        sc2=sc2, zero2=zero2, qs2=qs2, qz2=qz2,
        codes4=q4.astype(np.uint8),           # [N, K/4]
        sc4=p4["scale_code"], zero4=p4["zero"], qs4=p4["qs"], qz4=p4["qz"],
        w_deq32=Wq.astype(F32),
    )


def mxq_scales(p):
    """Regenerate the dequantised scales from the stored codes (quantizer.py:121):
    s = qs * (code - qz) with (qs, qz) shared by 16 consecutive rows."""
    rep = lambda a: np.repeat(a, QQ_GROUP, axis=0)
    s2 = (rep(p["qs2"]) * (p["sc2"].astype(F32) - rep(p["qz2"]))).astype(F32)
    s4 = (rep(p["qs4"]) * (p["sc4"].astype(F32) - rep(p["qz4"]))).astype(F32)
    return s2, s4


def mxq_compact_params(p):
    """The parameter set the COMPACT metadata mode stores (csrc/mxq_format.h, MXQ_LAYOUT_MIXEDC): identical to the
    exact one except that the 2-bit zero-points are rounded to fp16 (RNE) and read back as fp32.  Not a reference
    function (the reference has no packed format); it restates the build's own format for the parity tests."""
    q = dict(p)
    q["zero2"] = np.asarray(p["zero2"], F32).astype(np.float16).astype(F32)
    return q


def mxq_dequant(p):
    """codes + params -> fp32 fake-quant weight [N, K] (quantizer.py:19-20 applied
    in the layout of mxqgpt.py:404-443)."""
    N, K = int(p["N"]), int(p["K"])
    nc = K // CHUNK
    s2, s4 = mxq_scales(p)
    c2 = p["codes2"].reshape(N, nc * 3, GROUP).astype(F32)
    w2 = (s2[:, :, None] * (c2 - p["zero2"][:, :, None])).astype(F32)
    w4 = (s4[:, None] * (p["codes4"].astype(F32) - p["zero4"][:, None])).astype(F32)
    W = np.empty((N, nc, CHUNK), F32)
    W[:, :, :N2B] = w2.reshape(N, nc, N2B)
    W[:, :, N2B:] = w4.reshape(N, nc, N4B)
    return W.reshape(N, K)


def uniform_quantize(W, layout):
    """Uniform arms of the config-5 sweep, built from the same Quantizer arithmetic
    (quantizer.py:61-147, :14-20): "w2g16" = Quantizer(bits=2, perchannel, qq_scale_bits=4) on every
    16-column group; "w4row" = Quantizer(bits=4, perchannel, qq_scale_bits=4) on whole rows."""
    W = np.asarray(W, F32)
    N, K = W.shape
    if layout == "w4row":
        p = find_params(W, 4)
        q = quantize(W, p["scale"][:, None], p["zero"][:, None], 15)
        wq = dequantize(q, p["scale"][:, None], p["zero"][:, None])
        return dict(codes=q.astype(np.uint8), sc=p["scale_code"][:, None], zero=p["zero"][:, None],
                    qs=p["qs"][:, None], qz=p["qz"][:, None], w_deq32=wq)
    assert layout == "w2g16"
    G = K // GROUP
    codes = np.zeros((N, K), np.uint8)
    sc = np.zeros((N, G), np.uint8); zero = np.zeros((N, G), F32)
    qs = np.zeros((N // QQ_GROUP, G), F32); qz = np.zeros((N // QQ_GROUP, G), F32)
    wq = np.zeros((N, K), F32)
    for g in range(G):
        blk = W[:, g * GROUP:(g + 1) * GROUP]
        p = find_params(blk, 2)
        q = quantize(blk, p["scale"][:, None], p["zero"][:, None], 3)
        codes[:, g * GROUP:(g + 1) * GROUP] = q.astype(np.uint8)
        wq[:, g * GROUP:(g + 1) * GROUP] = dequantize(q, p["scale"][:, None], p["zero"][:, None])
        sc[:, g], zero[:, g], qs[:, g], qz[:, g] = p["scale_code"], p["zero"], p["qs"], p["qz"]
    return dict(codes=codes, sc=sc, zero=zero, qs=qs, qz=qz, w_deq32=wq)


def linear_ref(x16, w16):
    """a5: y = x16 . fp16(w')^T accumulated in fp32."""
    return np.asarray(x16, np.float16).astype(F32) @ np.asarray(w16, np.float16).astype(F32).T


# --------------------------------------------------------------------------- #
# LLM-QAT utils_quant.py  MXAsymQuantizer
# --------------------------------------------------------------------------- #
def _round_to(x32, dtype):
    """Round fp32 values to ``dtype`` precision (kept in fp32 storage)."""
    x32 = np.asarray(x32, F32)
    if dtype == "fp32":
        return x32
    if dtype == "fp16":
        return x32.astype(np.float16).astype(F32)
    if dtype == "bf16":
        u = x32.view(np.uint32).astype(np.uint64)
        nan = np.isnan(x32)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        out = (r & 0xFFFFFFFF).astype(np.uint32).view(F32)
        return np.where(nan, x32, out).astype(F32)
    raise ValueError(dtype)


def bf16_bits(x32):
    """fp32 array holding bf16-representable values -> uint16 bit patterns."""
    return (np.asarray(x32, F32).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(u16):
    return (np.asarray(u16, np.uint16).astype(np.uint32) << 16).view(F32)


def fakequant_fwd(w, num_bits=2, dtype="fp32"):
    """MXAsymQuantizer.forward, live branch (2-D, layerwise=False)
    (utils_quant.py:334-385,456-460).  ``w`` holds values representable in
    ``dtype`` (stored as fp32); every reference op is done in fp32 and rounded to
    ``dtype`` -- what torch does for bf16/fp16 tensors.  Returns fp32 storage."""
    with np.errstate(all="ignore"):
        r = lambda a: _round_to(a, dtype)
        w = np.asarray(w, F32)
        N, K = w.shape
        assert K % CHUNK == 0
        nc = K // CHUNK
        wc = w.reshape(N, nc, CHUNK)
        alpha = np.empty_like(wc)
        beta = np.empty_like(wc)
        s = np.empty_like(wc)
        g2 = wc[:, :, :N2B].reshape(N, nc, 3, GROUP)
        mx, mn = g2.max(axis=3), g2.min(axis=3)
        a2 = r(mx - mn)                                  # :354-362 in the tensor dtype
        alpha[:, :, :N2B] = np.repeat(a2, GROUP, axis=2)
        beta[:, :, :N2B] = np.repeat(mn, GROUP, axis=2)
        s[:, :, :N2B] = F32(2 ** num_bits - 1)           # :366
        w4 = wc[:, :, N2B:].reshape(N, nc * N4B)         # :347,368 fp32 buffer
        a4 = (w4.max(axis=1) - w4.min(axis=1)).astype(F32)   # :369-376 fp32 subtraction
        b4 = w4.min(axis=1)
        alpha[:, :, N2B:] = r(a4)[:, None, None]         # :383 cast on assignment
        beta[:, :, N2B:] = b4[:, None, None]
        s[:, :, N2B:] = F32(15)                          # :385
        e = r(alpha + F32(1e-8))                         # alpha + 1e-8
        xn = r(r(wc - beta) / e)                         # :456
        q = np.rint(r(xn * s))                           # :458 round
        out = r(r(r(q / s) * e) + beta)                  # :458-460
        return out.reshape(N, K).astype(F32)


def fakequant_bwd(grad_out, w, lo=-2.0, hi=2.0):
    """MXAsymQuantizer.backward (utils_quant.py:464-475): STE with clip mask."""
    g = np.array(grad_out, copy=True)
    w = np.asarray(w, F32)
    g[w >= F32(hi)] = 0
    g[w <= F32(lo)] = 0
    return g


# --------------------------------------------------------------------------- #
# Operand formats of the reference's two exported GEMV entry points
# --------------------------------------------------------------------------- #
def gemv_awq_ref(x16, kernel, scales16, zeros, group_size):
    """gemv_forward_cuda semantics (gemv_cuda.cu:45-242, :346-399).  PARITY UNPINNED: the reference holds no value
    check for this entry (only the timing script test_mxq_gemv.py calls it); restated as read.

    kernel int32 [OC, IC/8] (nibble j of word = element j), scales fp16
    [OC, sf_w], zeros int32 [OC, zeros_w] with zeros_w = ceil(ceil(IC/G/8)/4)*4
    and sf_w = zeros_w*8 (gemv_cuda.cu:54-59).  w = s[oc,g] * (q - z[oc,g])."""
    x = np.asarray(x16, np.float16).astype(F32)
    kern = np.asarray(kernel).view(np.uint32)
    OC, W8 = kern.shape
    IC = W8 * 8
    q = np.stack([(kern >> (4 * j)) & 0xF for j in range(8)], axis=-1).reshape(OC, IC).astype(F32)
    ng = IC // group_size
    zw = np.asarray(zeros).view(np.uint32)
    z = np.stack([(zw >> (4 * j)) & 0xF for j in range(8)], axis=-1).reshape(OC, -1)[:, :ng].astype(F32)
    s = np.asarray(scales16, np.float16).astype(F32)[:, :ng]
    w = (np.repeat(s, group_size, 1) * (q - np.repeat(z, group_size, 1))).astype(F32)
    return x @ w.T


# nibble that holds output channel e of an 8-channel word in the reference GEMM's operands: dequantize_s4_to_fp16x2
# returns the halves (n0, n4 | n1, n5 | n2, n6 | n3, n7) as elements 0..7 (dequantize.cuh:35-51)
AWQ_GEMM_NIBBLE = (0, 4, 1, 5, 2, 6, 3, 7)


def gemm_awq_pack(q, axis_words=None):
    """Pack integer codes q [rows, OC] (0..15) into int32 [rows, OC/8] in the GEMM's interleaved nibble order (the
    inverse of ``gemm_awq_unpack``; the reference ships no packer for this format -- this one only has to agree with
    dequantize.cuh:35-51, which ``gemm_awq_unpack`` restates)."""
    q = np.asarray(q).astype(np.uint32)
    rows, OC = q.shape
    q8 = q.reshape(rows, OC // 8, 8)
    w = np.zeros((rows, OC // 8), np.uint32)
    for e in range(8):
        w |= (q8[:, :, e] & 0xF) << np.uint32(4 * AWQ_GEMM_NIBBLE[e])
    return w.view(np.int32)


def gemm_awq_unpack(words):
    """int32 [rows, OC/8] -> codes [rows, OC]: channel 8 c + e of word c sits in nibble AWQ_GEMM_NIBBLE[e]
    (dequantize.cuh:35-51)."""
    w = np.asarray(words).view(np.uint32)
    out = np.stack([(w >> np.uint32(4 * AWQ_GEMM_NIBBLE[e])) & 0xF for e in range(8)], axis=-1)
    return out.reshape(w.shape[0], -1)


def gemm_awq_weight(kernel, scales16, zeros, group_size):
    """The fp16 weight [IC, OC] the reference GEMM multiplies by (gemm_cuda_gen.cu:124-141): per element
    sub.f16x2 (q - z: an exact integer) then fma.rn.f16x2 with a zero addend = ONE rounding of (q - z) * s to fp16.
    PARITY UNPINNED: the reference never compiles this kernel (setup.py:37-41) and holds no test or fixture for it;
    this restates its arithmetic as read."""
    q = gemm_awq_unpack(kernel).astype(F32)                                  # [IC, OC]
    z = np.repeat(gemm_awq_unpack(zeros).astype(F32), group_size, axis=0)    # [IC, OC]
    s = np.repeat(np.asarray(scales16, np.float16).astype(F32), group_size, axis=0)
    return ((q - z) * s).astype(np.float16)                                  # (q - z) * s is exact in fp32: one rounding


def gemm_awq_ref(x16, kernel, scales16, zeros, group_size):
    """gemm_forward_cuda semantics with ONE fp32 accumulation (the reference rounds each of its split_k partial sums to
    fp16 before adding them, gemm_cuda_gen.cu:210-216, :477: a property of its schedule, not of the operator)."""
    return np.asarray(x16, np.float16).astype(F32) @ gemm_awq_weight(kernel, scales16, zeros, group_size).astype(F32)


def gemv_mxq_proto_ref(x16, weight, weight_last, zeros_and_scales, scales_2nd, zeros_2nd,
                       scales_4b, zeros_4b):
    """PINNED AT ONE POINT (the constant-operand KAT, fixture g6) -- everything else as read:
    gemv_mxq_forward_cuda semantics (gemv_mxq_cuda.cu:39-208) with the intended
    column mapping: lane t, iteration it owns columns 2048*it + 64*t .. +63
    (the reference kernel's iteration-1 activation offset
    bug, :119, is deliberately not reproduced)."""
    x = np.asarray(x16, np.float16).astype(F32)
    B, IC = x.shape
    assert IC % 2048 == 0 and IC <= 4096
    w = np.asarray(weight).view(np.uint32)
    wl = np.asarray(weight_last).view(np.uint32)
    zs = np.asarray(zeros_and_scales).view(np.uint32)
    z2 = np.asarray(zeros_2nd).view(np.uint32)
    s2 = np.asarray(scales_2nd, np.float16).astype(F32).reshape(-1)
    s4 = np.asarray(scales_4b, np.float16).astype(F32)
    z4w = np.asarray(zeros_4b).view(np.uint32)
    OC = w.shape[0]
    weight_w = IC // 64 * 4
    last_w = IC // 64
    W = np.zeros((OC, IC), F32)
    oc = np.arange(OC)
    z4 = ((z4w[oc // 8] >> ((oc % 8) * 4)) & 0xF).astype(F32)
    for it in range(IC // 2048):
        for t in range(32):
            col0 = 2048 * it + 64 * t
            zsw = zs[:, t]
            z1 = (zsw >> (16 * it)) & 0xFF
            s1 = (zsw >> (16 * it + 8)) & 0xFF
            z2p = (z2[oc // 4, t] >> (8 * it)) & 0xFF
            for g in range(3):
                cz1 = ((z1 >> (2 * g)) & 3).astype(F32)
                cs1 = ((s1 >> (2 * g)) & 3).astype(F32)
                cz2 = ((z2p >> (2 * g)) & 3).astype(F32)
                sf2 = s2[(oc // 4) * 192 + it * 96 + t * 3 + g]
                sf = (sf2 * (cs1 - cz2)).astype(F32)
                word = w[:, it * (weight_w // 2) + 4 * t + g]
                for j in range(16):
                    qv = ((word >> (2 * j)) & 3).astype(F32)
                    W[:, col0 + 16 * g + j] = sf * (qv - cz1)
            word = w[:, it * (weight_w // 2) + 4 * t + 3]
            wordl = wl[:, it * (last_w // 2) + t]
            for j in range(8):
                W[:, col0 + 48 + j] = s4 * (((word >> (4 * j)) & 0xF).astype(F32) - z4)
                W[:, col0 + 56 + j] = s4 * (((wordl >> (4 * j)) & 0xF).astype(F32) - z4)
    return x @ W.T
