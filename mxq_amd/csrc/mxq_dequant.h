// Register-level dequantisation helpers (packed codes -> fp16 pairs), bit-exact with the
// reference's fp32 formula  w' = scale * (q - zero)  rounded once to fp16
// (reference mxq_quant/lib/quantizer.py:19-20, mxqgpt.py:448), where
// scale = qs * (scale_code - qz) (quantizer.py:121).
//
// A 2-bit group has only four possible values, so it is dequantised through a 4-entry
// fp16 LUT computed in fp32 and selected with v_perm_b32 on the
// byte-spread code word (mxq_format.h): ~2.3 VALU ops per weight instead of ~4.5.
// The 4-bit arm is arithmetic: v_cvt_f32_ubyteN, sub, mul, cvt.
//
// The same inline functions compile for the host (tests/host_emu.cpp) with a software
// byte-permute so the bit manipulation is unit-tested on CPU.
#pragma once
#include "mxq_format.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define MXQ_PERM(hi, lo, sel) __builtin_amdgcn_perm((hi), (lo), (sel))
typedef _Float16 mxq_half;
MXQ_HD uint32_t mxq_pack_f16(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = {(_Float16)a, (_Float16)b};   // RNE (default float mode)
    return __builtin_bit_cast(uint32_t, v);
}
#else
// host emulation of v_perm_b32: selector byte 0..3 -> bytes of `lo`, 4..7 -> bytes of `hi`
static inline uint32_t mxq_perm_emu(uint32_t hi, uint32_t lo, uint32_t sel) {
    uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        uint32_t s = (sel >> (8 * i)) & 0xFF;
        uint32_t b = (s < 8) ? (uint32_t)((src >> (8 * s)) & 0xFF) : (s == 0x0C ? 0u : 0xFFu);
        out |= b << (8 * i);
    }
    return out;
}
#define MXQ_PERM(hi, lo, sel) mxq_perm_emu((hi), (lo), (sel))
uint16_t mxq_host_f32_to_f16(float f);   // provided by the host harness (RNE)
MXQ_HD uint32_t mxq_pack_f16(float a, float b) {
    return (uint32_t)mxq_host_f32_to_f16(a) | ((uint32_t)mxq_host_f32_to_f16(b) << 16);
}
#endif

// scale from its 4-bit code: fp32 qs * (code - qz), one rounding per op
MXQ_HD float mxq_scale(float qs, float qz, uint32_t code) { return qs * ((float)code - qz); }

// 16 two-bit codes (byte-spread word d) -> 8 x packed fp16 pairs, out[i] = elements (2i, 2i+1)
MXQ_HD void mxq_deq2x16(uint32_t d, float s, float z, uint32_t out[8]) {
    const uint32_t p01 = mxq_pack_f16(s * (0.0f - z), s * (1.0f - z));
    const uint32_t p23 = mxq_pack_f16(s * (2.0f - z), s * (3.0f - z));
    const uint32_t lut_lo = MXQ_PERM(p23, p01, 0x06040200u);   // low bytes of e0..e3
    const uint32_t lut_hi = MXQ_PERM(p23, p01, 0x07050301u);   // high bytes of e0..e3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t m = (d >> (2 * j)) & 0x03030303u;        // codes of elements 4j..4j+3
        const uint32_t lo = MXQ_PERM(0u, lut_lo, m);
        const uint32_t hi = MXQ_PERM(0u, lut_hi, m);
        out[2 * j] = MXQ_PERM(hi, lo, 0x05010400u);
        out[2 * j + 1] = MXQ_PERM(hi, lo, 0x07030602u);
    }
}

// ... and one half of them: elements 8h .. 8h+7 -> 4 x packed fp16 pairs (the same LUT, the same selections)
MXQ_HD void mxq_deq2x8(uint32_t d, int h, float s, float z, uint32_t out[4]) {
    const uint32_t p01 = mxq_pack_f16(s * (0.0f - z), s * (1.0f - z));
    const uint32_t p23 = mxq_pack_f16(s * (2.0f - z), s * (3.0f - z));
    const uint32_t lut_lo = MXQ_PERM(p23, p01, 0x06040200u);
    const uint32_t lut_hi = MXQ_PERM(p23, p01, 0x07050301u);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t m = (d >> (2 * (2 * h + j))) & 0x03030303u;   // codes of elements 4(2h+j) .. +3
        const uint32_t lo = MXQ_PERM(0u, lut_lo, m);
        const uint32_t hi = MXQ_PERM(0u, lut_hi, m);
        out[2 * j] = MXQ_PERM(hi, lo, 0x05010400u);
        out[2 * j + 1] = MXQ_PERM(hi, lo, 0x07030602u);
    }
}

// byte N of a word as float (one v_cvt_f32_ubyteN; spelled as asm on the device because the optimiser folds the
// nibble masks of the caller into per-element shift+and+convert sequences otherwise: 3 ops instead of 1)
#if defined(__HIP_DEVICE_COMPILE__)
#define MXQ_UBYTE_F32(N)                                                          \
    __device__ __forceinline__ float mxq_ubyte##N(uint32_t v) {                   \
        float f;                                                                  \
        asm("v_cvt_f32_ubyte" #N " %0, %1" : "=v"(f) : "v"(v));                   \
        return f;                                                                 \
    }
MXQ_UBYTE_F32(0) MXQ_UBYTE_F32(1) MXQ_UBYTE_F32(2) MXQ_UBYTE_F32(3)
#undef MXQ_UBYTE_F32
#else
MXQ_HD float mxq_ubyte0(uint32_t v) { return (float)(v & 0xFF); }
MXQ_HD float mxq_ubyte1(uint32_t v) { return (float)((v >> 8) & 0xFF); }
MXQ_HD float mxq_ubyte2(uint32_t v) { return (float)((v >> 16) & 0xFF); }
MXQ_HD float mxq_ubyte3(uint32_t v) { return (float)(v >> 24); }
#endif

// 8 four-bit codes (byte-spread word d) -> 4 x packed fp16 pairs
MXQ_HD void mxq_deq4x8(uint32_t d, float s, float z, uint32_t out[4]) {
    const uint32_t m0 = d & 0x0F0F0F0Fu;          // elements 0..3, one per byte
    const uint32_t m1 = (d >> 4) & 0x0F0F0F0Fu;   // elements 4..7
    out[0] = mxq_pack_f16(s * (mxq_ubyte0(m0) - z), s * (mxq_ubyte1(m0) - z));
    out[1] = mxq_pack_f16(s * (mxq_ubyte2(m0) - z), s * (mxq_ubyte3(m0) - z));
    out[2] = mxq_pack_f16(s * (mxq_ubyte0(m1) - z), s * (mxq_ubyte1(m1) - z));
    out[3] = mxq_pack_f16(s * (mxq_ubyte2(m1) - z), s * (mxq_ubyte3(m1) - z));
}

// Integer unpack (the bit-exact contract of the unpack kernel)
MXQ_HD uint32_t mxq_code2(uint32_t d, int k) { return (d >> mxq_bit2(k)) & 3u; }
MXQ_HD uint32_t mxq_code4(uint32_t d, int k) { return (d >> mxq_bit4(k)) & 15u; }
