// Device helpers shared by the decode GEMV kernels (gemv.hip, gemv_chain.hip): the code-dot arithmetic on byte-spread
// code words and the activation staging order it needs.  See gemv.hip's header for the derivation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));


__device__ __forceinline__ float dot2(uint32_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w), __builtin_bit_cast(half2v, x), acc, false);
}

// (a & m) | o in one op (the mask in an SGPR, the fp16 ones in a VGPR: gfx9 VOP3 takes no literals, and left to itself
// the compiler emits v_and + v_or)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t m, uint32_t o) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m), "v"(o));
    return r;
}

// sum_k (1 + q_k / 4) x_k over the 16 two-bit codes of byte-spread word d.  Element k sits at bit 8 (k & 3) +
// 2 (k >> 2): field f of bytes (0, 2) = elements (4f, 4f + 2) lands in mantissa bits 9:8 of the two fp16 lanes
// with one shift, bytes (1, 3) = elements (4f + 1, 4f + 3) likewise.  xa / xb: the group's 16 activations in the
// staged order (x0, x2, x1, x3 | x4, x6, x5, x7 | ...).
__device__ __forceinline__ float codedot2x16(uint32_t d, const uint4 xa, const uint4 xb, float acc) {
    constexpr uint32_t M = 0x03000300u, ONE = 0x3C003C00u;
    acc = dot2(and_or(d << 8, M, ONE), xa.x, acc);
    acc = dot2(and_or(d, M, ONE), xa.y, acc);
    acc = dot2(and_or(d << 6, M, ONE), xa.z, acc);
    acc = dot2(and_or(d >> 2, M, ONE), xa.w, acc);
    acc = dot2(and_or(d << 4, M, ONE), xb.x, acc);
    acc = dot2(and_or(d >> 4, M, ONE), xb.y, acc);
    acc = dot2(and_or(d << 2, M, ONE), xb.z, acc);
    acc = dot2(and_or(d >> 6, M, ONE), xb.w, acc);
    return acc;
}

// sum_k (1 + q_k / 16) x_k over the 8 four-bit codes of byte-spread word d (element k at bit 8 (k & 3) + 4 (k >> 2));
// xa: the 8 activations in the staged order
__device__ __forceinline__ float codedot4x8(uint32_t d, const uint4 xa, float acc) {
    constexpr uint32_t M = 0x03C003C0u, ONE = 0x3C003C00u;
    acc = dot2(and_or(d << 6, M, ONE), xa.x, acc);
    acc = dot2(and_or(d >> 2, M, ONE), xa.y, acc);
    acc = dot2(and_or(d << 2, M, ONE), xa.z, acc);
    acc = dot2(and_or(d >> 6, M, ONE), xa.w, acc);
    return acc;
}

// Wave-wide butterflies WITHOUT the LDS crossbar.  `v += __shfl_xor(v, o)` compiles to ds_bpermute_b32: an LDS-queue round trip
// (~100 cycles behind whatever the workgroup's waves have queued there) per step, six DEPENDENT steps per reduction -- in
// the decode launches' serial prologue (the RMSNorm's sum of squares) and tail.  The same exchanges as vector-ALU operations:
// lane ^ 32 and lane ^ 16 by v_permlane32_swap / v_permlane16_swap (swap(v, v) returns (v.lo, v.lo | v.hi, v.hi): the sum of the
// two results is v[lane] + v[lane ^ 32] in every lane), lane ^ 8 = row_ror:8, lane ^ 2 / ^ 1 = quad_perm; and lane ^ 4 = row_ror:4
// ONCE lanes i and i ^ 8 hold the same value (true from the xor-8 step on: the value then depends on lane mod 8 only, and
// (i + 4) mod 8 = (i ^ 4) mod 8).  Same pairs in the same order, IEEE addition / maximum commute: BIT-IDENTICAL to the
// __shfl_xor butterfly that starts at 32 and halves down to 1.
// (spelled as asm: with the builtin, the sum of its two results came out as `v_add_f32 v6, v6, v6` -- the first result twice --
//  in this compiler when both inputs carry the same value; the s_nop covers the VALU-write -> permlane-read wait states the
//  hazard recogniser would have inserted, which it does not do inside an asm block)
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float swap32_sum(float v) {
    float a = v, b = v;
    swap32(a, b);
    return a + b;
}
__device__ __forceinline__ float swap16_sum(float v) {
    float a = v, b = v;
    swap16(a, b);
    return a + b;
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_ROR8 = 0x128, DPP_ROR4 = 0x124, DPP_XOR2 = 0x4E, DPP_XOR1 = 0xB1;
__device__ __forceinline__ float wave_allsum(float v) {      // == for (o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o)
    v = swap32_sum(v);
    v = swap16_sum(v);
    v += dpp_f32<DPP_ROR8>(v);
    v += dpp_f32<DPP_ROR4>(v);
    v += dpp_f32<DPP_XOR2>(v);
    v += dpp_f32<DPP_XOR1>(v);
    return v;
}
__device__ __forceinline__ float wave_allmax(float v) {      // == for (o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o))
    float a = v, b = v;
    swap32(a, b);
    v = fmaxf(a, b);
    a = v; b = v;
    swap16(a, b);
    v = fmaxf(a, b);
    v = fmaxf(v, dpp_f32<DPP_ROR8>(v));
    v = fmaxf(v, dpp_f32<DPP_ROR4>(v));
    v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v));
    return v;
}
// v[lane] + v[lane ^ 1], then + the same of lane ^ 2: the two quad steps alone (== v += shfl_xor(v, 1); v += shfl_xor(v, 2))
__device__ __forceinline__ float quad_allsum(float v) {
    v += dpp_f32<DPP_XOR1>(v);
    v += dpp_f32<DPP_XOR2>(v);
    return v;
}

// 8 fp16 activations (x0 .. x7) -> the staged order (x0, x2, x1, x3, x4, x6, x5, x7) and their fp32 sum
__device__ __forceinline__ uint4 stage8(const uint4 v, float& sum) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 h = __builtin_bit_cast(h8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += (float)h[j];
    uint4 o;
    o.x = __builtin_amdgcn_perm(v.y, v.x, 0x05040100u);
    o.y = __builtin_amdgcn_perm(v.y, v.x, 0x07060302u);
    o.z = __builtin_amdgcn_perm(v.w, v.z, 0x05040100u);
    o.w = __builtin_amdgcn_perm(v.w, v.z, 0x07060302u);
    return o;
}

}   // namespace
