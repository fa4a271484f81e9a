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
