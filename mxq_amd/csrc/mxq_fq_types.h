// Storage-type traits shared by the fake-quant kernels (fakequant.hip, actquant.hip): fp32 / bf16 / fp16
// tensors are processed 16 bytes per lane; every reference op is computed in fp32 and rounded to the
// tensor dtype with rnd(), which is what PyTorch does for 16-bit tensors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mxq_fq {

// Streaming forms of the 16-B access (the `nt` bit of the load / store): a fake-quant pass touches every byte once, so
// neither its reads nor its writes should displace anything in L2.  Measured on one Llama-2-7B decoder block in bf16
// (profiles/r03_fakequant_nt.txt): 212 -> 180 us, against 174-176 us for a plain device copy of the same tensors.
typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_nt(const void* p) {
    const u32x4_nt r = __builtin_nontemporal_load((const u32x4_nt*)p);
    return make_uint4(r[0], r[1], r[2], r[3]);
}
__device__ __forceinline__ void st16_nt(void* p, const uint4 q) {
    const u32x4_nt r = {q.x, q.y, q.z, q.w};
    __builtin_nontemporal_store(r, (u32x4_nt*)p);
}

struct F32 {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ float rnd(float x) { return x; }
    __device__ static __forceinline__ void rnd2(float&, float&) {}
    __device__ static __forceinline__ bool near_boundary(float) { return true; }
    __device__ static __forceinline__ uint32_t boundary_key(float) { return 0u; }
    static constexpr uint32_t KEY_LIMIT = 1u;
    static constexpr bool HAS_FAST_DIV = false;   // fp32 results are not re-rounded: always the IEEE divide
    __device__ static __forceinline__ uint4 load_raw(const void* p, int64_t e) { return *(const uint4*)((const float*)p + e); }
    __device__ static __forceinline__ void unpack(const uint4 a, float v[4]) {
        v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y); v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
    }
    __device__ static __forceinline__ void load(const void* p, int64_t e, float v[4]) { unpack(load_raw(p, e), v); }
    __device__ static __forceinline__ uint4 pack(const float v[4]) {
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
    __device__ static __forceinline__ void store(void* p, int64_t e, const float v[4]) {
        *(float4*)((float*)p + e) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ static __forceinline__ uint4 load_raw_nt(const void* p, int64_t e) { return ld16_nt((const float*)p + e); }
    __device__ static __forceinline__ void load_nt(const void* p, int64_t e, float v[4]) { unpack(load_raw_nt(p, e), v); }
    __device__ static __forceinline__ void store_nt(void* p, int64_t e, const float v[4]) { st16_nt((float*)p + e, pack(v)); }
};

struct BF16 {
    static constexpr int VEC = 8;
    // round-to-nearest-even to 8 significant bits: the plain cast compiles to
    // v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs (MI355X_MICROARCH.md, correctness boundaries)
    __device__ static __forceinline__ float rnd(float x) { return (float)(__bf16)x; }
    // two values per v_cvt_pk_bf16_f32: 1.5 instead of 2 VALU ops per rounding
    __device__ static __forceinline__ void rnd2(float& a, float& b) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 p = {(__bf16)a, (__bf16)b};
        const uint32_t u = __builtin_bit_cast(uint32_t, p);
        a = __uint_as_float(u << 16);
        b = __uint_as_float(u & 0xFFFF0000u);
    }
    // fast-division screen: fl32(x * rcp(e)) rounds to the same bf16 as the correctly rounded
    // fl32(x / e) unless it lies within a few fp32 ulps of a bf16 rounding boundary
    // (no range test needed: e = bf16(alpha + 1e-8) is a positive normal number, bf16 has fp32's
    // exponent range, and for inf / NaN inputs product and quotient agree.)
    __device__ static __forceinline__ bool near_boundary(float a) {
        return ((__float_as_uint(a) + 0x8004u) & 0xFFF8u) == 0u;   // low 16 bits in [0x7FFC, 0x8003]
    }
    // the same test as an unsigned key that can be min-reduced over many values before ONE compare:
    // near_boundary(a)  <=>  boundary_key(a) < KEY_LIMIT   (one v_lshl_add_u32 per value, one v_min3_u32 per two)
    __device__ static __forceinline__ uint32_t boundary_key(float a) { return (__float_as_uint(a) << 16) + 0x80040000u; }
    static constexpr uint32_t KEY_LIMIT = 0x00080000u;
    static constexpr bool HAS_FAST_DIV = true;
    __device__ static __forceinline__ uint4 load_raw(const void* p, int64_t e) { return *(const uint4*)((const uint16_t*)p + e); }
    __device__ static __forceinline__ void unpack(const uint4 a, float v[8]) {
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    __device__ static __forceinline__ void load(const void* p, int64_t e, float v[8]) { unpack(load_raw(p, e), v); }
    __device__ static __forceinline__ uint4 pack(const float v[8]) {      // values already rounded to bf16
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (__float_as_uint(v[2 * i]) >> 16) | (__float_as_uint(v[2 * i + 1]) & 0xFFFF0000u);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static __forceinline__ void store(void* p, int64_t e, const float v[8]) {
        *(uint4*)((uint16_t*)p + e) = pack(v);
    }
    __device__ static __forceinline__ uint4 load_raw_nt(const void* p, int64_t e) { return ld16_nt((const uint16_t*)p + e); }
    __device__ static __forceinline__ void load_nt(const void* p, int64_t e, float v[8]) { unpack(load_raw_nt(p, e), v); }
    __device__ static __forceinline__ void store_nt(void* p, int64_t e, const float v[8]) { st16_nt((uint16_t*)p + e, pack(v)); }
};

struct F16 {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float rnd(float x) { return (float)(_Float16)x; }
    __device__ static __forceinline__ void rnd2(float& a, float& b) {
        a = (float)(_Float16)a;
        b = (float)(_Float16)b;
    }
    __device__ static __forceinline__ bool near_boundary(float a) {   // 11 significant bits; no shortcut in the
        const uint32_t u = __float_as_uint(a);                          // fp16-subnormal range; either sign
        const float m = fabsf(a);
        return ((u & 0x1FFFu) - 0x0FFCu) < 8u || !(m == 0.0f || (m > 6.2e-5f && m < 6.0e4f));
    }
    __device__ static __forceinline__ uint32_t boundary_key(float a) { return near_boundary(a) ? 0u : 0xFFFFFFFFu; }
    static constexpr uint32_t KEY_LIMIT = 1u;
    static constexpr bool HAS_FAST_DIV = true;
    __device__ static __forceinline__ uint4 load_raw(const void* p, int64_t e) { return *(const uint4*)((const uint16_t*)p + e); }
    __device__ static __forceinline__ void unpack(const uint4 r, float v[8]) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 a = __builtin_bit_cast(h8, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    __device__ static __forceinline__ void load(const void* p, int64_t e, float v[8]) { unpack(load_raw(p, e), v); }
    __device__ static __forceinline__ uint4 pack(const float v[8]) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (_Float16)v[i];
        return __builtin_bit_cast(uint4, a);
    }
    __device__ static __forceinline__ void store(void* p, int64_t e, const float v[8]) {
        *(uint4*)((uint16_t*)p + e) = pack(v);
    }
    __device__ static __forceinline__ uint4 load_raw_nt(const void* p, int64_t e) { return ld16_nt((const uint16_t*)p + e); }
    __device__ static __forceinline__ void load_nt(const void* p, int64_t e, float v[8]) { unpack(load_raw_nt(p, e), v); }
    __device__ static __forceinline__ void store_nt(void* p, int64_t e, const float v[8]) { st16_nt((uint16_t*)p + e, pack(v)); }
};

}   // namespace mxq_fq
