// Packed-format kernels: pack-from-codes, unpack, dequant-to-fp16, and the fused
// on-device quantise-and-pack (reference MXQGPT.fasterquant, mxq_quant/lib/mxqgpt.py:387-448,
// Quantizer.find_params lib/quantizer.py:61-147, restated per SURVEY.md Appendix A1).
#include <hip/hip_runtime.h>
#include <math.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_pack.h"
#include "mxq_kernels.h"

// ------------------------------------------------------------------------------------ //
// pack / unpack: one thread per (row, chunk); utility kernels, not on the timed path.
// ------------------------------------------------------------------------------------ //
__global__ void mxq_pack_codes_kernel(const uint8_t* __restrict__ codes2, const uint8_t* __restrict__ sc2,
                                      const float* __restrict__ zero2, const float* __restrict__ qs2,
                                      const float* __restrict__ qz2, const uint8_t* __restrict__ codes4,
                                      const uint8_t* __restrict__ sc4, const float* __restrict__ zero4,
                                      const float* __restrict__ qs4, const float* __restrict__ qz4,
                                      uint32_t* __restrict__ qweight, float4* __restrict__ rowmeta, int N, int K) {
    const int NC = K / 64;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * NC) return;
    const int n = (int)(idx / NC), c = (int)(idx % NC);
    uint32_t* tile = qweight + mxq_blk_index(n, c, K) * MXQ_BLK_DW;
    const int r = n & 15;
    mxq_pack_row_chunk(tile, r, codes2 + (int64_t)n * NC * 48 + c * 48, sc2 + (int64_t)n * NC * 3 + c * 3,
                       zero2 + (int64_t)n * NC * 3 + c * 3, codes4 + (int64_t)n * NC * 16 + c * 16);
    if (r == 0) {
        for (int g = 0; g < 3; ++g) {
            tile[mxq_qq(g)] = __float_as_uint(qs2[(int64_t)(n / 16) * NC * 3 + c * 3 + g]);
            tile[mxq_qq(g) + 1] = __float_as_uint(qz2[(int64_t)(n / 16) * NC * 3 + c * 3 + g]);
        }
        tile[mxq_qq(3)] = 0u;   // unused slot: keep the packed bytes deterministic
        tile[mxq_qq(3) + 1] = 0u;
    }
    if (c == 0) rowmeta[n] = make_float4(zero4[n], (float)sc4[n], qs4[n / 16], qz4[n / 16]);
}

template <bool COMPACT>
__global__ void mxq_unpack_kernel(const uint32_t* __restrict__ qweight, const float4* __restrict__ rowmeta,
                                  uint8_t* __restrict__ codes2, uint8_t* __restrict__ sc2, float* __restrict__ zero2,
                                  float* __restrict__ qs2, float* __restrict__ qz2, uint8_t* __restrict__ codes4,
                                  uint8_t* __restrict__ sc4, float* __restrict__ zero4, float* __restrict__ qs4,
                                  float* __restrict__ qz4, int N, int K) {
    const int NC = K / 64;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * NC) return;
    const int n = (int)(idx / NC), c = (int)(idx % NC);
    typedef MxqMixed<COMPACT> F;
    const uint32_t* tile = qweight + mxq_blk_index(n, c, K) * F::BLK_DW;
    const int r = n & 15;
    if constexpr (!COMPACT) {
        mxq_unpack_row_chunk(tile, r, codes2 + (int64_t)n * NC * 48 + c * 48, sc2 + (int64_t)n * NC * 3 + c * 3,
                             zero2 + (int64_t)n * NC * 3 + c * 3, codes4 + (int64_t)n * NC * 16 + c * 16);
    } else {   // same code words; zero-points widened from fp16, scale codes from the compact SC field
        uint8_t* c2 = codes2 + (int64_t)n * NC * 48 + c * 48;
        uint8_t* c4 = codes4 + (int64_t)n * NC * 16 + c * 16;
        const uint32_t scw = F::scw(tile, r);
        for (int g = 0; g < 3; ++g) {
            const uint32_t w = tile[mxq_c2(g, r)];
            for (int k = 0; k < 16; ++k) c2[g * 16 + k] = (uint8_t)mxq_code2(w, k);
            zero2[(int64_t)n * NC * 3 + c * 3 + g] = F::z2(tile, g, r);
            sc2[(int64_t)n * NC * 3 + c * 3 + g] = (uint8_t)((scw >> (4 * g)) & 15u);
        }
        for (int h = 0; h < 2; ++h) {
            const uint32_t w = tile[mxq_c4(h, r)];
            for (int k = 0; k < 8; ++k) c4[h * 8 + k] = (uint8_t)mxq_code4(w, k);
        }
    }
    if (r == 0) {
        for (int g = 0; g < 3; ++g) {
            qs2[(int64_t)(n / 16) * NC * 3 + c * 3 + g] = __uint_as_float(tile[F::qq(g)]);
            qz2[(int64_t)(n / 16) * NC * 3 + c * 3 + g] = __uint_as_float(tile[F::qq(g) + 1]);
        }
    }
    if (c == 0) {
        const float4 m = rowmeta[n];
        zero4[n] = m.x;
        sc4[n] = (uint8_t)m.y;
        if (r == 0) {
            qs4[n / 16] = m.z;
            qz4[n / 16] = m.w;
        }
    }
}

// ------------------------------------------------------------------------------------ //
// dequant to a dense fp16 [N, K] matrix (bit-exact fake-quant weight, mxqgpt.py:448): the first half of the
// hoisted mode.  Write-bound (2 B out per 0.56 B in), so the layout follows the STORES: a lane owns one 16-byte
// slot (8 weights = half a 16-column quarter) of a row's 128-byte chunk line, 8 lanes cover the line and a wave
// instruction writes 8 rows x 128 B in full lines.  (Round 2's thread -> (row, quarter) mapping wrote 32-byte
// pieces of 16 different rows per instruction: 13.0 -> 9.6 us at 4096^2, profiles/r03_dense256.txt.)
// Block = 4 waves = the 4 chunks of a chunk quad x one 16-row block; a wave does its chunk's rows 0-7, then 8-15.
// ------------------------------------------------------------------------------------ //
template <bool COMPACT>
__global__ __launch_bounds__(256) void mxq_dequant_f16_kernel(const uint32_t* __restrict__ qweight,
                                                              const float4* __restrict__ rowmeta,
                                                              uint16_t* __restrict__ out, int N, int K) {
    const int NC = K / 64, NC4 = (NC + 3) / 4;
    const int rb = blockIdx.x / NC4, c4 = blockIdx.x % NC4;
    const int lane = threadIdx.x & 63;
    const int c = c4 * 4 + (threadIdx.x >> 6);          // wave-uniform
    if (c >= NC) return;
    const int slot = lane & 7, qt = slot >> 1, h = slot & 1;
    typedef MxqMixed<COMPACT> F;
    const uint32_t* tile = qweight + ((int64_t)rb * NC + c) * F::BLK_DW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = i * 8 + (lane >> 3), n = rb * 16 + r;
        uint32_t o[4];
        if (qt < 3) {
            const uint32_t d = tile[mxq_c2(qt, r)];
            const float z = F::z2(tile, qt, r);
            const uint32_t scw = F::scw(tile, r);
            const float qs = __uint_as_float(tile[F::qq(qt)]), qz = __uint_as_float(tile[F::qq(qt) + 1]);
            mxq_deq2x8(d, h, mxq_scale(qs, qz, (scw >> (4 * qt)) & 15u), z, o);
        } else {
            const float4 m = rowmeta[n];
            mxq_deq4x8(tile[mxq_c4(h, r)], mxq_scale(m.z, m.w, (uint32_t)m.y), m.x, o);
        }
        *(uint4*)(out + (int64_t)n * K + c * 64 + slot * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------ //
// Fused quantise-and-pack.  One 256-thread block per 16-row block (the second-order
// scale group, quantizer.py:115).  lane -> (r = lane & 15, qt = lane >> 4): 16 rows x the
// four 16-wide quarters of a chunk; wave w takes chunks c = w (mod 4).
//   pass 1: 2-bit groups (qt < 3) are finished per chunk: per-row min/max -> s0, z; the
//           16 lanes of a quarter reduce min/max of s0 (second-order) -> code, scale;
//           lanes with qt == 3 only track the row min/max of the gathered 4-bit slice
//           (mxqgpt.py:431-435).
//   pass 2: 4-bit arm, lane -> (r, chunk cc of a tile).
// ------------------------------------------------------------------------------------ //
__device__ __forceinline__ void load16(const void* W, int dtype, int64_t off, float v[16]) {
    if (dtype == MXQ_DTYPE_F16) {
        const uint4* p = (const uint4*)((const uint16_t*)W + off);
        uint4 a = p[0], b = p[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 h = __builtin_bit_cast(h2, w[i]);
            v[2 * i] = (float)h[0];
            v[2 * i + 1] = (float)h[1];
        }
    } else if (dtype == MXQ_DTYPE_BF16) {
        const uint4* p = (const uint4*)((const uint16_t*)W + off);
        uint4 a = p[0], b = p[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    } else {
        const float4* p = (const float4*)((const float*)W + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 a = p[i];
            v[4 * i] = a.x; v[4 * i + 1] = a.y; v[4 * i + 2] = a.z; v[4 * i + 3] = a.w;
        }
    }
}

__device__ __forceinline__ float grp16_min(float x) {
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) x = fminf(x, __shfl_xor(x, o, 64));
    return x;
}
__device__ __forceinline__ float grp16_max(float x) {
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

// Quantizer.find_params second half (quantizer.py:90-99,114-121) for one row of a group:
// (lo, hi) -> zero, scale code, dequantised scale; (qs, qz) shared by the 16 lanes.
__device__ __forceinline__ void find_params16(float lo, float hi, float maxq, float& z, float& scode, float& s,
                                              float& qs, float& qz) {
    if (lo == hi) { lo = -1.0f; hi = 1.0f; }
    const float s0 = (hi - lo) / maxq;
    z = (-lo) / s0;
    float slo = grp16_min(s0), shi = grp16_max(s0);
    if (slo == shi) { slo = -1.0f; shi = 1.0f; }
    qs = (shi - slo) / 15.0f;
    qz = (-slo) / qs;
    scode = fminf(fmaxf(rintf(s0 / fmaxf(qs, 1e-9f) + qz), 0.0f), 15.0f);
    s = qs * (scode - qz);
}

// q = clamp(rne(x / sd + z), 0, maxq) (quantizer.py:14-16) for 16 values.  An IEEE division is ~10 VALU ops; here
// a' = fl(x * v_rcp_f32(sd)) stands in for a = fl(x / sd) behind a screen: |a' - a| <= |a| * 2^-22 (v_rcp_f32 is good to
// 1 ulp, plus the two roundings), so t' = fl(a' + z) is within |a'| * 2^-21 + 2^-19 of t = fl(a + z) for |t| < 17
// (two ulps of slack on the bound, one ulp of t at that magnitude), and both are first clamped to [-1, maxq + 1], where
// nothing changes the final code.  If every value of the WAVE keeps its clamped t' farther than that from the nearest
// rounding boundary (k + 0.5), the codes are those of the exact formula; otherwise -- ties included -- the wave takes
// the division.  Typical weights: 1-2 % of the wave iterations.  Non-finite input, stated exactly:
//   * an infinite weight makes a' infinite, its key -inf: the wave takes the division;
//   * a NaN weight has a NaN key, which v_min_f32 DROPS (it returns its non-NaN operand), so it stays on the fast
//     path; its code is 0 there (v_med3_f32 with a NaN operand returns the minimum of the others, -1, clamped to 0)
//     and 0 on the division path (fmaxf(NaN, 0) = 0) -- the same code either way, no other element is affected
//     (tests: test_quantize_pack_nonfinite_and_huge_scale);
//   * v_rcp_f32 flushes a denormal result to zero, so for sd > 2^126 a' would be 0 with a positive key: such a
//     group is sent to the division by rcp_usable() below.
template <int MAXQ>
__device__ __forceinline__ float quant_key(float x, float r, float z, float& q) {
    const float a = x * r;
    const float t = __builtin_amdgcn_fmed3f(a + z, -1.0f, (float)(MAXQ + 1));
    const float n = rintf(t);
    q = __builtin_amdgcn_fmed3f(n, 0.0f, (float)MAXQ);
    return (0.5f - fabsf(t - n)) - fmaf(fabsf(a), 0x1p-21f, 0x1p-19f);
}
template <int MAXQ>
__device__ __forceinline__ float quant_exact(float x, float sd, float z) {
    return fminf(fmaxf(rintf(x / sd + z), 0.0f), (float)MAXQ);
}
// false when 1 / sd is not a normal number (sd > 2^126, or sd NaN): the reciprocal was flushed and proves nothing
__device__ __forceinline__ bool rcp_usable(float r) { return r >= 0x1p-126f; }
__device__ __forceinline__ uint32_t quant2x16(const float (&v)[16], float sd, float z) {
    const float r = __builtin_amdgcn_rcpf(sd);
    float key = INFINITY;
    uint32_t word = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float q;
        key = fminf(key, quant_key<3>(v[j], r, z, q));
        word |= (uint32_t)q << mxq_bit2(j);
    }
    if (__any(!(key > 0.0f) || !rcp_usable(r))) {
        word = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) word |= (uint32_t)quant_exact<3>(v[j], sd, z) << mxq_bit2(j);
    }
    return word;
}
__device__ __forceinline__ void quant4x16(const float (&v)[16], float sd, float z, uint32_t& w0, uint32_t& w1) {
    const float r = __builtin_amdgcn_rcpf(sd);
    float key = INFINITY;
    w0 = 0;
    w1 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float q0, q1;
        key = fminf(key, quant_key<15>(v[j], r, z, q0));
        key = fminf(key, quant_key<15>(v[j + 8], r, z, q1));
        w0 |= (uint32_t)q0 << mxq_bit4(j);
        w1 |= (uint32_t)q1 << mxq_bit4(j);
    }
    if (__any(!(key > 0.0f) || !rcp_usable(r))) {
        w0 = 0;
        w1 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            w0 |= (uint32_t)quant_exact<15>(v[j], sd, z) << mxq_bit4(j);
            w1 |= (uint32_t)quant_exact<15>(v[j + 8], sd, z) << mxq_bit4(j);
        }
    }
}

// 16 waves per 16-row block (a wave takes the chunks c = w mod 16) and the next chunk's loads issued before the
// current one is worked on: the kernel is a stream of 32-byte loads per lane with ~300 VALU ops between them, and
// with the 4 waves per block of round 1 (one wave per SIMD, one load pair in flight each) it ran at the bytes in
// flight, 1.4-2.1 TB/s (profiles/r02_kernels_bench.txt).
constexpr int QP_WAVES = 16;
struct Raw16 { uint4 a, b, c, d; };   // 16 elements as loaded: 32 bytes (16-bit dtypes) or 64 bytes (fp32)
__device__ __forceinline__ void load16_raw(const void* W, int dtype, int64_t off, Raw16& t) {
    if (dtype == MXQ_DTYPE_F32) {
        const uint4* p = (const uint4*)((const float*)W + off);
        t.a = p[0]; t.b = p[1]; t.c = p[2]; t.d = p[3];
    } else {
        const uint4* p = (const uint4*)((const uint16_t*)W + off);
        t.a = p[0]; t.b = p[1];
    }
}
__device__ __forceinline__ void cvt16(const Raw16& t, int dtype, float v[16]) {
    if (dtype == MXQ_DTYPE_F16) {
        const uint32_t w[8] = {t.a.x, t.a.y, t.a.z, t.a.w, t.b.x, t.b.y, t.b.z, t.b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 h = __builtin_bit_cast(h2, w[i]);
            v[2 * i] = (float)h[0];
            v[2 * i + 1] = (float)h[1];
        }
    } else if (dtype == MXQ_DTYPE_BF16) {
        const uint32_t w[8] = {t.a.x, t.a.y, t.a.z, t.a.w, t.b.x, t.b.y, t.b.z, t.b.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    } else {
        const uint32_t w[16] = {t.a.x, t.a.y, t.a.z, t.a.w, t.b.x, t.b.y, t.b.z, t.b.w,
                                t.c.x, t.c.y, t.c.z, t.c.w, t.d.x, t.d.y, t.d.z, t.d.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __uint_as_float(w[i]);
    }
}

__global__ __launch_bounds__(QP_WAVES * 64) void mxq_quantize_pack_kernel(const void* __restrict__ W, int dtype,
                                                                         const uint8_t* __restrict__ dead,
                                                                         uint32_t* __restrict__ qweight,
                                                                         float4* __restrict__ rowmeta, int N, int K) {
    const int NC = K / 64, NC4 = (NC + 3) / 4;
    const int rb = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, qt = lane >> 4;
    const int n = rb * 16 + r;
    __shared__ float red[2][QP_WAVES][16];
    float mn4 = INFINITY, mx4 = -INFINITY;
    float v[16];
    Raw16 nxt = {};

    if (wave < NC) load16_raw(W, dtype, (int64_t)n * K + wave * 64 + qt * 16, nxt);
    for (int c = wave; c < NC; c += QP_WAVES) {
        const int k0 = c * 64 + qt * 16;
        const Raw16 cur = nxt;
        if (c + QP_WAVES < NC) load16_raw(W, dtype, (int64_t)n * K + k0 + QP_WAVES * 64, nxt);
        cvt16(cur, dtype, v);
        if (dead) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (dead[k0 + j]) v[j] = 0.0f;
        }
        float lo = v[0], hi = v[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) { lo = fminf(lo, v[j]); hi = fmaxf(hi, v[j]); }
        if (qt == 3) { mn4 = fminf(mn4, lo); mx4 = fmaxf(mx4, hi); }
        float z, scode, s, qs, qz;
        find_params16(lo, hi, 3.0f, z, scode, s, qs, qz);
        const float sd = fmaxf(s, 1e-9f);
        const uint32_t word = quant2x16(v, sd, z);
        const uint32_t sci = (uint32_t)scode;
        const uint32_t sc1 = __shfl(sci, r + 16, 64), sc2 = __shfl(sci, r + 32, 64);
        uint32_t* tile = qweight + ((int64_t)rb * NC + c) * MXQ_BLK_DW;
        if (qt < 3) {
            tile[mxq_c2(qt, r)] = word;
            tile[mxq_z2(qt, r)] = __float_as_uint(z);
            if (r == 0) {
                tile[mxq_qq(qt)] = __float_as_uint(qs);
                tile[mxq_qq(qt) + 1] = __float_as_uint(qz);
            }
        } else if (r == 0) {   // unused fourth QQ slot: keep the packed bytes deterministic
            tile[mxq_qq(3)] = 0u;
            tile[mxq_qq(3) + 1] = 0u;
        }
        if (qt == 0) ((uint16_t*)tile)[mxq_sc_u16(r)] = (uint16_t)(sci | (sc1 << 4) | (sc2 << 8));
    }
    // the 4-bit arm's first slices are on their way while the row ranges are reduced (c = 4 * wave + qt: qt plays the
    // role of the chunk slot in pass 2)
    const bool has4 = 4 * wave + qt < NC;
    if (has4) load16_raw(W, dtype, (int64_t)n * K + (4 * wave + qt) * 64 + 48, nxt);
    if (qt == 3) { red[0][wave][r] = mn4; red[1][wave][r] = mx4; }
    __syncthreads();
    float lo4 = red[0][0][r], hi4 = red[1][0][r];
#pragma unroll
    for (int w = 1; w < QP_WAVES; ++w) { lo4 = fminf(lo4, red[0][w][r]); hi4 = fmaxf(hi4, red[1][w][r]); }
    float z4, sc4, s4, qs4, qz4;
    find_params16(lo4, hi4, 15.0f, z4, sc4, s4, qs4, qz4);
    if (wave == 0 && qt == 0) rowmeta[n] = make_float4(z4, sc4, qs4, qz4);
    const float sd4 = fmaxf(s4, 1e-9f);

    for (int c4 = wave; c4 < NC4; c4 += QP_WAVES) {
        const int c = c4 * 4 + qt;
        const Raw16 cur = nxt;
        const int cn = c + 4 * QP_WAVES;
        if (cn < NC) load16_raw(W, dtype, (int64_t)n * K + cn * 64 + 48, nxt);
        if (c >= NC) continue;
        const int k0 = c * 64 + 48;
        cvt16(cur, dtype, v);
        if (dead) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (dead[k0 + j]) v[j] = 0.0f;
        }
        uint32_t w0, w1;
        quant4x16(v, sd4, z4, w0, w1);
        uint32_t* tile = qweight + ((int64_t)rb * NC + c) * MXQ_BLK_DW;
        tile[mxq_c4(0, r)] = w0;
        tile[mxq_c4(1, r)] = w1;
    }
}

// ------------------------------------------------------------------------------------ //
// Uniform layouts (BASELINE config 5 sweep arms): W2G16 = Quantizer(bits=2, qq_scale_bits=4) on
// every 16-column group; W4ROW = Quantizer(bits=4, qq_scale_bits=4) on whole rows
// (reference lib/quantizer.py:61-147; same arithmetic as the mixed layout's two arms).
// Same workgroup shape as mxq_quantize_pack_kernel: 16 rows x (4 waves over the chunks).
// ------------------------------------------------------------------------------------ //
template <int LAYOUT>
__global__ __launch_bounds__(QP_WAVES * 64) void mxq_quantize_uniform_kernel(const void* __restrict__ W, int dtype,
                                                                   uint32_t* __restrict__ qweight,
                                                                   float4* __restrict__ rowmeta, int N, int K) {
    const int NC = K / 64;
    const int rb = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, qt = lane >> 4;
    const int n = rb * 16 + r;
    constexpr int BLK = LAYOUT == MXQ_LAYOUT_W4ROW ? 128 : 144;
    float v[16];
    if constexpr (LAYOUT == MXQ_LAYOUT_W2G16) {
        for (int c = wave; c < NC; c += QP_WAVES) {
            load16(W, dtype, (int64_t)n * K + c * 64 + qt * 16, v);
            float lo = v[0], hi = v[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) { lo = fminf(lo, v[j]); hi = fmaxf(hi, v[j]); }
            float z, scode, s, qs, qz;
            find_params16(lo, hi, 3.0f, z, scode, s, qs, qz);
            const float sd = fmaxf(s, 1e-9f);
            const uint32_t word = quant2x16(v, sd, z);
            const uint32_t sci = (uint32_t)scode;
            const uint32_t s1 = __shfl(sci, r + 16, 64), s2 = __shfl(sci, r + 32, 64), s3 = __shfl(sci, r + 48, 64);
            uint32_t* blk = qweight + ((int64_t)rb * NC + c) * BLK;
            blk[mxq_w2_c2(qt, r)] = word;
            blk[mxq_w2_z2(qt, r)] = __float_as_uint(z);
            if (r == 0) {
                blk[mxq_qq(qt)] = __float_as_uint(qs);
                blk[mxq_qq(qt) + 1] = __float_as_uint(qz);
            }
            if (qt == 0) ((uint16_t*)blk)[mxq_sc_u16(r)] = (uint16_t)(sci | (s1 << 4) | (s2 << 8) | (s3 << 12));
        }
        if (wave == 0 && qt == 0) rowmeta[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        __shared__ float red[2][QP_WAVES][16];
        float mn = INFINITY, mx = -INFINITY;
        for (int c = wave; c < NC; c += QP_WAVES) {
            load16(W, dtype, (int64_t)n * K + c * 64 + qt * 16, v);
#pragma unroll
            for (int j = 0; j < 16; ++j) { mn = fminf(mn, v[j]); mx = fmaxf(mx, v[j]); }
        }
        mn = fminf(mn, __shfl_xor(mn, 16, 64)); mn = fminf(mn, __shfl_xor(mn, 32, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (qt == 0) { red[0][wave][r] = mn; red[1][wave][r] = mx; }
        __syncthreads();
        float lo = red[0][0][r], hi = red[1][0][r];
#pragma unroll
        for (int w = 1; w < QP_WAVES; ++w) { lo = fminf(lo, red[0][w][r]); hi = fmaxf(hi, red[1][w][r]); }
        float z4, sc4, s4, qs4, qz4;
        find_params16(lo, hi, 15.0f, z4, sc4, s4, qs4, qz4);
        if (wave == 0 && qt == 0) rowmeta[n] = make_float4(z4, sc4, qs4, qz4);
        const float sd4 = fmaxf(s4, 1e-9f);
        for (int c = wave; c < NC; c += QP_WAVES) {
            load16(W, dtype, (int64_t)n * K + c * 64 + qt * 16, v);
            uint32_t w0, w1;
            quant4x16(v, sd4, z4, w0, w1);
            uint32_t* blk = qweight + ((int64_t)rb * NC + c) * BLK;
            blk[mxq_w4_c4(qt, 0, r)] = w0;
            blk[mxq_w4_c4(qt, 1, r)] = w1;
        }
    }
}

// one thread per (row, chunk quarter): fp16 dequant (dense [N, K]) and / or integer unpack
template <int LAYOUT>
__global__ __launch_bounds__(256) void mxq_uniform_expand_kernel(const uint32_t* __restrict__ qweight,
                                                                 const float4* __restrict__ rowmeta,
                                                                 uint16_t* __restrict__ w16, uint8_t* __restrict__ codes,
                                                                 uint8_t* __restrict__ sc, float* __restrict__ zero,
                                                                 float* __restrict__ qs, float* __restrict__ qz, int N,
                                                                 int K) {
    const int NC = K / 64, NC4 = (NC + 3) / 4;
    const int rb = blockIdx.x / NC4, c4 = blockIdx.x % NC4;
    const int t = threadIdx.x;
    const int r = t & 15, cs = (t >> 4) & 3, qt = t >> 6;
    const int n = rb * 16 + r, c = c4 * 4 + cs;
    if (c >= NC) return;
    constexpr int BLK = LAYOUT == MXQ_LAYOUT_W4ROW ? 128 : 144;
    const uint32_t* blk = qweight + ((int64_t)rb * NC + c) * BLK;
    uint32_t o[8];
    const int64_t col0 = (int64_t)c * 64 + qt * 16;
    if constexpr (LAYOUT == MXQ_LAYOUT_W2G16) {
        const uint32_t d = blk[mxq_w2_c2(qt, r)];
        const float z = __uint_as_float(blk[mxq_w2_z2(qt, r)]);
        const uint32_t code = (((const uint16_t*)blk)[mxq_sc_u16(r)] >> (4 * qt)) & 15u;
        const float a = __uint_as_float(blk[mxq_qq(qt)]), b = __uint_as_float(blk[mxq_qq(qt) + 1]);
        mxq_deq2x16(d, mxq_scale(a, b, code), z, o);
        if (codes) {
            for (int k = 0; k < 16; ++k) codes[(int64_t)n * K + col0 + k] = (uint8_t)mxq_code2(d, k);
            const int64_t g = (int64_t)c * 4 + qt;
            sc[(int64_t)n * (K / 16) + g] = (uint8_t)code;
            zero[(int64_t)n * (K / 16) + g] = z;
            if (r == 0) { qs[(int64_t)rb * (K / 16) + g] = a; qz[(int64_t)rb * (K / 16) + g] = b; }
        }
    } else {
        const float4 m = rowmeta[n];
        const float s = mxq_scale(m.z, m.w, (uint32_t)m.y);
        const uint32_t d0 = blk[mxq_w4_c4(qt, 0, r)], d1 = blk[mxq_w4_c4(qt, 1, r)];
        mxq_deq4x8(d0, s, m.x, o);
        mxq_deq4x8(d1, s, m.x, o + 4);
        if (codes) {
            for (int k = 0; k < 8; ++k) {
                codes[(int64_t)n * K + col0 + k] = (uint8_t)mxq_code4(d0, k);
                codes[(int64_t)n * K + col0 + 8 + k] = (uint8_t)mxq_code4(d1, k);
            }
            if (c == 0 && qt == 0) {
                sc[n] = (uint8_t)m.y;
                zero[n] = m.x;
                if (r == 0) { qs[rb] = m.z; qz[rb] = m.w; }
            }
        }
    }
    if (w16) {
        uint4* dst = (uint4*)(w16 + (int64_t)n * K + col0);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

// ------------------------------------------------------------------------------------ //
// exact (v1, 576-B blocks) -> compact (480-B blocks) metadata: a byte shuffle plus one fp16 rounding
// per 2-bit zero-point.  One thread per dword of the compact block; utility kernel, not on the timed path.
// ------------------------------------------------------------------------------------ //
__global__ __launch_bounds__(128) void mxq_compact_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                          int64_t blocks) {
    const int64_t b = blockIdx.x;
    const int t = threadIdx.x;
    if (b >= blocks || t >= MXQC_BLK_DW) return;
    const uint32_t* s = src + b * MXQ_BLK_DW;
    uint32_t v;
    if (t < MXQ_OFF_Z2) v = s[t];                                   // C2, C4: verbatim
    else if (t < MXQC_OFF_SC) {                                     // Z2H: two fp16 zero-points per dword
        const int i = (t - MXQC_OFF_Z2H) * 2;                       // index into [g][r] (48 entries)
        const _Float16 h0 = (_Float16)__uint_as_float(s[MXQ_OFF_Z2 + i]);
        const _Float16 h1 = (_Float16)__uint_as_float(s[MXQ_OFF_Z2 + i + 1]);
        v = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
    } else if (t < MXQC_OFF_QQ) v = s[MXQ_OFF_SC + (t - MXQC_OFF_SC)];
    else v = s[MXQ_OFF_QQ + (t - MXQC_OFF_QQ)];
    dst[b * MXQC_BLK_DW + t] = v;
}

// ------------------------------------------------------------------------------------ //
// launchers
// ------------------------------------------------------------------------------------ //
int mxq_launch_pack_codes(const uint8_t* codes2, const uint8_t* sc2, const float* zero2, const float* qs2,
                          const float* qz2, const uint8_t* codes4, const uint8_t* sc4, const float* zero4,
                          const float* qs4, const float* qz4, void* qweight, void* rowmeta, int N, int K,
                          hipStream_t stream) {
    const int64_t total = (int64_t)N * (K / 64);
    // (every dword of every block is written by the kernel except the unused 4th QQ slot)
    mxq_pack_codes_kernel<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(
        codes2, sc2, zero2, qs2, qz2, codes4, sc4, zero4, qs4, qz4, (uint32_t*)qweight, (float4*)rowmeta, N, K);
    return (int)hipGetLastError();
}

int mxq_launch_unpack(const void* qweight, const void* rowmeta, uint8_t* codes2, uint8_t* sc2, float* zero2,
                      float* qs2, float* qz2, uint8_t* codes4, uint8_t* sc4, float* zero4, float* qs4, float* qz4,
                      int N, int K, int compact, hipStream_t stream) {
    const int64_t total = (int64_t)N * (K / 64);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (compact)
        mxq_unpack_kernel<true><<<grid, 256, 0, stream>>>((const uint32_t*)qweight, (const float4*)rowmeta, codes2, sc2,
                                                          zero2, qs2, qz2, codes4, sc4, zero4, qs4, qz4, N, K);
    else
        mxq_unpack_kernel<false><<<grid, 256, 0, stream>>>((const uint32_t*)qweight, (const float4*)rowmeta, codes2, sc2,
                                                           zero2, qs2, qz2, codes4, sc4, zero4, qs4, qz4, N, K);
    return (int)hipGetLastError();
}

int mxq_launch_dequant_f16(const void* qweight, const void* rowmeta, void* out, int N, int K, int compact,
                           hipStream_t stream) {
    const unsigned grid = (unsigned)((N / 16) * ((K / 64 + 3) / 4));
    if (compact)
        mxq_dequant_f16_kernel<true><<<grid, 256, 0, stream>>>((const uint32_t*)qweight, (const float4*)rowmeta,
                                                               (uint16_t*)out, N, K);
    else
        mxq_dequant_f16_kernel<false><<<grid, 256, 0, stream>>>((const uint32_t*)qweight, (const float4*)rowmeta,
                                                                (uint16_t*)out, N, K);
    return (int)hipGetLastError();
}

int mxq_launch_compact(const void* qweight_exact, void* qweight_compact, int N, int K, hipStream_t stream) {
    const int64_t blocks = (int64_t)(N / 16) * (K / 64);
    if (blocks >= ((int64_t)1 << 31)) return (int)hipErrorInvalidValue;
    mxq_compact_kernel<<<(unsigned)blocks, 128, 0, stream>>>((const uint32_t*)qweight_exact, (uint32_t*)qweight_compact,
                                                             blocks);
    return (int)hipGetLastError();
}

int mxq_launch_quantize_pack(const void* W, int dtype, const uint8_t* dead, void* qweight, void* rowmeta, int N,
                             int K, hipStream_t stream) {
    mxq_quantize_pack_kernel<<<(unsigned)(N / 16), QP_WAVES * 64, 0, stream>>>(W, dtype, dead, (uint32_t*)qweight,
                                                                      (float4*)rowmeta, N, K);
    return (int)hipGetLastError();
}

int mxq_launch_quantize_uniform(const void* W, int dtype, void* qweight, void* rowmeta, int N, int K, int layout,
                                hipStream_t stream) {
    if (layout == MXQ_LAYOUT_W2G16)
        mxq_quantize_uniform_kernel<MXQ_LAYOUT_W2G16><<<(unsigned)(N / 16), QP_WAVES * 64, 0, stream>>>(
            W, dtype, (uint32_t*)qweight, (float4*)rowmeta, N, K);
    else
        mxq_quantize_uniform_kernel<MXQ_LAYOUT_W4ROW><<<(unsigned)(N / 16), QP_WAVES * 64, 0, stream>>>(
            W, dtype, (uint32_t*)qweight, (float4*)rowmeta, N, K);
    return (int)hipGetLastError();
}

int mxq_launch_uniform_expand(const void* qweight, const void* rowmeta, void* w16, uint8_t* codes, uint8_t* sc,
                              float* zero, float* qs, float* qz, int N, int K, int layout, hipStream_t stream) {
    const unsigned grid = (unsigned)((N / 16) * ((K / 64 + 3) / 4));
    if (layout == MXQ_LAYOUT_W2G16)
        mxq_uniform_expand_kernel<MXQ_LAYOUT_W2G16><<<grid, 256, 0, stream>>>(
            (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)w16, codes, sc, zero, qs, qz, N, K);
    else
        mxq_uniform_expand_kernel<MXQ_LAYOUT_W4ROW><<<grid, 256, 0, stream>>>(
            (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)w16, codes, sc, zero, qs, qz, N, K);
    return (int)hipGetLastError();
}
