// Skinny quantised Linear for 4 < M <= 48 tokens, up to 64 on request (small-batch decode / speculative verification):
//   y[m, n] = sum_k x[m, k] * fp16(scale * (q - zero))[n, k], fp32 accumulate, fp16 out.
//
// The reference's kernel handles batch rows by re-reading the weights once per row (gridDim.z,
// mxq_quant/cuda_kernel/csrc/quantization/gemv_mxq_cuda.cu:261-262); the streaming GEMV here (gemv.hip) shares one
// pass over up to 4 rows, but its v_dot2 work grows with M, and the prefill kernel's 256-token tile wastes the
// matrix pipe on 5..32 tokens (64 tokens x 4096^2: 34 us against 7 us for 4 tokens).  This kernel keeps the
// GEMV's economy -- every packed byte read once, straight from HBM into registers, no LDS copy of the weights --
// and hands the products to the matrix cores:
//   * one workgroup per 16-row block, its waves split K by chunks (as the GEMV);
//   * a lane dequantises EXACTLY the MFMA A-operand it owns: v_mfma_f32_16x16x32_f16 wants lane (r = lane & 15,
//     kq = lane >> 4) to hold W[row r][8 kq .. 8 kq + 7] of a 32-wide K slice, i.e. half of a 16-column group.  K
//     slice 0 of a chunk is groups 0 / 1 (kq >> 1) half (kq & 1); slice 1 is group 2 (kq < 2) or one of the two
//     four-bit code words (kq >= 2).  The 4-entry LUT of a 2-bit group is built by both lanes that share it;
//   * the B operand (8 consecutive activations of token lane & 15) comes straight from global memory / L2 with one
//     16-byte load per lane (x is at most 32 x K fp16: L2-resident, too big for an LDS copy at 32 tokens);
//   * per chunk 2 x MT MFMAs (MT = 1..4 blocks of 16 tokens; beyond 48 tokens the prefill kernel's tile is the faster
//     one on the MLP shapes: 54 vs 51 us at 64 tokens, but 42 vs 50 us at 48); the waves' partial tiles meet in LDS.
// D^T = W . x^T as in the prefill kernel: a lane ends with 4 consecutive channels of one token.
// Layouts: mixed with exact (v1) or compact metadata, and the uniform layouts of BASELINE config 5: W2G16 (both K slices
// of a chunk are 2-bit groups: slice 1 holds groups 2 + (kq >> 1)) and W4ROW (every lane converts one 4-bit code word per
// slice with the row's scale / zero-point) -- the reference serves uniform W4 at any batch (gemv_cuda.cu:346-399).
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// half `h` (8 weights) of a 2-bit group -> 4 packed fp16 pairs
MXQ_HD void deq2_half(uint32_t d, int h, float s, float z, uint32_t o[4]) {
    const uint32_t p01 = mxq_pack_f16(s * (0.0f - z), s * (1.0f - z));
    const uint32_t p23 = mxq_pack_f16(s * (2.0f - z), s * (3.0f - z));
    const uint32_t lut_lo = MXQ_PERM(p23, p01, 0x06040200u);
    const uint32_t lut_hi = MXQ_PERM(p23, p01, 0x07050301u);
    const uint32_t dd = d >> (4 * h);                     // elements 8h .. 8h+7 sit at bit 8(k&3) + 2(k>>2), k>>2 in {2h, 2h+1}
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t m = (dd >> (2 * j)) & 0x03030303u;
        const uint32_t lo = MXQ_PERM(0u, lut_lo, m), hi = MXQ_PERM(0u, lut_hi, m);
        o[2 * j] = MXQ_PERM(hi, lo, 0x05010400u);
        o[2 * j + 1] = MXQ_PERM(hi, lo, 0x07030602u);
    }
}

struct Chunk {            // one lane's packed words of one chunk
    uint32_t c0, z0;      // slice 0: code word and zero-point (fp32 bits) of group kq >> 1
    uint32_t c1, z1;      // slice 1: group 2's (kq < 2) or the four-bit word kq - 2 (z1 unused)
    uint32_t scw;
    uint32_t qs0, qz0, qs2, qz2;
};

template <int MT, int THREADS, int LAYOUT>
__global__ __launch_bounds__(THREADS) void mxq_skinny_f16_kernel(const uint16_t* __restrict__ x,
                                                                 const uint32_t* __restrict__ qweight,
                                                                 const float4* __restrict__ rowmeta,
                                                                 uint16_t* __restrict__ y, int M, int N, int K) {
    constexpr int WAVES = THREADS / 64;
    constexpr bool COMPACT = LAYOUT == MXQ_LAYOUT_MIXEDC;
    constexpr bool MIXED = LAYOUT == MXQ_LAYOUT_MIXED || COMPACT, W2 = LAYOUT == MXQ_LAYOUT_W2G16, W4 = LAYOUT == MXQ_LAYOUT_W4ROW;
    constexpr int BLK_DW = W4 ? 128 : COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    typedef MxqMixed<COMPACT> F;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [WAVES][MT][16 rows][16 tokens] fp32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int rb = blockIdx.x;
    const int NC = K / 64;
    const uint32_t* tiles = qweight + (int64_t)rb * NC * BLK_DW;
    const int g0 = kq >> 1, h0 = kq & 1;
    const int g1 = W2 ? 2 + (kq >> 1) : 2;                        // slice 1's 2-bit group
    const bool four = MIXED && kq >= 2;                           // mixed: slice 1 of lanes kq >= 2 is the four-bit quarter
    // lane-dependent dword offsets of the slice-1 words (both arms load SOMETHING: no divergent loads)
    const int off_c1 = W4 ? mxq_w4_c4(2 + (kq >> 1), kq & 1, r) : W2 ? mxq_w2_c2(g1, r) : four ? mxq_c4(kq - 2, r) : mxq_c2(2, r);

    auto load_chunk = [&](int c) {
        Chunk k = {};
        if (c < NC) {
            const uint32_t* t = tiles + (int64_t)c * BLK_DW;
            k.c1 = t[off_c1];
            if constexpr (W4) {
                k.c0 = t[mxq_w4_c4(kq >> 1, kq & 1, r)];
            } else if constexpr (W2) {                            // SC / QQ sit where the mixed layout has them
                k.c0 = t[mxq_w2_c2(g0, r)];
                k.z0 = t[mxq_w2_z2(g0, r)];
                k.z1 = t[mxq_w2_z2(g1, r)];
                k.scw = ((const uint16_t*)t)[mxq_sc_u16(r)];
                k.qs0 = t[mxq_qq(g0)];
                k.qz0 = t[mxq_qq(g0) + 1];
                k.qs2 = t[mxq_qq(g1)];
                k.qz2 = t[mxq_qq(g1) + 1];
            } else {
                k.c0 = t[mxq_c2(g0, r)];
                k.z0 = __float_as_uint(F::z2(t, g0, r));
                k.z1 = __float_as_uint(F::z2(t, 2, r));
                k.scw = F::scw(t, r);
                k.qs0 = t[F::qq(g0)];
                k.qz0 = t[F::qq(g0) + 1];
                k.qs2 = t[F::qq(2)];
                k.qz2 = t[F::qq(2) + 1];
            }
        }
        return k;
    };
    // the lane's B-operand rows: token lane & 15 of block tb (clamped: rows beyond M compute garbage nobody stores)
    const uint16_t* xrow[MT];
#pragma unroll
    for (int tb = 0; tb < MT; ++tb) {
        int m = tb * 16 + r;
        m = m < M ? m : M - 1;
        xrow[tb] = x + (int64_t)m * K + kq * 8;
    }
    auto load_x = [&](int c, half8 (&b)[MT][2]) {
        const int cc = c < NC ? c : NC - 1;
#pragma unroll
        for (int tb = 0; tb < MT; ++tb) {
            b[tb][0] = *(const half8*)(xrow[tb] + cc * 64);
            b[tb][1] = *(const half8*)(xrow[tb] + cc * 64 + 32);
        }
    };

    const float4 rm = rowmeta[rb * 16 + r];
    const float s4 = mxq_scale(rm.z, rm.w, (uint32_t)rm.y), z4 = rm.x;

    f32x4 acc[MT];
#pragma unroll
    for (int tb = 0; tb < MT; ++tb) acc[tb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Chunk cur = load_chunk(wave);
    half8 bx[MT][2];
    load_x(wave, bx);
    for (int c = wave; c < NC; c += WAVES) {
        const Chunk nxt = load_chunk(c + WAVES);          // next chunk in flight during the math
        half8 bn[MT][2];
        load_x(c + WAVES, bn);
        uint32_t o[4];
        // slice 0: columns 0..31 of the chunk
        if constexpr (W4) mxq_deq4x8(cur.c0, s4, z4, o);
        else deq2_half(cur.c0, h0, mxq_scale(__uint_as_float(cur.qs0), __uint_as_float(cur.qz0), (cur.scw >> (4 * g0)) & 15u),
                       __uint_as_float(cur.z0), o);
        half8 a0 = __builtin_bit_cast(half8, (u32x4){o[0], o[1], o[2], o[3]});
        // slice 1: columns 32..63: mixed: group 2 (lanes kq < 2) or the four-bit quarter (kq >= 2); W2G16: group 2 + (kq >> 1)
        if (W4 || four) mxq_deq4x8(cur.c1, s4, z4, o);
        else deq2_half(cur.c1, MIXED ? kq : h0, mxq_scale(__uint_as_float(cur.qs2), __uint_as_float(cur.qz2), (cur.scw >> (4 * g1)) & 15u),
                       __uint_as_float(cur.z1), o);
        half8 a1 = __builtin_bit_cast(half8, (u32x4){o[0], o[1], o[2], o[3]});
#pragma unroll
        for (int tb = 0; tb < MT; ++tb) {
            acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bx[tb][0], acc[tb], 0, 0, 0);
            acc[tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bx[tb][1], acc[tb], 0, 0, 0);
        }
        cur = nxt;
#pragma unroll
        for (int tb = 0; tb < MT; ++tb) { bx[tb][0] = bn[tb][0]; bx[tb][1] = bn[tb][1]; }
    }

    // D^T[W row 4 kq + i][token r]: the waves' partial tiles meet in LDS, [wave][tb][token][row]
    float* red = (float*)smem;
#pragma unroll
    for (int tb = 0; tb < MT; ++tb) *(f32x4*)(red + ((wave * MT + tb) * 16 + r) * 16 + kq * 4) = acc[tb];
    __syncthreads();
    for (int i = tid; i < MT * 256; i += THREADS) {       // i -> (tb, token, row): consecutive threads, consecutive channels
        const int tb = i >> 8, tok = (i >> 4) & 15, row = i & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += red[((w * MT + tb) * 16 + tok) * 16 + row];
        const int m = tb * 16 + tok;
        if (m < M) {
            const _Float16 hv = (_Float16)v;
            y[(int64_t)m * N + rb * 16 + row] = __builtin_bit_cast(uint16_t, hv);
        }
    }
}

template <int MT, int THREADS, int LAYOUT>
int launch_t(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, hipStream_t stream) {
    const size_t smem = (size_t)(THREADS / 64) * MT * 256 * 4;
    mxq_skinny_f16_kernel<MT, THREADS, LAYOUT><<<N / 16, THREADS, smem, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K);
    return (int)hipGetLastError();
}

template <int LAYOUT>
int launch_c(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, hipStream_t stream) {
    const bool big = N / 16 > 384;     // many row blocks: 8 waves, so that several workgroups share a CU
    if (M <= 16)
        return big ? launch_t<1, 512, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream)
                   : launch_t<1, 1024, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    if (M <= 32)
        return big ? launch_t<2, 512, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream)
                   : launch_t<2, 1024, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    if (M <= 48)
        return big ? launch_t<3, 512, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream)
                   : launch_t<3, 1024, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    return launch_t<4, 512, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);   // (16 waves would spill at 128 VGPRs)
}

}   // namespace

// 1 <= M <= 64; any layout
int mxq_launch_skinny_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                          int layout, hipStream_t stream) {
    if (M < 1 || M > 64) return -1;
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launch_c<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_MIXEDC: return launch_c<MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_W2G16: return launch_c<MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_W4ROW: return launch_c<MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, stream);
    }
    return -1;
}
