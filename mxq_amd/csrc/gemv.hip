// Decode-path GEMV / skinny GEMM (M <= 4) on the v1 packed format:
//   y[m, n] = sum_k x[m, k] * fp16(scale * (q - zero))[n, k], fp32 accumulate, fp16 out.
//
// Native counterpart of the reference's fused unpack+dot kernel
// gemv_mxq_kernel_g16_v0 (mxq_quant/cuda_kernel/csrc/quantization/gemv_mxq_cuda.cu:39-208)
// -- same idea (never materialise fp16 weights in memory), different everything else:
//   * HBM-bound: every packed byte is read exactly once; a wave consumes 4 consecutive
//     16x64 blocks (2304 contiguous bytes) per iteration, straight to VGPRs
//     (no LDS round trip for weights; cdna guide section 5, "GEMV / M <= 16" row).
//   * one workgroup per 16-row block (N/16 >= 256 workgroups for Llama shapes), 16 waves
//     split K; lane -> (row r = lane & 15, chunk slot cs = lane >> 4).
//   * activations are staged once per workgroup in LDS as fp16 and read as broadcast
//     ds_read_b128; products use v_dot2_f32_f16 on the LUT-selected fp16 pairs.
//   * wave64 reduction: 2 xor-shuffles over the 4 chunk slots, then 16 waves through LDS.
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));


__device__ __forceinline__ float dot8(const uint32_t* w, const uint4 xa, float acc) {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w[0]), __builtin_bit_cast(half2v, xa.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w[1]), __builtin_bit_cast(half2v, xa.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w[2]), __builtin_bit_cast(half2v, xa.z), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w[3]), __builtin_bit_cast(half2v, xa.w), acc, false);
    return acc;
}

// GEMV_THREADS: 1024 (16 waves: one 16x256 tile per wave at K = 4096, every load in flight at
// once) when there are few row blocks, 512 when N/16 alone oversubscribes the chip.
// PRO: prologue fused into the activation staging (decode, M = 1 only):
//   0 = none; 1 = RMSNorm: x <- fp16(x * rsqrt(mean(x^2) + eps)) * norm_w (every workgroup
//   recomputes the 4096-element reduction -- cheaper than a separate launch);
//   2 = SwiGLU gate: the input row is [2K] = (gate, up) and x <- fp16(silu(gate)) * up.
// residual (nullable): y <- residual + W.x (the decoder layer's skip connection).
// LAYOUT: MXQ_LAYOUT_MIXED (3 two-bit groups + the 4-bit quarter per chunk), MXQ_LAYOUT_W2G16 (4 two-bit groups)
// or MXQ_LAYOUT_W4ROW (4 four-bit quarters, scale / zero per row from rowmeta): csrc/mxq_format.h.
template <int MB, int GEMV_THREADS, int PRO, int LAYOUT = MXQ_LAYOUT_MIXED>
__global__ __launch_bounds__(GEMV_THREADS) void mxq_gemv_f16_kernel(const uint16_t* __restrict__ x,
                                                                     const uint32_t* __restrict__ qweight,
                                                                     const float4* __restrict__ rowmeta,
                                                                     uint16_t* __restrict__ y, int M, int N, int K,
                                                                     const uint16_t* __restrict__ norm_w, float eps,
                                                                     const uint16_t* __restrict__ residual) {
    constexpr int GEMV_WAVES = GEMV_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // MB*K halfs, then reduction scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, cs = lane >> 4;
    const int rb = blockIdx.x;
    const int NC = K / 64, NC4 = (NC + 3) / 4;

    // packed operands of one (row, chunk): 13 registers (mixed layout), loaded straight from HBM
    constexpr bool MIXED = LAYOUT == MXQ_LAYOUT_MIXED || LAYOUT == MXQ_LAYOUT_MIXEDC;   // exact / compact metadata
    constexpr bool COMPACT = LAYOUT == MXQ_LAYOUT_MIXEDC;
    constexpr int NG2 = MIXED ? 3 : LAYOUT == MXQ_LAYOUT_W2G16 ? 4 : 0;   // two-bit groups per chunk
    constexpr int NW4 = MIXED ? 2 : LAYOUT == MXQ_LAYOUT_W4ROW ? 8 : 0;   // four-bit code words per chunk
    constexpr int BLK_DW = LAYOUT == MXQ_LAYOUT_W4ROW ? 128 : COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    struct Tile {
        uint32_t c2w[NG2 ? NG2 : 1], z2w[NG2 ? NG2 : 1], c4w[NW4 ? NW4 : 1], scw;
        uint2 qq[NG2 ? NG2 : 1];
    };
    const uint32_t* tiles = qweight + (int64_t)rb * NC * BLK_DW;
    auto load_tile = [&](int c4) {
        Tile t = {};
        const int chunk = c4 * 4 + cs;
        if (chunk < NC) {   // ragged tail: K/64 not a multiple of 4
            // the wave reads 4 consecutive blocks = 2304 contiguous bytes; each load touches
            // four 64-B segments (one per chunk slot)
            const uint32_t* tile = tiles + (int64_t)chunk * BLK_DW;
            if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
#pragma unroll
                for (int i = 0; i < 8; ++i) t.c4w[i] = tile[mxq_w4_c4(i >> 1, i & 1, r)];
            } else {
#pragma unroll
                for (int g = 0; g < NG2; ++g) {
                    t.c2w[g] = tile[MIXED ? mxq_c2(g, r) : mxq_w2_c2(g, r)];
                    if constexpr (COMPACT) {   // fp16 zero-point, widened once here (the arithmetic below is fp32 either way)
                        const uint16_t zh = ((const uint16_t*)tile)[mxqc_z2_u16(g, r)];
                        t.z2w[g] = __float_as_uint((float)__builtin_bit_cast(_Float16, zh));
                    } else {
                        t.z2w[g] = tile[MIXED ? mxq_z2(g, r) : mxq_w2_z2(g, r)];
                    }
                    t.qq[g] = *(const uint2*)(tile + (COMPACT ? mxqc_qq(g) : mxq_qq(g)));   // SC / QQ: same offsets in v1 and W2G16
                }
                if constexpr (MIXED) {
                    t.c4w[0] = tile[mxq_c4(0, r)];
                    t.c4w[1] = tile[mxq_c4(1, r)];
                }
                t.scw = ((const uint16_t*)tile)[COMPACT ? mxqc_sc_u16(r) : mxq_sc_u16(r)];
            }
        }
        return t;
    };

    // the weight stream is started BEFORE the activations are staged, so the HBM latency of
    // the first tile overlaps the x copy and the barrier
    Tile cur = load_tile(wave);
    const float4 rm = rowmeta[rb * 16 + r];

    // stage x[0..MB) in LDS (rows beyond M are zero)
    if constexpr (PRO == 0) {
        const int vec_per_row = K / 8;
        for (int i = tid; i < MB * vec_per_row; i += GEMV_THREADS) {
            const int m = i / vec_per_row, v = i % vec_per_row;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (m < M) val = *(const uint4*)(x + (int64_t)m * K + v * 8);
            *(uint4*)(smem + (size_t)i * 16) = val;
        }
    } else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        float* wsum = (float*)(smem + (size_t)K * 2);   // reduction scratch (reused by the final reduce)
        float ss = 0.f;
        for (int v = tid; v < K / 8; v += GEMV_THREADS) {
            h8 a = *(const h8*)(x + v * 8);
            if constexpr (PRO == 2) {
                const h8 u = *(const h8*)(x + K + v * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = (float)a[j];
                    a[j] = (_Float16)(g / (1.0f + __expf(-g))) * u[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += (float)a[j] * (float)a[j];
            }
            *(h8*)(smem + (size_t)v * 16) = a;
        }
        if constexpr (PRO == 1) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (lane == 0) wsum[wave] = ss;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < GEMV_WAVES; ++w) tot += wsum[w];
            const float inv = rsqrtf(tot / (float)K + eps);
            for (int v = tid; v < K / 8; v += GEMV_THREADS) {
                h8 a = *(h8*)(smem + (size_t)v * 16);
                const h8 g = *(const h8*)(norm_w + v * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (_Float16)((float)a[j] * inv) * g[j];
                *(h8*)(smem + (size_t)v * 16) = a;
            }
        }
    }
    const float s4 = mxq_scale(rm.z, rm.w, (uint32_t)rm.y), z4 = rm.x;
    __syncthreads();

    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;

    for (int c4 = wave; c4 < NC4; c4 += GEMV_WAVES) {
        Tile nxt = {};
        if (c4 + GEMV_WAVES < NC4) nxt = load_tile(c4 + GEMV_WAVES);   // next tile in flight during the math
        const int chunk = c4 * 4 + cs;
        if (chunk < NC) {
            const char* xk = smem + (size_t)chunk * 128;
            uint32_t o[8];
#pragma unroll
            for (int g = 0; g < NG2; ++g) {
                mxq_deq2x16(cur.c2w[g],
                            mxq_scale(__uint_as_float(cur.qq[g].x), __uint_as_float(cur.qq[g].y),
                                      (cur.scw >> (4 * g)) & 15u),
                            __uint_as_float(cur.z2w[g]), o);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const uint4 xa = *(const uint4*)(xk + (size_t)m * K * 2 + g * 32);
                    const uint4 xb = *(const uint4*)(xk + (size_t)m * K * 2 + g * 32 + 16);
                    acc[m] = dot8(o, xa, acc[m]);
                    acc[m] = dot8(o + 4, xb, acc[m]);
                }
            }
#pragma unroll
            for (int q = 0; q < NW4 / 2; ++q) {      // 16 four-bit weights per pair of code words
                constexpr int X0 = MIXED ? 96 : 0;   // the mixed layout's quarter is the chunk's last
                mxq_deq4x8(cur.c4w[2 * q], s4, z4, o);
                mxq_deq4x8(cur.c4w[2 * q + 1], s4, z4, o + 4);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const uint4 xa = *(const uint4*)(xk + (size_t)m * K * 2 + X0 + q * 32);
                    const uint4 xb = *(const uint4*)(xk + (size_t)m * K * 2 + X0 + q * 32 + 16);
                    acc[m] = dot8(o, xa, acc[m]);
                    acc[m] = dot8(o + 4, xb, acc[m]);
                }
            }
        }
        cur = nxt;
    }

    // reduce over the 4 chunk slots of the wave, then over waves
    float* red = (float*)(smem + (size_t)MB * K * 2);   // [GEMV_WAVES][MB][16]
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        float v = acc[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (cs == 0) red[(wave * MB + m) * 16 + r] = v;
    }
    __syncthreads();
    if (tid < MB * 16) {
        const int m = tid >> 4, rr = tid & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < GEMV_WAVES; ++w) v += red[(w * MB + m) * 16 + rr];
        if (m < M) {
            _Float16 h = (_Float16)v;
            if (residual) h = __builtin_bit_cast(_Float16, residual[(int64_t)m * N + rb * 16 + rr]) + h;
            y[(int64_t)m * N + rb * 16 + rr] = __builtin_bit_cast(uint16_t, h);
        }
    }
}

template <int MB, int THREADS, int PRO, int LAYOUT = MXQ_LAYOUT_MIXED>
int launch_t(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
             const void* norm_w, float eps, const void* residual, hipStream_t stream) {
    const size_t smem = (size_t)MB * K * 2 + (size_t)(THREADS / 64) * MB * 16 * 4;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mxq_gemv_f16_kernel<MB, THREADS, PRO, LAYOUT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    mxq_gemv_f16_kernel<MB, THREADS, PRO, LAYOUT><<<N / 16, THREADS, smem, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K,
        (const uint16_t*)norm_w, eps, (const uint16_t*)residual);
    return (int)hipGetLastError();
}

// Workgroup size by row-block count: every workgroup should be resident at once (a second, partial round of
// workgroups costs a whole workgroup lifetime).  <= 384 row blocks: 16 waves (one workgroup per CU, every load of a
// K = 4096 row block in flight at once); <= 768: 8 waves (3 per CU at ~68 VGPRs); more (fused gate|up: 1376): 4 waves
// (7 per CU = 1792 slots), each wave then walks 4+ tiles with the next one's loads in flight.
__host__ inline int gemv_threads(int N, int forced) {
    if (forced) return forced;
    const int rbs = N / 16;
    return rbs <= 384 ? 1024 : rbs <= 768 ? 512 : 256;
}

template <int MB, int LAYOUT = MXQ_LAYOUT_MIXED>
int launch(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, hipStream_t stream,
           int threads = 0) {
    switch (gemv_threads(N, threads)) {
        case 1024: return launch_t<MB, 1024, 0, LAYOUT>(x, qweight, rowmeta, y, M, N, K, nullptr, 0.f, nullptr, stream);
        case 512: return launch_t<MB, 512, 0, LAYOUT>(x, qweight, rowmeta, y, M, N, K, nullptr, 0.f, nullptr, stream);
        case 256: return launch_t<MB, 256, 0, LAYOUT>(x, qweight, rowmeta, y, M, N, K, nullptr, 0.f, nullptr, stream);
    }
    return (int)hipErrorInvalidValue;
}

template <int LAYOUT>
int launch_layout(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                  hipStream_t stream) {
    if (M == 1) return launch<1, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    if (M == 2) return launch<2, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    if (M <= 4) return launch<4, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    return (int)hipErrorInvalidValue;
}

}   // namespace

int mxq_launch_gemv_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                        hipStream_t stream) {
    if (M == 1) return launch<1>(x, qweight, rowmeta, y, M, N, K, stream);
    if (M == 2) return launch<2>(x, qweight, rowmeta, y, M, N, K, stream);
    if (M <= 4) return launch<4>(x, qweight, rowmeta, y, M, N, K, stream);
    return (int)hipErrorInvalidValue;
}

int mxq_launch_gemv_layout_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                               int layout, hipStream_t stream) {
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launch_layout<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_W2G16: return launch_layout<MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_W4ROW: return launch_layout<MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_MIXEDC: return launch_layout<MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, M, N, K, stream);
    }
    return (int)hipErrorInvalidValue;
}

template <int LAYOUT>
static int fused_layout(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K, int prologue,
                        const void* norm_w, float eps, const void* residual, hipStream_t stream) {
    const int th = gemv_threads(N, 0);
#define MXQ_FUSED(PRO)                                                                                                   \
    (th == 1024 ? launch_t<1, 1024, PRO, LAYOUT>(x, qweight, rowmeta, y, 1, N, K, norm_w, eps, residual, stream)         \
     : th == 512 ? launch_t<1, 512, PRO, LAYOUT>(x, qweight, rowmeta, y, 1, N, K, norm_w, eps, residual, stream)         \
                 : launch_t<1, 256, PRO, LAYOUT>(x, qweight, rowmeta, y, 1, N, K, norm_w, eps, residual, stream))
    switch (prologue) {
        case 0: return MXQ_FUSED(0);
        case 1: return MXQ_FUSED(1);
        case 2: return MXQ_FUSED(2);
    }
#undef MXQ_FUSED
    return (int)hipErrorInvalidValue;
}

int mxq_launch_gemv_fused_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K,
                              int prologue, const void* norm_w, float eps, const void* residual, int compact,
                              hipStream_t stream) {
    return compact ? fused_layout<MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, N, K, prologue, norm_w, eps, residual, stream)
                   : fused_layout<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, N, K, prologue, norm_w, eps, residual, stream);
}

#ifdef MXQ_PROFILING
// A/B entry for tools/ (correct results): explicit workgroup size (256 / 512 / 1024 threads), M = 1
extern "C" int mxq_prof_gemv_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                 int threads, void* stream) {
    if (M != 1) return -1;
    return launch<1>(x, qweight, rowmeta, y, M, N, K, (hipStream_t)stream, threads);
}
#endif
