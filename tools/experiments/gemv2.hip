// EXPERIMENT RECORD (round 2) -- not part of the product library, not built by the Makefile.
// Outcome: correct (<= 1e-3 vs the dequantised weight on all five Llama GEMV shapes) but 8-25 % SLOWER than gemv.hip
// (4096^2: 6.9 vs 5.5 us; fused gate|up 22016 x 4096: 18.7 vs 16.8 us; gpurun_out/r2_gemv1.log).  tools/probes/
// stream_probe.hip then showed why the premise was wrong: the 4-byte-load pattern of gemv.hip already streams at the
// rate of an ideal 16-byte grid-stride read (5.2-5.4 TB/s loads-only); what the GEMV lacks is overlap between its
// ~230 VALU ops per tile and the load latency, not load width.
//
// Decode-path GEMV / skinny product (M <= 8 tokens), v2: 16-byte weight loads, activations shared by 4 rows.
//
//   y[m, n] = sum_k x[m, k] * fp16(scale * (q - zero))[n, k], fp32 accumulate, fp16 out.
//
// Native counterpart of the reference's fused unpack+dot kernel gemv_mxq_kernel_g16_v0
// (mxq_quant/cuda_kernel/csrc/quantization/gemv_mxq_cuda.cu:39-208); the reference handles batch rows by
// re-reading the weights per row (gridDim.z, :261-262), here up to 8 rows share one pass over the weights.
//
// Round 1's kernel mapped a lane to (row, chunk) and pulled its 13 packed words with 13 four-byte loads (64-B
// segments): loads alone ran at 4.0 TB/s.  The v1 block is field-major per 16 rows (csrc/mxq_format.h), so
// the same bytes come in 16-B pieces if a lane takes FOUR ROWS of ONE field instead:
//   * a team = 4 waves walks 16 consecutive chunks (blocks) of one 16-row block per iteration;
//     lane -> (chunk b = lane / 4, row quad q = lane % 4); wave role g (wave-uniform, no divergence):
//       g = 0..2: two-bit group g      C2[g][4q..4q+3] (16 B), Z2[g][4q..4q+3] (16 B), SC[4q..4q+3] (8 B), QQ[g] (8 B)
//       g = 3:    the four-bit quarter C4[0][4q..4q+3] (16 B), C4[1][4q..4q+3] (16 B)
//     -> 2 x dwordx4 + 2 x dwordx2 per lane and iteration instead of 13 x dword;
//   * the lane's 16 activations (32 B of the LDS copy) serve its 4 rows: 2 LDS reads per 64 weights, not 8;
//   * the next iteration's loads are in flight while this one is dequantised (LUT / v_perm_b32, v_dot2_f32_f16);
//   * reduction: xor-shuffles over the 16 chunk lanes, then the waves through LDS.
// One workgroup per 16-row block; TEAMS teams split K.  PRO: the decode harness' fused prologues (RMSNorm /
// SwiGLU); residual: fused skip connection.  LAYOUT: mixed, W2G16 (role 3 is a fourth two-bit group), W4ROW.
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XPITCH = 144;   // bytes per chunk of the LDS activation copy: 128 + 16 keeps the 16 chunk lanes off each other's banks

// (hipcc, ROCm 7.2: __builtin_bit_cast applied directly to an element of an ext_vector_type value reads element 0;
// the elements are copied to scalars first)
__device__ __forceinline__ float dot8(const uint32_t* w, const u32x4 xa, float acc) {
    const uint32_t x0 = xa[0], x1 = xa[1], x2 = xa[2], x3 = xa[3];
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w[0]), __builtin_bit_cast(half2v, x0), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w[1]), __builtin_bit_cast(half2v, x1), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w[2]), __builtin_bit_cast(half2v, x2), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w[3]), __builtin_bit_cast(half2v, x3), acc, false);
    return acc;
}

struct Tile {
    u32x4 a, b;      // role < 3: code words / zero-points of rows 4q..4q+3; four-bit role: code words h = 0 / h = 1
    u32x2 sc, qq;    // role < 3: 4 x u16 scale codes, (qs, qz) of the group
};

template <int MB, int TEAMS, int PRO, int LAYOUT>
__global__ __launch_bounds__(TEAMS * 256) void mxq_gemv2_f16_kernel(const uint16_t* __restrict__ x,
                                                                    const uint32_t* __restrict__ qweight,
                                                                    const float4* __restrict__ rowmeta,
                                                                    uint16_t* __restrict__ y, int M, int N, int K,
                                                                    const uint16_t* __restrict__ norm_w, float eps,
                                                                    const uint16_t* __restrict__ residual) {
    constexpr int THREADS = TEAMS * 256, WAVES = TEAMS * 4;
    constexpr int BLK_DW = LAYOUT == MXQ_LAYOUT_W4ROW ? 128 : MXQ_BLK_DW;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // MB x (K/64 chunks x XPITCH), then reduction scratch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2, role = wave & 3;                  // wave-uniform
    const int b = lane >> 2, q = lane & 3;
    const int rb = blockIdx.x;
    const int NC = K / 64;
    const int xrow = NC * XPITCH;                                 // bytes of one token's LDS copy
    const bool four = LAYOUT == MXQ_LAYOUT_W4ROW || (LAYOUT == MXQ_LAYOUT_MIXED && role == 3);

    // dword offsets of this lane's two 16-B pieces inside a block
    int offA, offB;
    if constexpr (LAYOUT == MXQ_LAYOUT_MIXED) {
        offA = four ? MXQ_OFF_C4 + 4 * q : MXQ_OFF_C2 + role * 16 + 4 * q;
        offB = four ? MXQ_OFF_C4 + 16 + 4 * q : MXQ_OFF_Z2 + role * 16 + 4 * q;
    } else if constexpr (LAYOUT == MXQ_LAYOUT_W2G16) {
        offA = role * 16 + 4 * q;
        offB = 64 + role * 16 + 4 * q;
    } else {   // W4ROW: role = quarter
        offA = (role * 2) * 16 + 4 * q;
        offB = (role * 2 + 1) * 16 + 4 * q;
    }
    const uint32_t* tiles = qweight + (int64_t)rb * NC * BLK_DW;
    auto load_tile = [&](int it) {
        Tile t = {};
        const int chunk = (it * TEAMS + team) * 16 + b;
        if (chunk < NC) {
            const uint32_t* p = tiles + (int64_t)chunk * BLK_DW;
            t.a = *(const u32x4*)(p + offA);
            t.b = *(const u32x4*)(p + offB);
            if (!four) {
                t.sc = *(const u32x2*)(p + MXQ_OFF_SC + 2 * q);
                t.qq = *(const u32x2*)(p + MXQ_OFF_QQ + 2 * role);
            }
        }
        return t;
    };
    const int n_it = (NC + 16 * TEAMS - 1) / (16 * TEAMS);

    // the weight stream starts BEFORE the activations are staged: its HBM latency overlaps the x copy
    Tile cur = load_tile(0);
    float s4[4], z4[4];
    if (four) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 rm = rowmeta[rb * 16 + 4 * q + j];
            s4[j] = mxq_scale(rm.z, rm.w, (uint32_t)rm.y);
            z4[j] = rm.x;
        }
    }

    // stage x[0..MB) in LDS, chunk pitch XPITCH (rows beyond M are zero)
    if constexpr (PRO == 0) {
        const int vec_per_row = K / 8;
        for (int i = tid; i < MB * vec_per_row; i += THREADS) {
            const int m = i / vec_per_row, v = i % vec_per_row;
            u32x4 val = {0, 0, 0, 0};
            if (m < M) val = *(const u32x4*)(x + (int64_t)m * K + v * 8);
            *(u32x4*)(smem + m * xrow + (v >> 3) * XPITCH + (v & 7) * 16) = val;
        }
    } else {
        float* wsum = (float*)(smem + xrow);   // reduction scratch (reused by the final reduce)
        float ss = 0.f;
        for (int v = tid; v < K / 8; v += THREADS) {
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
            *(h8*)(smem + (v >> 3) * XPITCH + (v & 7) * 16) = a;
        }
        if constexpr (PRO == 1) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (lane == 0) wsum[wave] = ss;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) tot += wsum[w];
            const float inv = rsqrtf(tot / (float)K + eps);
            for (int v = tid; v < K / 8; v += THREADS) {
                h8 a = *(h8*)(smem + (v >> 3) * XPITCH + (v & 7) * 16);
                const h8 g = *(const h8*)(norm_w + v * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (_Float16)((float)a[j] * inv) * g[j];
                *(h8*)(smem + (v >> 3) * XPITCH + (v & 7) * 16) = a;
            }
        }
    }
    __syncthreads();

    float acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;

    // byte offset of this lane's 16 activations inside a chunk of the LDS copy
    const int xoff = role * 32;

    for (int it = 0; it < n_it; ++it) {
        Tile nxt = {};
        if (it + 1 < n_it) nxt = load_tile(it + 1);   // next tile in flight during the math
        const int chunk = (it * TEAMS + team) * 16 + b;
        if (chunk < NC) {
            u32x4 xa[MB], xb[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const char* xk = smem + m * xrow + chunk * XPITCH + xoff;
                xa[m] = *(const u32x4*)xk;
                xb[m] = *(const u32x4*)(xk + 16);
            }
            uint32_t o[8];
            if (four) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t ca = cur.a[j], cb = cur.b[j];
                    mxq_deq4x8(ca, s4[j], z4[j], o);
                    mxq_deq4x8(cb, s4[j], z4[j], o + 4);
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m][j] = dot8(o + 4, xb[m], dot8(o, xa[m], acc[m][j]));
                }
            } else {
                const uint32_t qs_u = cur.qq[0], qz_u = cur.qq[1];
                const float qs = __uint_as_float(qs_u), qz = __uint_as_float(qz_u);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t scw = (cur.sc[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                    const uint32_t cw = cur.a[j], zw = cur.b[j];
                    mxq_deq2x16(cw, mxq_scale(qs, qz, (scw >> (4 * role)) & 15u), __uint_as_float(zw), o);
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[m][j] = dot8(o + 4, xb[m], dot8(o, xa[m], acc[m][j]));
                }
            }
        }
        cur = nxt;
    }

    // reduce over the 16 chunk lanes of the wave (lane bits 2..5), then over waves through LDS
    float* red = (float*)(smem + (size_t)MB * xrow);   // [WAVES][MB][16]; nobody reads wsum (same place) any more
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[m][j];
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (b == 0) red[(wave * MB + m) * 16 + 4 * q + j] = v;
        }
    __syncthreads();
    if (tid < MB * 16) {
        const int m = tid >> 4, rr = tid & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += red[(w * MB + m) * 16 + rr];
        if (m < M) {
            _Float16 h = (_Float16)v;
            if (residual) h = __builtin_bit_cast(_Float16, residual[(int64_t)m * N + rb * 16 + rr]) + h;
            y[(int64_t)m * N + rb * 16 + rr] = __builtin_bit_cast(uint16_t, h);
        }
    }
}

template <int MB, int TEAMS, int PRO, int LAYOUT = MXQ_LAYOUT_MIXED>
int launch_t(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
             const void* norm_w, float eps, const void* residual, hipStream_t stream) {
    const size_t smem = (size_t)MB * (K / 64) * XPITCH + (size_t)(TEAMS * 4) * MB * 16 * 4;
    if (smem > 160 * 1024) return -1;   // MXQ_E_SHAPE: the activations do not fit the LDS copy
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mxq_gemv2_f16_kernel<MB, TEAMS, PRO, LAYOUT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    mxq_gemv2_f16_kernel<MB, TEAMS, PRO, LAYOUT><<<N / 16, TEAMS * 256, smem, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K,
        (const uint16_t*)norm_w, eps, (const uint16_t*)residual);
    return (int)hipGetLastError();
}

// few row blocks (N/16 <= 384): 4 teams = 16 waves per workgroup, one workgroup per CU; more: 2 teams, so that two
// workgroups share a CU and one's loads overlap the other's arithmetic
template <int MB, int PRO, int LAYOUT>
int launch_n(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, const void* norm_w,
             float eps, const void* residual, hipStream_t stream, int teams) {
    if (teams == 0) teams = N / 16 <= 384 ? 4 : 2;
    if (MB == 8 && teams == 4) teams = 2;   // 8 tokens x 4 rows of accumulators do not fit 128 VGPRs (16 waves per CU)
    if constexpr (MB < 8) {
        if (teams == 4) return launch_t<MB, 4, PRO, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream);
    }
    if (teams == 2) return launch_t<MB, 2, PRO, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream);
    if (teams == 1) return launch_t<MB, 1, PRO, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream);
    return -1;
}

template <int LAYOUT>
int launch_m(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, hipStream_t stream,
             int teams) {
    if (M == 1) return launch_n<1, 0, LAYOUT>(x, qweight, rowmeta, y, M, N, K, nullptr, 0.f, nullptr, stream, teams);
    if (M == 2) return launch_n<2, 0, LAYOUT>(x, qweight, rowmeta, y, M, N, K, nullptr, 0.f, nullptr, stream, teams);
    if (M <= 4) return launch_n<4, 0, LAYOUT>(x, qweight, rowmeta, y, M, N, K, nullptr, 0.f, nullptr, stream, teams);
    if (M <= 8) return launch_n<8, 0, LAYOUT>(x, qweight, rowmeta, y, M, N, K, nullptr, 0.f, nullptr, stream, teams);
    return -1;
}

}   // namespace

int mxq_launch_gemv2_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                         int layout, int teams, hipStream_t stream) {
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launch_m<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, stream, teams);
        case MXQ_LAYOUT_W2G16: return launch_m<MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, stream, teams);
        case MXQ_LAYOUT_W4ROW: return launch_m<MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, stream, teams);
    }
    return -1;
}

int mxq_launch_gemv2_fused_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K,
                               int prologue, const void* norm_w, float eps, const void* residual, int teams,
                               hipStream_t stream) {
    switch (prologue) {
        case 0: return launch_n<1, 0, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, 1, N, K, norm_w, eps, residual, stream, teams);
        case 1: return launch_n<1, 1, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, 1, N, K, norm_w, eps, residual, stream, teams);
        case 2: return launch_n<1, 2, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, 1, N, K, norm_w, eps, residual, stream, teams);
    }
    return -1;
}

#ifdef MXQ_PROFILING
// A/B entry for tools/ (correct results): explicit team count
extern "C" int mxq_prof_gemv2_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                  int layout, int teams, void* stream) {
    return mxq_launch_gemv2_f16(x, qweight, rowmeta, y, M, N, K, layout, teams, (hipStream_t)stream);
}
#endif
