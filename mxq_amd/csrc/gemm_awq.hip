// W4 x A16 group-wise GEMM on the OPERANDS of the reference's gemm_forward_cuda
// (mxq_quant/cuda_kernel/csrc/quantization/gemm_cuda.h:3-4; launcher gemm_cuda_gen.cu:424-478; kernels :28-218, :221-416;
// nibble order dequantize.cuh:15-78):
//   in_feats  x      [M, IC]      fp16
//   kernel    B      [IC, OC/8]   int32: 8 four-bit codes of 8 consecutive output channels at ONE input channel
//   scales    S      [IC/G, OC]   fp16
//   zeros     Z      [IC/G, OC/8] int32: integer zero-points, packed like the codes
//   y[m, n] = sum_k x[m, k] * fp16(fp16(q[k, n] - z[k/G, n]) * s[k/G, n]),  fp32 accumulation.
// A 32-bit word holds channels e = 0..7 in nibbles (0, 4, 1, 5, 2, 6, 3, 7): dequantize_s4_to_fp16x2 returns the halves
// (n0, n4 | n1, n5 | n2, n6 | n3, n7) as elements 0..7 (dequantize.cuh:35-51).  The reference subtracts the zero and
// multiplies by the scale in fp16 (sub.f16x2, fma.rn.f16x2 with a zero addend, gemm_cuda_gen.cu:134-141): q - z is an
// exact integer, so the weight is the product rounded ONCE to fp16 -- reproduced here with the same packed fp16 operations.
//
// The weight is K-major (a word = 8 channels at one k), the MFMA operand wants 8 consecutive k of one channel: a thread
// takes the 4 words of (channel octet c, 4 consecutive k), dequantises the 8 x 4 block in registers and writes 8 rows of
// 8 bytes into the same swizzled [channel][64 k] fp16 tile the mixed-layout kernel uses (csrc/gemm.hip) -- the transpose
// costs nothing beyond the register shuffle.  Tile 128 tokens x 128 channels x 64 k, 4 waves of 64 x 64, D^T = W . x^T
// with v_mfma_f32_16x16x32_f16, x by LDS-DMA, two LDS stages.
// split_k (the launcher's split_k_iters, gemm_cuda_gen.cu:429-436): slice z of S takes the K-steps t = z, z + S, ... and
// writes its own fp32 partial [M, OC] (the reference: fp16 partials, summed by torch afterwards); S = 1 writes fp16 y.
#include <hip/hip_runtime.h>

#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;   // 64 KiB -> 2 workgroups per CU

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
// (channel e of a word sits in nibble (0, 4, 1, 5, 2, 6, 3, 7)[e], dequantize.cuh:35-51: the even nibbles' pair masks give
//  channels (0, 1) and (4, 5), the odd nibbles' (2, 3) and (6, 7))

struct WStage {
    uint32_t q[4];   // code words of (octet c, k = 4 kq .. 4 kq + 3)
    uint32_t z;      // zero-point word of the k range's group
    uint4 s;         // 8 fp16 scales of the octet
};

template <bool PARTIAL>
__global__ __launch_bounds__(256, 2) void mxq_gemm_awq_f16_kernel(const uint16_t* __restrict__ x,
                                                                  const uint32_t* __restrict__ kernel,
                                                                  const uint16_t* __restrict__ scales,
                                                                  const uint32_t* __restrict__ zeros, void* __restrict__ yv,
                                                                  int M, int IC, int OC, int G, int tiles_m, int tiles_n,
                                                                  int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = IC / BK, z = blockIdx.y;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware tile order (speed only)
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int OC8 = OC / 8;

    const uint16_t* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_src[i] = x + (int64_t)gm * IC + (((lane & 7) ^ (row & 7)) << 3);
    }
    auto issue_a = [&](int t, int buf) {
        char* dst = smem + buf * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + t * BK),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };

    // thread -> (channel octet c = tid & 15, k quad kq = tid >> 4): 16 consecutive lanes read 64 contiguous bytes of a k row
    const int c = tid & 15, kq = tid >> 4;
    int oc8 = n0 / 8 + c;
    oc8 = oc8 < OC8 ? oc8 : OC8 - 1;               // (a tile past the last channel re-reads the last octet; never stored)
    auto load_w = [&](int t, WStage& st) {
        const int k = t * BK + kq * 4;
        const int64_t g = k / G;
#pragma unroll
        for (int i = 0; i < 4; ++i) st.q[i] = kernel[(int64_t)(k + i) * OC8 + oc8];
        st.z = zeros[g * OC8 + oc8];
        st.s = *(const uint4*)(scales + g * OC + (int64_t)oc8 * 8);
    };
    // Dequant = the reference's own method (dequantize.cuh:15-78 + gemm_cuda_gen.cu:134-141) in packed fp16: a masked word OR-ed
    // into 0x6400 reads (1024 + q) [or 1024 + 16 q for the odd nibbles], and
    //     (1024 + q) - (1024 + z)            = q - z        exact
    //     (1024 + 16 q) / 16 - (64 + z)      = q - z        exact (one v_pk_fma_f16)
    // then ONE rounding in the multiply by the scale -- 13 packed ops per word of 8 weights.  The packed results hold
    // channels (2 j, 2 j + 1) at one k; two v_perm_b32 per channel gather its four k's.
    auto write_w = [&](const WStage& st, int buf) {
        char* base = smem + buf * STAGE_BYTES + A_BYTES;
        constexpr uint32_t LO = 0x000f000fu, HI = 0x00f000f0u, MAGIC = 0x64006400u;
        const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
        auto pairs = [&](uint32_t w, h2 (&o)[4]) {    // o[j] = (1024 + n) or (1024 + 16 n) of channels (2 j, 2 j + 1)
            const uint32_t t = w >> 8;
            o[0] = __builtin_bit_cast(h2, (w & LO) | MAGIC);
            o[1] = __builtin_bit_cast(h2, (w & HI) | MAGIC);
            o[2] = __builtin_bit_cast(h2, (t & LO) | MAGIC);
            o[3] = __builtin_bit_cast(h2, (t & HI) | MAGIC);
        };
        h2 zc[4];                                     // 1024 + z (even pairs) / 64 + z (odd pairs)
        pairs(st.z, zc);
        zc[1] = __builtin_elementwise_fma(zc[1], sixteenth, (h2){(_Float16)0.f, (_Float16)0.f});   // 64 + z, exact
        zc[3] = __builtin_elementwise_fma(zc[3], sixteenth, (h2){(_Float16)0.f, (_Float16)0.f});
        const h2 sp[4] = {__builtin_bit_cast(h2, st.s.x), __builtin_bit_cast(h2, st.s.y), __builtin_bit_cast(h2, st.s.z),
                          __builtin_bit_cast(h2, st.s.w)};
        uint32_t r[4][4];                              // r[i][j]: weights of channels (2 j, 2 j + 1) at k = 4 kq + i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h2 q[4];
            pairs(st.q[i], q);
            r[i][0] = __builtin_bit_cast(uint32_t, (q[0] - zc[0]) * sp[0]);
            r[i][1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(q[1], sixteenth, -zc[1]) * sp[1]);
            r[i][2] = __builtin_bit_cast(uint32_t, (q[2] - zc[2]) * sp[2]);
            r[i][3] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(q[3], sixteenth, -zc[3]) * sp[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int odd = 0; odd < 2; ++odd) {
                const uint32_t sel = odd ? 0x07060302u : 0x05040100u;      // the high / low halves of (second, first)
                uint2 w;
                w.x = __builtin_amdgcn_perm(r[1][j], r[0][j], sel);
                w.y = __builtin_amdgcn_perm(r[3][j], r[2][j], sel);
                // row = channel 8 c + 2 j + odd of the tile, k offset 4 kq: 16-byte slot kq >> 1, its half kq & 1
                *(uint2*)(base + swz(c * 8 + 2 * j + odd, kq >> 1) + (kq & 1) * 8) = w;
            }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const char* a_base = smem + buf * STAGE_BYTES;
        const char* b_base = a_base + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *(const half8*)(b_base + swz(wn * 64 + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = *(const half8*)(a_base + swz(wm * 64 + j * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    };

    // this slice's K-steps: t = z, z + S, ...  (a slice without a step writes zeros)
    WStage st;
    int t = z;
    if (t < NT) {
        issue_a(t, 0);
        load_w(t, st);
        write_w(st, 0);
    }
    __syncthreads();   // (also drains the LDS-DMA)
    for (int it = 0; t < NT; t += S, ++it) {
        const int cur = it & 1;
        const bool more = t + S < NT;
        if (more) {
            issue_a(t + S, cur ^ 1);
            load_w(t + S, st);
        }
        compute(cur);
        if (more) write_w(st, cur ^ 1);
        __syncthreads();
    }

    // lane holds y[m = .. + fr][n = .. + 4 fq + 0..3]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fq * 4;
            if (n >= OC) continue;   // OC % 8 == 0 and n % 4 == 0: a 4-wide store never straddles the edge
            if constexpr (PARTIAL) {
                *(f32x4*)((float*)yv + ((int64_t)z * M + m) * OC + n) = acc[i][j];
            } else {
                half4 h = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2], (_Float16)acc[i][j][3]};
                *(half4*)((uint16_t*)yv + (int64_t)m * OC + n) = h;
            }
        }
    }
}

}   // namespace

// split_k == 1: y = fp16 [M, OC].  split_k > 1: y = fp32 [split_k, M, OC] partial sums (the caller adds the slices).
int mxq_launch_gemm_awq_f16(const void* x, const void* kernel, const void* scales, const void* zeros, void* y, int M, int IC,
                            int OC, int G, int split_k, hipStream_t stream) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (OC + BN - 1) / BN;
    hipError_t e = split_k > 1 ? mxq_set_dyn_lds_once<&mxq_gemm_awq_f16_kernel<true>>(SMEM_BYTES)
                               : mxq_set_dyn_lds_once<&mxq_gemm_awq_f16_kernel<false>>(SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const dim3 grid(tiles_m * tiles_n, split_k > 1 ? split_k : 1);
    if (split_k > 1)
        mxq_gemm_awq_f16_kernel<true><<<grid, 256, SMEM_BYTES, stream>>>((const uint16_t*)x, (const uint32_t*)kernel,
                                                                         (const uint16_t*)scales, (const uint32_t*)zeros, y, M,
                                                                         IC, OC, G, tiles_m, tiles_n, split_k);
    else
        mxq_gemm_awq_f16_kernel<false><<<grid, 256, SMEM_BYTES, stream>>>((const uint16_t*)x, (const uint32_t*)kernel,
                                                                          (const uint16_t*)scales, (const uint32_t*)zeros, y, M,
                                                                          IC, OC, G, tiles_m, tiles_n, 1);
    return (int)hipGetLastError();
}
