// W2/4 x A16 dequant-GEMM, "ping-pong" version (v3).
//
//   y[M, N] = x[M, K] . fp16(W')[N, K]^T      (fp16 in, fp32 accumulate, fp16 out)
//
// Arithmetic contract as gemm.hip / gemm2.hip (reference mxq_quant/lib/quantizer.py:19-20,
// mxqgpt.py:448; structural precedent cuda_kernel/csrc/quantization/gemm_cuda_gen.cu:28-218).
//
// gemm2's waves all run the same phases at the same time, so each SIMD's matrix pipe idles
// while both of its waves read fragments / dequantise / sit in the barrier.  Here the 8 waves
// of the 256 x 128 workgroup tile are split into two groups (waves 0-3 = X, waves 4-7 = Y; one
// wave of each per SIMD) that alternate two kinds of half-step ("slot"), half a K-step apart:
//
//     slot:      |   2s        |   2s+1      |   2s+2      |
//     X waves:   |  MFMA(s)    |  other(s+1) |  MFMA(s+1)  |       32 MFMAs from registers
//     Y waves:   |  other(s)   |  MFMA(s)    |  other(s+1) |
//
//   other(s) = issue this wave's LDS-DMAs for x tile s+2 and packed-W block s+3
//              + read the 16 MFMA fragments of step s into registers (ds_read_b128)
//              + dequantise this group's 64 rows of W16(s+1) from the LDS copy of block s+1
//              + s_waitcnt vmcnt(5) lgkmcnt(0)          (this slot's 5 DMAs stay in flight)
//   every slot ends with one raw s_barrier; an MFMA slot touches no memory at all.
//
// Rings: x tile 3 slots (32 KiB each), packed W 4 slots (4.5 KiB), W16 2 slots (16 KiB):
// 146 KiB of LDS, one workgroup per CU.  Hazard bookkeeping is in DESIGN.md section 4.
// LDS accesses use scalar / ext_vector types only (see the hipcc note in gemm2.hip).
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = 64, THREADS = 512;
constexpr int A_STAGE = BM * BK * 2, A_SLOTS = 3;
constexpr int BP_STAGE = (BN / 16) * MXQ_BLK_BYTES, BP_SLOTS = 4;
constexpr int W_STAGE = BN * BK * 2;
constexpr int OFF_A = 0;
constexpr int OFF_BP = OFF_A + A_SLOTS * A_STAGE;
constexpr int OFF_W = OFF_BP + BP_SLOTS * BP_STAGE;
constexpr int SMEM_BYTES = OFF_W + 2 * W_STAGE;   // 149,504 B

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(THREADS, 2) void mxq_gemm3_f16_kernel(const uint16_t* __restrict__ x,
                                                                  const uint32_t* __restrict__ qweight,
                                                                  const float4* __restrict__ rowmeta,
                                                                  uint16_t* __restrict__ y, int M, int N, int K,
                                                                  int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;   // 0 = X, 1 = Y
    const int NT = K / BK;

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (guide T1): speed only
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- DMA sources (as gemm2) -----------------------------------------------------------
    const uint16_t* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_src[i] = x + (int64_t)gm * K + (((lane & 7) ^ (row & 7)) << 3);
    }
    int rb = (n0 >> 4) + wave;   // wave w copies the packed block of rows 16w..16w+15 (its own group's rows)
    rb = rb < (N >> 4) ? rb : (N >> 4) - 1;
    const char* bp_src = (const char*)(qweight + (int64_t)rb * NT * MXQ_BLK_DW) + lane * 16;

    auto issue_a = [&](int t) {
        char* dst = smem + OFF_A + (t % A_SLOTS) * A_STAGE + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src[i] + t * BK, dst + i * 1024);
    };
    auto issue_bp = [&](int t) {
        char* dst = smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE + wave * MXQ_BLK_BYTES;
        if (lane < 36) glds16(bp_src + (int64_t)t * MXQ_BLK_BYTES, dst);
    };

    // ---- dequant role: thread -> (W row = 64*grp + lane, chunk quarter = wave & 3) ----------
    const int d_row = grp * 64 + lane, d_q = wave & 3;   // d_q wave-uniform
    const int d_blk = d_row >> 4, d_r = d_row & 15;
    float s4 = 0.f, z4 = 0.f;
    if (d_q == 3) {
        int gn = n0 + d_row;
        gn = gn < N ? gn : N - 1;
        const float4 m = rowmeta[gn];
        s4 = mxq_scale(m.z, m.w, (uint32_t)m.y);
        z4 = m.x;
    }
    auto dequant = [&](int t) {   // this group's 64 rows of step t: packed LDS copy -> W16[t & 1]
        const uint32_t* blk = (const uint32_t*)(smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE + d_blk * MXQ_BLK_BYTES);
        uint32_t o[8];
        if (d_q < 3) {
            const uint32_t d = blk[mxq_c2(d_q, d_r)];
            const float z = __uint_as_float(blk[mxq_z2(d_q, d_r)]);
            const uint32_t scw = ((const uint16_t*)blk)[mxq_sc_u16(d_r)];
            const uint32_t qq_x = blk[mxq_qq(d_q)], qq_y = blk[mxq_qq(d_q) + 1];
            mxq_deq2x16(d, mxq_scale(__uint_as_float(qq_x), __uint_as_float(qq_y), (scw >> (4 * d_q)) & 15u), z, o);
        } else {
            mxq_deq4x8(blk[mxq_c4(0, d_r)], s4, z4, o);
            mxq_deq4x8(blk[mxq_c4(1, d_r)], s4, z4, o + 4);
        }
        char* wt = smem + OFF_W + (t & 1) * W_STAGE;
        *(u32x4*)(wt + swz(d_row, d_q * 2)) = (u32x4){o[0], o[1], o[2], o[3]};
        *(u32x4*)(wt + swz(d_row, d_q * 2 + 1)) = (u32x4){o[4], o[5], o[6], o[7]};
    };

    // ---- MFMA role: wave (wm, wn) owns tokens [64wm, +64) x channels [64wn, +64) -----------
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][4];   // [channel block i][token block j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    half8 wf[2][4], xf[2][4];   // fragments of one K-step, [kk][block]

    auto load_frags = [&](int t) {
        const char* a_base = smem + OFF_A + (t % A_SLOTS) * A_STAGE;
        const char* w_base = smem + OFF_W + (t & 1) * W_STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[kk][i] = *(const half8*)(w_base + swz(wn * 64 + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[kk][j] = *(const half8*)(a_base + swz(wm * 64 + j * 16 + fr, kk * 4 + fq));
        }
    };
    auto mfma_slot = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][i], xf[kk][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto other_slot = [&](int s) {   // s < NT
        if (s + 2 < NT) issue_a(s + 2);
        if (s + 3 < NT) issue_bp(s + 3);
        load_frags(s);
        if (s + 1 < NT) dequant(s + 1);
        if (s + 3 < NT) {
            asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");   // this slot's 5 DMAs stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    };

    // ---- prologue ---------------------------------------------------------------------------------
    issue_a(0);
    if (NT > 1) issue_a(1);
    issue_bp(0);
    if (NT > 1) issue_bp(1);
    if (NT > 2) issue_bp(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dequant(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 0) other_slot(0);
    __builtin_amdgcn_s_barrier();

    // ---- main loop: two slots per K-step -------------------------------------------------------------
    for (int s = 0; s < NT; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) {
            mfma_slot();
        } else {
            other_slot(s);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) {
            if (s + 1 < NT) other_slot(s + 1);
        } else {
            mfma_slot();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue -------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fq * 4;
            if (n >= N) continue;
            half4 h = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2],
                       (_Float16)acc[i][j][3]};
            *(half4*)(y + (int64_t)m * N + n) = h;
        }
    }
}

}   // namespace

int mxq_launch_gemm3_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                         hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)mxq_gemm3_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    mxq_gemm3_f16_kernel<<<tiles_m * tiles_n, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n);
    return (int)hipGetLastError();
}
