// W2/4 x A16 dequant-GEMM, pipelined version (v2) -- the prefill workhorse.
//
//   y[M, N] = x[M, K] . fp16(W')[N, K]^T      (fp16 in, fp32 accumulate, fp16 out)
//
// Same arithmetic contract as gemm.hip (x16 . fp16(scale*(q-zero))^T; reference
// mxq_quant/lib/quantizer.py:19-20 + mxqgpt.py:448; structural precedent
// cuda_kernel/csrc/quantization/gemm_cuda_gen.cu:28-218), different machine mapping:
//
//   * workgroup tile 256 tokens (M) x 128 output channels (N), K-step 64 = one MXQ chunk;
//     8 waves (2 per SIMD), each a 64 x 64 sub-tile = 4 x 4 v_mfma_f32_16x16x32_f16.
//     256 x 128 gives exactly 256 workgroups (one per CU) for [2048 x 4096] outputs.
//   * EVERY global->LDS byte moves by LDS-DMA (global_load_lds_dwordx4), so the only VMEM
//     counter traffic in the loop is counted: 4 DMAs/wave/step for the x tile (XOR-swizzled
//     through the source address) + 1 DMA/wave/step that copies one whole 576-B packed block
//     (16 rows x 64 channels: codes, zeros, scale codes, (qs,qz)) verbatim.
//   * the x tile is prefetched 2 K-steps ahead (3-slot ring), the packed W blocks 3 ahead
//     (4-slot ring); each step ends with a COUNTED s_waitcnt vmcnt(5) + raw s_barrier, so
//     the newest stage stays in flight across the barrier (cdna guide T3/T4).
//   * dequant is done ONCE per workgroup per K-step: every thread turns 16 packed weights
//     (read from the LDS copy of the block) into fp16 with the LUT / v_perm_b32 helpers and
//     writes 32 B into a double-buffered, XOR-swizzled W16 tile; the MFMAs of step t overlap
//     the dequant of step t+1.  The fp16 weight never exists outside LDS.
//   * D^T = W . x^T: a lane owns 4 consecutive output channels of one token (8-B stores).
//
// hipcc note (ROCm 7.2): SIInsertWaitcnts puts `s_waitcnt vmcnt(0)` in front of any LDS access
// that TBAA says may alias an in-flight LDS-DMA.  Struct-typed accesses (`uint2`, `uint4` =
// HIP_vector_type) do; scalar and ext_vector_type accesses do not.  Every LDS access in the
// loop therefore uses uint32_t / ext-vector types, and tests/test_build_asm.py asserts that
// the only vmcnt waits inside the loop are the hand-placed counted ones.
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
constexpr int A_STAGE = BM * BK * 2;            // 32 KiB
constexpr int A_SLOTS = 3;
constexpr int BP_STAGE = (BN / 16) * MXQ_BLK_BYTES;   // 8 blocks = 4608 B
constexpr int BP_SLOTS = 4;
constexpr int W_STAGE = BN * BK * 2;            // 16 KiB
constexpr int OFF_A = 0;
constexpr int OFF_BP = OFF_A + A_SLOTS * A_STAGE;
constexpr int OFF_W = OFF_BP + BP_SLOTS * BP_STAGE;
constexpr int SMEM_BYTES = OFF_W + 2 * W_STAGE;   // 149,504 B of the CU's 160 KiB

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ABL: ablation bits for profiling builds only (wrong results): 1 = no x DMA in the loop,
// 2 = no MFMA, 4 = no dequant, 8 = no fragment reads.  ABL = 0 is the product kernel.
template <int ABL>
__global__ __launch_bounds__(THREADS, 2) void mxq_gemm2_f16_kernel(const uint16_t* __restrict__ x,
                                                                  const uint32_t* __restrict__ qweight,
                                                                  const float4* __restrict__ rowmeta,
                                                                  uint16_t* __restrict__ y, int M, int N, int K,
                                                                  int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = K / BK;

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (guide T1): speed only
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- DMA sources -------------------------------------------------------------------
    // x: DMA i of wave w fills rows 8*(4w+i) .. +7 of the A slot; LDS slot lane%8 of a row
    // receives global 16-B slot (lane%8) ^ (row&7).
    const uint16_t* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_src[i] = x + (int64_t)gm * K + (((lane & 7) ^ (row & 7)) << 3);
    }
    // packed W: wave w copies the 576-B block of 16-row block (n0/16 + w), lanes 0..35.
    int rb = (n0 >> 4) + wave;
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

    // ---- dequant role: thread -> (W row = 64*(wave&1) + lane, chunk quarter = wave>>1) ---
    const int d_row = (wave & 1) * 64 + lane, d_q = wave >> 1;   // d_q wave-uniform
    const int d_blk = d_row >> 4, d_r = d_row & 15;
    float s4 = 0.f, z4 = 0.f;
    if (d_q == 3) {
        int gn = n0 + d_row;
        gn = gn < N ? gn : N - 1;
        const float4 m = rowmeta[gn];
        s4 = mxq_scale(m.z, m.w, (uint32_t)m.y);
        z4 = m.x;
    }
    auto dequant = [&](int t) {   // packed block copy of step t -> W16[t & 1]
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

    // ---- MFMA role: wave (wm, wn) owns tokens [64wm, +64) x channels [64wn, +64) ----------
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][4];   // [channel block i][token block j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int t) {
        const char* a_base = smem + OFF_A + (t % A_SLOTS) * A_STAGE;
        const char* w_base = smem + OFF_W + (t & 1) * W_STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (ABL & 8) wf[i] = (half8){1, 2, 3, 4, 5, 6, 7, 8};
                else wf[i] = *(const half8*)(w_base + swz(wn * 64 + i * 16 + fr, kk * 4 + fq));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (ABL & 8) xf[j] = (half8){1, 2, 3, 4, 5, 6, 7, 8};
                else xf[j] = *(const half8*)(a_base + swz(wm * 64 + j * 16 + fr, kk * 4 + fq));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (ABL & 2) asm volatile("" ::"v"(wf[i]), "v"(xf[j]));
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    // ---- prologue: fill the rings, dequantise step 0 ---------------------------------------
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

    // ---- main loop ---------------------------------------------------------------------------
    for (int t = 0; t < NT; ++t) {
        const bool steady = (t + 3 < NT);
        if constexpr (!(ABL & 1)) {
            if (t + 2 < NT) issue_a(t + 2);
        }
        if (t + 3 < NT) issue_bp(t + 3);
        compute(t);
        if constexpr (!(ABL & 4)) {
            if (t + 1 < NT) dequant(t + 1);
        }
        if (steady) {
            if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            else
            asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");   // this step's 5 DMAs stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue ------------------------------------------------------------------------------
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

template <int ABL>
static int launch2(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)mxq_gemm2_f16_kernel<ABL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    mxq_gemm2_f16_kernel<ABL><<<tiles_m * tiles_n, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n);
    return (int)hipGetLastError();
}

int mxq_launch_gemm2_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                         hipStream_t stream) {
    return launch2<0>(x, qweight, rowmeta, y, M, N, K, stream);
}

// profiling-only ablation builds (outputs are wrong by construction)
int mxq_launch_gemm2_ablate_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                int abl, hipStream_t stream) {
    switch (abl) {
        case 1: return launch2<1>(x, qweight, rowmeta, y, M, N, K, stream);
        case 2: return launch2<2>(x, qweight, rowmeta, y, M, N, K, stream);
        case 4: return launch2<4>(x, qweight, rowmeta, y, M, N, K, stream);
        case 8: return launch2<8>(x, qweight, rowmeta, y, M, N, K, stream);
        case 6: return launch2<6>(x, qweight, rowmeta, y, M, N, K, stream);
        case 14: return launch2<14>(x, qweight, rowmeta, y, M, N, K, stream);
        case 13: return launch2<13>(x, qweight, rowmeta, y, M, N, K, stream);
    }
    return (int)hipErrorInvalidValue;
}
