// W2/4 x A16 dequant-GEMM for gfx950:  y[M, N] = x[M, K] . fp16(W')[N, K]^T, fp32 accumulate.
//
// Native counterpart of the reference's (never built) AWQ tensor-core GEMM
// mxq_quant/cuda_kernel/csrc/quantization/gemm_cuda_gen.cu:28-218 and of the implicit
// nn.Linear on fake-quant weights (mxq_quant/main.py:85); the arithmetic contract is
// x16 . fp16(scale*(q-zero))^T of lib/quantizer.py:19-20 + mxqgpt.py:448.
//
// Structure (v1, "dequant into LDS once per workgroup"):
//   * workgroup tile 128 (M) x 128 (N), K-step 64 = exactly one MXQ chunk; 4 waves, each
//     a 64 x 64 sub-tile = 4 x 4 blocks of v_mfma_f32_16x16x32_f16.
//   * x tile: global -> LDS directly (global_load_lds_dwordx4), XOR-swizzled through the
//     per-lane SOURCE address so ds_read_b128 of the fragments is conflict-poor.
//   * W tile: each thread reads the packed codes + group metadata of (row, two chunk
//     quarters) into registers one K-step ahead (<= 36 B per thread), dequantises them
//     with the LUT/v_perm helpers (mxq_dequant.h) and ds_write_b128's 64 B of fp16 into
//     the swizzled W tile.  Packed HBM traffic is ~4.4 bit/weight; the fp16 form exists
//     only in LDS.
//   * MFMA is issued as D^T = W . x^T (W rows are the "A" operand) so that each lane owns
//     4 consecutive output channels of one token -> 8-byte stores.
//   * two LDS buffers; loads for step t+1 are issued before the MFMAs of step t.
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;   // 16 KiB
constexpr int B_BYTES = BN * BK * 2;   // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;   // 64 KiB -> 2 workgroups per CU

// byte offset of 16-B slot `slot` (0..7) of row `row` in a [rows][64 halfs] tile
__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

// packed operands of one thread for one K-step (qp 0: groups 0,1; qp 1: group 2 + 4-bit arm)
struct BStage {
    uint32_t c0, c1;     // qp0: 2-bit words of g0, g1        qp1: 2-bit word of g2, 4-bit word h0
    uint32_t z0, z1;     // qp0: zero-points of g0, g1        qp1: zero-point of g2, 4-bit word h1
    uint32_t sc;         // 3 x 4-bit scale codes of the row
    uint2 qa, qb;        // qp0: (qs,qz) of g0 and g1         qp1: (qs,qz) of g2, unused
};

__global__ __launch_bounds__(256, 2) void mxq_gemm_f16_kernel(const uint16_t* __restrict__ x,
                                                              const uint32_t* __restrict__ qweight,
                                                              const float4* __restrict__ rowmeta,
                                                              uint16_t* __restrict__ y, int M, int N, int K,
                                                              int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = K / BK;

    // XCD-aware tile order: blocks b, b+8, ... share an L2; give each XCD a contiguous run
    // of tiles (bijective remap, cdna guide T1).  Placement only affects speed.
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- x (activation) staging: 4 global_load_lds per wave per K-step ----------------
    // wave-instruction i of wave w fills LDS rows 8*(4w+i) .. +7 (1 KiB, lane-linear);
    // lane -> row 8*(4w+i) + lane/8; LDS slot lane%8 receives global slot (lane%8)^(row&7).
    const uint16_t* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        a_src[i] = x + (int64_t)gm * K + (((lane & 7) ^ (row & 7)) << 3);
    }

    // ---- W staging: thread -> (row = tid & 127, quarter pair qp = tid >> 7) ------------
    const int b_row = tid & 127, qp = tid >> 7;   // qp is wave-uniform
    int gn = n0 + b_row;
    gn = gn < N ? gn : N - 1;
    const int b_r = gn & 15;
    const uint32_t* b_tile0 = qweight + (int64_t)(gn >> 4) * NT * MXQ_BLK_DW;
    float s4 = 0.f, z4 = 0.f;
    if (qp == 1) {
        const float4 m = rowmeta[gn];
        s4 = mxq_scale(m.z, m.w, (uint32_t)m.y);
        z4 = m.x;
    }

    auto issue_a = [&](int t, int buf) {
        char* dst = smem + buf * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + t * BK),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
    };
    auto load_b = [&](int t, BStage& st) {
        const uint32_t* tile = b_tile0 + (int64_t)t * MXQ_BLK_DW;
        st.sc = ((const uint16_t*)tile)[mxq_sc_u16(b_r)];
        if (qp == 0) {
            st.c0 = tile[mxq_c2(0, b_r)];
            st.c1 = tile[mxq_c2(1, b_r)];
            st.z0 = tile[mxq_z2(0, b_r)];
            st.z1 = tile[mxq_z2(1, b_r)];
            st.qa = *(const uint2*)(tile + mxq_qq(0));
            st.qb = *(const uint2*)(tile + mxq_qq(1));
        } else {
            st.c0 = tile[mxq_c2(2, b_r)];
            st.c1 = tile[mxq_c4(0, b_r)];
            st.z0 = tile[mxq_z2(2, b_r)];
            st.z1 = tile[mxq_c4(1, b_r)];
            st.qa = *(const uint2*)(tile + mxq_qq(2));
            st.qb = make_uint2(0u, 0u);
        }
    };
    auto write_b = [&](const BStage& st, int buf) {
        char* base = smem + buf * STAGE_BYTES + A_BYTES;
        uint32_t o[16];
        if (qp == 0) {
            mxq_deq2x16(st.c0, mxq_scale(__uint_as_float(st.qa.x), __uint_as_float(st.qa.y), st.sc & 15u),
                        __uint_as_float(st.z0), o);
            mxq_deq2x16(st.c1, mxq_scale(__uint_as_float(st.qb.x), __uint_as_float(st.qb.y), (st.sc >> 4) & 15u),
                        __uint_as_float(st.z1), o + 8);
        } else {
            mxq_deq2x16(st.c0, mxq_scale(__uint_as_float(st.qa.x), __uint_as_float(st.qa.y), (st.sc >> 8) & 15u),
                        __uint_as_float(st.z0), o);
            mxq_deq4x8(st.c1, s4, z4, o + 8);
            mxq_deq4x8(st.z1, s4, z4, o + 12);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
            *(uint4*)(base + swz(b_row, qp * 4 + s)) = make_uint4(o[4 * s], o[4 * s + 1], o[4 * s + 2], o[4 * s + 3]);
    };

    // ---- MFMA tiling: wave (wm, wn) owns x rows [64wm, +64) and W rows [64wn, +64) ----
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][4];   // [W block i][x block j]
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

    BStage st;
    issue_a(0, 0);
    load_b(0, st);
    write_b(st, 0);
    __syncthreads();   // also drains the LDS-DMA (vmcnt(0)) of issue_a

    for (int t = 0; t < NT; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < NT);
        if (more) {
            issue_a(t + 1, cur ^ 1);
            load_b(t + 1, st);
        }
        compute(cur);
        if (more) write_b(st, cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds y[m = .. + fr][n = .. + 4*fq + 0..3] ---------------------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fq * 4;
            if (n >= N) continue;   // N % 16 == 0, so a 4-wide store never straddles the edge
            half4 h = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2],
                       (_Float16)acc[i][j][3]};
            *(half4*)(y + (int64_t)m * N + n) = h;
        }
    }
}

}   // namespace

int mxq_launch_gemm_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                        hipStream_t stream) {
    // tokens <= 128 fit one 128-row tile; beyond that the 256 x 128 wave-specialised kernel wins
    // (without a workspace it runs whole tiles only: no stream-K tail)
    if (M > 128) return mxq_launch_gemm8_f16(x, qweight, rowmeta, y, M, N, K, nullptr, 0, 0, stream);
    return mxq_launch_gemm1_f16(x, qweight, rowmeta, y, M, N, K, stream);
}

int mxq_launch_gemm1_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                         hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)mxq_gemm_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    mxq_gemm_f16_kernel<<<tiles_m * tiles_n, 256, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m,
        tiles_n);
    return (int)hipGetLastError();
}
