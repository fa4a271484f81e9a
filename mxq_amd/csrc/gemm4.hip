// W2/4 x A16 dequant-GEMM, wave-specialised version (v4).
//
//   y[M, N] = x[M, K] . fp16(W')[N, K]^T      (fp16 in, fp32 accumulate, fp16 out)
//
// Arithmetic contract as gemm.hip / gemm2.hip (reference mxq_quant/lib/quantizer.py:19-20,
// mxqgpt.py:448; structural precedent cuda_kernel/csrc/quantization/gemm_cuda_gen.cu:28-218).
//
// Profiling of gemm2 (round 1: profiles/r01_gemm2_*.txt) showed the matrix pipe busy only ~46 % of
// the time: every wave spends a third of each K-step issuing LDS-DMAs (~70 cycles each), running
// the dequant VALU chain and waiting for its LDS writes, and all waves do so at the same time.
// Here the 12 waves of a workgroup (3 per SIMD) have fixed roles:
//
//   * waves 0-7   CONSUMERS: 64 x 64 sub-tile each (4 x 4 v_mfma_f32_16x16x32_f16), nothing but
//                 ds_read_b128 + MFMA; fragments double-buffered in registers by half K-steps
//                 (while 16 MFMAs run, the other half's 8 fragments are read).
//   * waves 8-11  PRODUCERS: one per SIMD.  Each K-step they issue all LDS-DMAs (x tile t+2:
//                 32 x 1 KiB, packed W blocks t+3: 8 x 576 B) and dequantise chunk t+1 from the
//                 LDS copy of its packed blocks into the fp16 W16 tile (2 x 16 weights per
//                 thread, LUT / v_perm_b32).
//
// One raw s_barrier per K-step joins all 12 waves; producers end a step with a counted
// s_waitcnt vmcnt(10) so their newest DMAs stay in flight.  Tile 256 (M) x 128 (N), K-step 64,
// LDS: x ring 3 x 32 KiB, packed ring 3 x 4.5 KiB, W16 2 x 16 KiB.  Hazards as gemm2 (every
// wave passes the same barrier every step): DESIGN.md section 4.
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int N_CONS = 8, N_PROD = 4, THREADS = (N_CONS + N_PROD) * 64;
constexpr int A_STAGE = BM * BK * 2, A_SLOTS = 3;
constexpr int BP_BLK = MXQ_BLK_BYTES;            // 576 B: stride of 144 dwords keeps blocks on distinct banks
constexpr int BP_STAGE = (BN / 16) * BP_BLK, BP_SLOTS = 3;
constexpr int W_STAGE = BN * BK * 2;
constexpr int OFF_A = 0;
constexpr int OFF_BP = OFF_A + A_SLOTS * A_STAGE;
constexpr int OFF_W = (OFF_BP + BP_SLOTS * BP_STAGE + 255) / 256 * 256;
constexpr int SMEM_BYTES = OFF_W + 2 * W_STAGE;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// same XCD-aware tile order as gemm2 (speed only)
__device__ __forceinline__ void tile_of_block(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
    if ((tiles_m & 3) == 0 && (tiles_n & 1) == 0) {
        const int e = bid & 7, l = bid >> 3;
        const int rm = tiles_m >> 2, rn = tiles_n >> 1;
        const int full = rm * 16;
        const int p = l / full;
        const int j = l - p * full;
        const int left = rn - p * 16;
        const int pw = left < 16 ? left : 16;
        tm = (e & 3) * rm + j / pw;
        tn = (e >> 2) * rn + p * 16 + j % pw;
        return;
    }
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tm = lin % tiles_m;
    tn = lin / tiles_m;
}

// ------------------------------------------------------------------------------------------------
// consumer
// ------------------------------------------------------------------------------------------------
typedef half8 Frag4[4];

__device__ __forceinline__ void load_frags(const char* smem, int t, int kk, int wm, int wn, int fr, int fq, Frag4& wf,
                                           Frag4& xf) {
    const char* a_base = smem + OFF_A + (t % A_SLOTS) * A_STAGE;
    const char* w_base = smem + OFF_W + (t & 1) * W_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *(const half8*)(w_base + swz(wn * 64 + i * 16 + fr, kk * 4 + fq));
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = *(const half8*)(a_base + swz(wm * 64 + j * 16 + fr, kk * 4 + fq));
}

template <int I0, int I1, int ABL = 0>
__device__ __forceinline__ void mfma_rows(f32x4 (&acc)[4][4], const Frag4& wf, const Frag4& xf) {
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (ABL & 2) asm volatile("" ::"v"(wf[i]), "v"(xf[j]));
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
}

template <int ABL>
__device__ __forceinline__ void consumer(char* smem, int wave, int lane, int NT, uint16_t* __restrict__ y, int M, int N,
                                         int m0, int n0) {
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag4 wf0, xf0, wf1, xf1;

    __builtin_amdgcn_s_barrier();   // prologue barrier 1: x tiles 0,1 and packed blocks 0..2 landed
    __builtin_amdgcn_s_barrier();   // prologue barrier 2: W16(0) written

    // step 0: no previous half
    load_frags(smem, 0, 0, wm, wn, fr, fq, wf0, xf0);
    load_frags(smem, 0, 1, wm, wn, fr, fq, wf1, xf1);
    mfma_rows<0, 4, ABL>(acc, wf0, xf0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int t = 1; t < NT; ++t) {
        // (wf1, xf1) = fragments of (t-1, kk=1), waited for at the end of the previous step
        mfma_rows<0, 1, ABL>(acc, wf1, xf1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 8)) load_frags(smem, t, 0, wm, wn, fr, fq, wf0, xf0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows<1, 4, ABL>(acc, wf1, xf1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 8)) load_frags(smem, t, 1, wm, wn, fr, fq, wf1, xf1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows<0, 4, ABL>(acc, wf0, xf0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    mfma_rows<0, 4, ABL>(acc, wf1, xf1);   // (NT-1, kk=1)

    // Output through LDS (the x ring is idle after the last barrier) so that the tile leaves as full 128-B lines,
    // 16 B per lane, instead of 32-B pieces in 16 rows per instruction: see store_tile_staged in gemm6.hip.
    constexpr int ROW = 144;
    char* st = smem + OFF_A + wave * (64 * ROW);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            half4 h = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2],
                       (_Float16)acc[i][j][3]};
            *(half4*)(st + (j * 16 + fr) * ROW + (i * 16 + fq * 4) * 2) = h;
        }
    const int n = n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3);
        const u32x4 v = *(const u32x4*)(st + row * ROW + (lane & 7) * 16);
        const int m = m0 + wm * 64 + row;
        if (m < M && n < N) *(u32x4*)(y + (int64_t)m * N + n) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// producer
// ------------------------------------------------------------------------------------------------
struct Prod {
    char* smem;
    const uint16_t* a_src[8];
    const char* bp_src[2];
    int p, lane, NT;
    int d_row, d_qp, d_blk, d_r;
    float s4, z4;
};

__device__ __forceinline__ void issue_a(const Prod& c, int t) {
    // producer p fills rows 64p .. 64p+63 of the x slot: DMA i covers rows 64p + 8i .. +7
    char* dst = c.smem + OFF_A + (t % A_SLOTS) * A_STAGE + c.p * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(c.a_src[i] + t * BK, dst + i * 1024);
}
template <int LAYOUT>
__device__ __forceinline__ void issue_bp(const Prod& c, int t) {
    // producer p copies packed blocks 2p, 2p+1 (rows 32p .. 32p+31), 36 lanes each (32 for W4ROW);
    // the LDS stride stays 576 B for every layout (bank-conflict-free block spacing)
    constexpr int BYTES = LAYOUT == MXQ_LAYOUT_W4ROW ? 512 : MXQ_BLK_BYTES;
    char* dst = c.smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE + c.p * 2 * BP_BLK;
    if (c.lane < BYTES / 16) {
        glds16(c.bp_src[0] + (int64_t)t * BYTES, dst);
        glds16(c.bp_src[1] + (int64_t)t * BYTES, dst + BP_BLK);
    }
}

// thread -> (W row d_row, quarter pair d_qp): chunk t's 32 weights of that row -> W16[t & 1]
template <int LAYOUT>
__device__ __forceinline__ void dequant(const Prod& c, int t) {
    const uint32_t* blk = (const uint32_t*)(c.smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE + c.d_blk * BP_BLK);
    uint32_t o[16];
    if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            mxq_deq4x8(blk[mxq_w4_c4(c.d_qp * 2 + q, 0, c.d_r)], c.s4, c.z4, o + 8 * q);
            mxq_deq4x8(blk[mxq_w4_c4(c.d_qp * 2 + q, 1, c.d_r)], c.s4, c.z4, o + 8 * q + 4);
        }
    } else if constexpr (LAYOUT == MXQ_LAYOUT_W2G16) {
        const uint32_t scw = ((const uint16_t*)blk)[mxq_sc_u16(c.d_r)];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = c.d_qp * 2 + q;
            mxq_deq2x16(blk[mxq_w2_c2(g, c.d_r)],
                        mxq_scale(__uint_as_float(blk[mxq_qq(g)]), __uint_as_float(blk[mxq_qq(g) + 1]),
                                  (scw >> (4 * g)) & 15u),
                        __uint_as_float(blk[mxq_w2_z2(g, c.d_r)]), o + 8 * q);
        }
    } else {
    const uint32_t scw = ((const uint16_t*)blk)[mxq_sc_u16(c.d_r)];
    if (c.d_qp == 0) {
        mxq_deq2x16(blk[mxq_c2(0, c.d_r)],
                    mxq_scale(__uint_as_float(blk[mxq_qq(0)]), __uint_as_float(blk[mxq_qq(0) + 1]), scw & 15u),
                    __uint_as_float(blk[mxq_z2(0, c.d_r)]), o);
        mxq_deq2x16(blk[mxq_c2(1, c.d_r)],
                    mxq_scale(__uint_as_float(blk[mxq_qq(1)]), __uint_as_float(blk[mxq_qq(1) + 1]), (scw >> 4) & 15u),
                    __uint_as_float(blk[mxq_z2(1, c.d_r)]), o + 8);
    } else {
        mxq_deq2x16(blk[mxq_c2(2, c.d_r)],
                    mxq_scale(__uint_as_float(blk[mxq_qq(2)]), __uint_as_float(blk[mxq_qq(2) + 1]), (scw >> 8) & 15u),
                    __uint_as_float(blk[mxq_z2(2, c.d_r)]), o);
        mxq_deq4x8(blk[mxq_c4(0, c.d_r)], c.s4, c.z4, o + 8);
        mxq_deq4x8(blk[mxq_c4(1, c.d_r)], c.s4, c.z4, o + 12);
    }
    }
    char* wt = c.smem + OFF_W + (t & 1) * W_STAGE;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        *(u32x4*)(wt + swz(c.d_row, c.d_qp * 4 + s)) = (u32x4){o[4 * s], o[4 * s + 1], o[4 * s + 2], o[4 * s + 3]};
}

template <int ABL, int LAYOUT>
__device__ __forceinline__ void producer(const Prod& c) {
    // prologue: x tiles 0,1; packed blocks 0..2; W16(0)
    for (int t = 0; t < 2 && t < c.NT; ++t) issue_a(c, t);
    for (int t = 0; t < 3 && t < c.NT; ++t) issue_bp<LAYOUT>(c, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dequant<LAYOUT>(c, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int t = 0;
    for (; t + 3 < c.NT; ++t) {   // steady state: everything unconditional
        if constexpr (!(ABL & 1)) issue_a(c, t + 2);
        issue_bp<LAYOUT>(c, t + 3);
        if constexpr (!(ABL & 4)) dequant<LAYOUT>(c, t + 1);
        if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");   // this step's 10 DMAs stay in flight
        __builtin_amdgcn_s_barrier();
    }
    for (; t < c.NT; ++t) {
        if (t + 2 < c.NT) issue_a(c, t + 2);
        if (t + 3 < c.NT) issue_bp<LAYOUT>(c, t + 3);
        if (t + 1 < c.NT) dequant<LAYOUT>(c, t + 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

template <int ABL, int LAYOUT>
__global__ __launch_bounds__(THREADS) void mxq_gemm4_f16_kernel(const uint16_t* __restrict__ x,
                                                               const uint32_t* __restrict__ qweight,
                                                               const float4* __restrict__ rowmeta,
                                                               uint16_t* __restrict__ y, int M, int N, int K,
                                                               int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = K / BK;
    int tm, tn;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    if (wave < N_CONS) {
        consumer<ABL>(smem, wave, lane, NT, y, M, N, m0, n0);
        return;
    }
    Prod c;
    c.smem = smem;
    c.p = wave - N_CONS;
    c.lane = lane;
    c.NT = NT;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = c.p * 64 + i * 8 + (lane >> 3);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        c.a_src[i] = x + (int64_t)gm * K + (((lane & 7) ^ (row & 7)) << 3);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int rb = (n0 >> 4) + c.p * 2 + b;
        rb = rb < (N >> 4) ? rb : (N >> 4) - 1;
        c.bp_src[b] = (const char*)(qweight + (int64_t)rb * NT * (LAYOUT == MXQ_LAYOUT_W4ROW ? 128 : 144)) + lane * 16;
    }
    const int ptid = tid - N_CONS * 64;   // 0..255
    c.d_row = ptid & 127;
    c.d_qp = __builtin_amdgcn_readfirstlane(ptid >> 7);   // wave-uniform: producers 0,1 -> 0; 2,3 -> 1
    c.d_blk = c.d_row >> 4;
    c.d_r = c.d_row & 15;
    c.s4 = 0.f;
    c.z4 = 0.f;
    if (c.d_qp == 1 || LAYOUT == MXQ_LAYOUT_W4ROW) {
        int gn = n0 + c.d_row;
        gn = gn < N ? gn : N - 1;
        const float4 m = rowmeta[gn];
        c.s4 = mxq_scale(m.z, m.w, (uint32_t)m.y);
        c.z4 = m.x;
    }
    producer<ABL, LAYOUT>(c);
}

template <int ABL, int LAYOUT = MXQ_LAYOUT_MIXED>
static int launch4(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)mxq_gemm4_f16_kernel<ABL, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    mxq_gemm4_f16_kernel<ABL, LAYOUT><<<tiles_m * tiles_n, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n);
    return (int)hipGetLastError();
}

}   // namespace

// uniform layouts of the config-5 sweep (mxq_format.h): same kernel, different producer dequant
int mxq_launch_gemm4_layout_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                int layout, hipStream_t stream) {
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launch4<0, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_W2G16: return launch4<0, MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_W4ROW: return launch4<0, MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, stream);
    }
    return (int)hipErrorInvalidValue;
}
