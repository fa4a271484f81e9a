// EXPERIMENT (round 3): the quadrant-phase ping-pong schedule on a 256-token x 128-channel tile with DEDICATED DMA
// WAVES: 8 MFMA waves (64 x 64 wave tiles; the two groups of four run one barrier interval apart; 8 fragment reads per
// 16 MFMAs) + 4 DMA waves that issue every LDS-DMA piece and own the counted vmcnt.  Self-contained translation unit;
// built into a variant library by tools/experiments/pp128dma/build.sh and timed through tools/ab_gemm.py
// (variant xlib:<lib>:mxq_exp_pp128dma_f16).  y = x . w16^T, fp16 operands, fp32 accumulation in K order.
// RESULT (gpurun_out/r3c47; correct on ragged shapes and at 2048 tokens): 63.4 / 167.7 / 149.9 us at 2048 tokens x
// (4096^2, 11008 x 4096, 4096 x 11008) against 60.0 / 159.8 / 148.5 for the product's 256 x 128 kernel (gemm8.hip dense
// instantiation: the same tile and DMA waves, MFMA waves in phase, one barrier per K-step) and 69.9 / 185.5 / 165.9 for the
// fused kernel: no gain from the ping-pong on 64 x 64 wave tiles -- four barriers per K-tile buy nothing when a wave's
// read segment (8 fragment reads) is as long relative to its 16 MFMAs as here.  Not shipped; the hoisted mode at 2048
// tokens stays behind the fused kernel by the dequant pass (r03_dense256.txt section 4).
// Allocator note: the ring position must be a RUN-TIME base.  Three instantiations behind a switch (the first version,
// and tools/experiments/dense128pp) made hipcc give every MFMA different source and destination accumulators and spill.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int BM = 256, BN = 128, BK = 64, N_MMA = 8, N_DMA = 4, THREADS = (N_MMA + N_DMA) * 64;
constexpr int UNIT = 128 * BK * 2;
constexpr int SMEM = 9 * UNIT;   // ring of three K-tiles x {A, W, D}
enum { KA = 0, KW = 1, KD = 2 };
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define LANE_ID(v) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v))
__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
__device__ __forceinline__ void bufdma16(rsrc_t rsrc, uint32_t voff, uint32_t soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void tile_of(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
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
struct Src { rsrc_t xr, wr; };
__device__ __forceinline__ void src_of(Src& s, const uint16_t* x, const uint16_t* w, int M, int N, int K, int tm, int tn, bool valid) {
    const int m0 = tm * BM, n0 = tn * BN;
    const int rx = !valid ? 0 : (M - m0 < BM ? M - m0 : BM), rw = !valid ? 0 : (N - n0 < BN ? N - n0 : BN);
    s.xr = make_rsrc(x + (int64_t)(valid ? m0 : 0) * K, (uint32_t)rx * (uint32_t)K * 2u);
    s.wr = make_rsrc(w + (int64_t)(valid ? n0 : 0) * K, (uint32_t)rw * (uint32_t)K * 2u);
}
typedef half8 XF[2][2];
typedef half8 WF[2][4];
struct Regs { f32x4 acc[4][4]; XF x; WF w; };   // the second token half reuses the first one's fragment registers
template <int XS>
__device__ __forceinline__ void mfma_half(f32x4 (&acc)[4][4], const WF& wf, const XF& xf) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][XS * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][i], xf[kk][j], acc[i][XS * 2 + j], 0, 0, 0);
}
// MFMA waves: K-tile in ring position R.  ad = the lane's swizzled fragment offsets inside a unit {W kk 0, W kk 1, x kk 0,
// x kk 1}; the ring position's base is added here and kept opaque, so that the compiler does not carry 3 x 6 address
// registers through the loop (it spilled them at the 168-VGPR cap)
#define OPAQUE(v) asm volatile("" : "+v"(v))
__device__ __forceinline__ void ktile_mma(Regs& Q, const char* smem, const uint32_t (&ad)[4], uint32_t rbase) {
    // (the ring position is a RUN-TIME base: three instantiations behind a switch made the allocator copy the accumulators)
    uint32_t w0 = ad[0] + rbase + KW * UNIT, w1 = ad[1] + rbase + KW * UNIT;
    uint32_t a0 = ad[2] + rbase + KA * UNIT, a1 = ad[3] + rbase + KA * UNIT;
    OPAQUE(w0); OPAQUE(w1); OPAQUE(a0); OPAQUE(a1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        Q.w[0][i] = *(const half8*)(smem + w0 + i * 2048);
        Q.w[1][i] = *(const half8*)(smem + w1 + i * 2048);
    }
    FENCE();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        Q.x[0][j] = *(const half8*)(smem + a0 + j * 2048);
        Q.x[1][j] = *(const half8*)(smem + a1 + j * 2048);
    }
    FENCE();
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    FENCE();
    mfma_half<0>(Q.acc, Q.w, Q.x);
    FENCE();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        Q.x[0][j] = *(const half8*)(smem + a0 + (KD - KA) * UNIT + j * 2048);
        Q.x[1][j] = *(const half8*)(smem + a1 + (KD - KA) * UNIT + j * 2048);
    }
    FENCE();
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    FENCE();
    mfma_half<1>(Q.acc, Q.w, Q.x);
    FENCE();
    __builtin_amdgcn_s_barrier();
}
// DMA waves: a unit is 16 pieces, 4 per DMA wave; the wave's pieces of a unit are 8 rows apart, A and D 32 rows apart:
// ONE per-lane offset per operand, everything else in the scalar offset
template <int KIND>
__device__ __forceinline__ void stage(const Src& s, const uint32_t (&voff)[2], char* smem, int ringpos, int dw, int kt, uint32_t row_bytes) {
    char* dst = smem + (ringpos * 3 + KIND) * UNIT + dw * 4096;
    const rsrc_t r = KIND == KW ? s.wr : s.xr;
    const uint32_t v = voff[KIND == KW ? 1 : 0];
#pragma unroll
    for (int h = 0; h < 4; ++h)
        bufdma16(r, v, (uint32_t)kt * (BK * 2) + (uint32_t)(h * 8 + (KIND == KD ? 32 : 0)) * row_bytes, dst + h * 1024);
}
__device__ __forceinline__ void ktile_dma(char* smem, const Src& cur, const Src& nxt, const uint32_t (&voff)[2], int kt, int NT, int dw, uint32_t row_bytes, int ring) {
    const int RS = ring == 0 ? 2 : ring - 1;      // (ring + 2) % 3
    const bool wrap = kt + 2 >= NT;
    const int skt = wrap ? kt + 2 - NT : kt + 2;
    Src s;
    s.xr = wrap ? nxt.xr : cur.xr;
    s.wr = wrap ? nxt.wr : cur.wr;
    stage<KA>(s, voff, smem, RS, dw, skt, row_bytes);
    stage<KW>(s, voff, smem, RS, dw, skt, row_bytes);
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // D of this K-tile has landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    stage<KD>(s, voff, smem, RS, dw, skt, row_bytes);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // A, W of the next K-tile have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ void store_tile(const f32x4 (&acc)[4][4], uint16_t* y, int M, int N, int m0, int n0, int mb, int nb, int fr, int fq) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    const int n = n0 + nb * 64 + fq * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t c[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i][0] = pack_f16(acc[i][j][0], acc[i][j][1]);
            c[i][1] = pack_f16(acc[i][j][2], acc[i][j][3]);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            u32x2v r;
            r = __builtin_amdgcn_permlane32_swap(c[0][d], c[2][d], false, false); c[0][d] = r[0]; c[2][d] = r[1];
            r = __builtin_amdgcn_permlane32_swap(c[1][d], c[3][d], false, false); c[1][d] = r[0]; c[3][d] = r[1];
            r = __builtin_amdgcn_permlane16_swap(c[0][d], c[1][d], false, false); c[0][d] = r[0]; c[1][d] = r[1];
            r = __builtin_amdgcn_permlane16_swap(c[2][d], c[3][d], false, false); c[2][d] = r[0]; c[3][d] = r[1];
        }
        const int m = m0 + mb * 64 + j * 16 + fr;
        if (m < M && n < N) {
            uint16_t* dst = y + (int64_t)m * N + n;
            __builtin_nontemporal_store((u32x4){c[0][0], c[0][1], c[1][0], c[1][1]}, (u32x4*)dst);
            __builtin_nontemporal_store((u32x4){c[2][0], c[2][1], c[3][0], c[3][1]}, (u32x4*)(dst + 8));
        }
    }
}

__global__ __launch_bounds__(THREADS) void pp128dma_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                           uint16_t* __restrict__ y, int M, int N, int K, int tiles_m,
                                                           int tiles_n, int tiles, int grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = K / BK;
    int tm, tn;
    tile_of(blockIdx.x, tiles_m, tiles_n, tm, tn);
    int ring = 0;
    if (wave >= N_MMA) {
        // ---------------- DMA waves
        const int dw = wave - N_MMA;
        int ln;
        LANE_ID(ln);
        const uint32_t row_bytes = (uint32_t)K * 2u;
        const uint32_t sw = (uint32_t)((ln & 7) ^ ((ln >> 3) & 7)) << 4;
        const uint32_t voff[2] = {(uint32_t)(dw * 64 + (ln >> 3)) * row_bytes + sw,     // A (D = + 32 rows): token block64 dw
                                  (uint32_t)(dw * 32 + (ln >> 3)) * row_bytes + sw};    // W: channels 32 dw ..
        Src cur, nxt;
        src_of(cur, x, w, M, N, K, tm, tn, true);
        stage<KA>(cur, voff, smem, 0, dw, 0, row_bytes);
        stage<KW>(cur, voff, smem, 0, dw, 0, row_bytes);
        stage<KD>(cur, voff, smem, 0, dw, 0, row_bytes);
        stage<KA>(cur, voff, smem, 1, dw, 1, row_bytes);
        stage<KW>(cur, voff, smem, 1, dw, 1, row_bytes);
        stage<KD>(cur, voff, smem, 1, dw, 1, row_bytes);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // A(0), W(0)
        __builtin_amdgcn_s_barrier();
        for (int tile = blockIdx.x; tile < tiles; tile += grid) {
            const bool more = tile + grid < tiles;
            if (more) tile_of(tile + grid, tiles_m, tiles_n, tm, tn);
            src_of(nxt, x, w, M, N, K, tm, tn, more);
            for (int kt = 0; kt < NT; ++kt) {
                ktile_dma(smem, cur, nxt, voff, kt, NT, dw, row_bytes, ring);
                ring = ring == 2 ? 0 : ring + 1;
            }
            cur = nxt;
        }
        __builtin_amdgcn_s_barrier();                        // the stagger's extra barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ---------------- MFMA waves
    const int g = wave >> 2, q = wave & 3;
    const int mb = g * 2 + (q >> 1), nb = q & 1;
    __builtin_amdgcn_s_barrier();                            // prologue data
    if (g) __builtin_amdgcn_s_barrier();                     // group 1 runs one barrier interval behind
    Regs Q;
    int ln;
    LANE_ID(ln);
    const int fr = ln & 15, fq = ln >> 4;
    const uint32_t ad[4] = {(uint32_t)swz(nb * 64 + fr, fq), (uint32_t)swz(nb * 64 + fr, 4 + fq),
                            (uint32_t)swz(mb * 32 + fr, fq), (uint32_t)swz(mb * 32 + fr, 4 + fq)};
    for (int tile = blockIdx.x; tile < tiles; tile += grid) {
        const int m0 = tm * BM, n0 = tn * BN;
        if (tile + grid < tiles) tile_of(tile + grid, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Q.acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < NT; ++kt) {
            ktile_mma(Q, smem, ad, (uint32_t)ring * (3 * UNIT));
            ring = ring == 2 ? 0 : ring + 1;
        }
        store_tile(Q.acc, y, M, N, m0, n0, mb, nb, fr, fq);
    }
    if (!g) __builtin_amdgcn_s_barrier();
}
}   // namespace

extern "C" int mxq_exp_pp128dma_f16(const void* x, const void* w16, void* y, int M, int N, int K, void* stream) {
    if (K < 2 * BK || K % BK) return -1;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    hipError_t e = hipFuncSetAttribute((const void*)pp128dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    const int grid = tiles < 256 ? tiles : 256;
    pp128dma_kernel<<<grid, THREADS, SMEM, (hipStream_t)stream>>>((const uint16_t*)x, (const uint16_t*)w16, (uint16_t*)y, M, N, K,
                                                                 tiles_m, tiles_n, tiles, grid);
    return (int)hipGetLastError();
}
