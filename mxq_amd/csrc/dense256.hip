// Hoisted-dequant mode, large launches: y[M, N] = x[M, K] . w16[N, K]^T on the dequantised fp16 weight
// (mxq_linear_f16_hoisted: the dequant kernel has written w16 once; arithmetic contract as gemm8.hip's: fp16 operands,
// fp32 accumulation in K order, fp16 result -- reference: the implicit nn.Linear on the fake-quant weight,
// mxq_quant/main.py:85 / lib/eval.py:54).
//
// 256 x 256 tile, K-tile 64, 8 waves as 2 (tokens) x 4 (channels): a wave owns 128 tokens x 64 channels = 128
// accumulator registers, two waves per SIMD at up to 256 VGPRs.  The two wave groups (token halves) run a QUADRANT-PHASE
// PING-PONG: a K-tile is four phases, one 64-token x 32-channel quadrant of the wave tile each; a phase is
//     [fragment reads of the quadrant's NEW operand sub-tile | 2 LDS-DMA pieces | counted vmcnt]  s_barrier
//     [16 MFMAs]                                                                                  s_barrier
// and group 1 runs one barrier interval behind group 0, so that on every SIMD one wave multiplies while its partner
// reads / stages -- at the grain of 16 MFMAs (256 cycles), not of a whole K-tile.  (Round 2's 256 x 256 experiments,
// tools/experiments/dense256.hip, alternated whole K-tiles or interleaved everything in one stream: +5 % at best.)
//
// LDS: 8 slots of 16 KB = two K-tiles x four UNITS, one per phase, in the order the phases read them:
//     A(t) = the first 64 tokens of both token halves  (phase 1)      C(t) = the last 32 channels of all four channel
//     D(t) = the last 64 tokens of both                (phase 3)             quarters (phase 2)
//     B(t+1) = the first 32 channels of the NEXT K-tile, read in phase 4 into a second register set (so every phase
//     reads 8 or 4 fragments; reading it in the next phase 1 makes that phase's read segment 12 long: 0.1-1.1 % slower)
// Phase P (counted over the whole persistent loop, 4 per K-tile) reads unit P, stages unit P + 6 into slot (P + 6) % 8 --
// the slot's previous unit P - 2 was read two phases ago -- and waits vmcnt(10): everything up to unit P + 1 has
// landed, which is what phase P + 1 reads after this phase's barriers.  Every unit has five phases (>= 2500 cycles)
// between issue and wait; the wait is the same in every phase, and the unit sequence runs on across tile boundaries
// (the last phases of a tile stage the next tile's first units), so the pipeline never drains.
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
constexpr int UNIT = 128 * BK * 2;      // 16 KB: 128 rows x 64 k fp16
constexpr int SMEM = 8 * UNIT;          // 128 KB
enum { UA = 0, UB = 1, UC = 2, UD = 3 };

#define D256_FENCE() __builtin_amdgcn_sched_barrier(0)

#define D256_LANE_ID(v) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v))

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
__device__ __forceinline__ void bufdma16(rsrc_t rsrc, uint32_t voff, uint32_t soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

// Tile order (speed only): XCD e (= tile index & 7) owns a band of token-tile rows and walks it in blocks of 4 x-tiles
// against 8 weight tiles, so that the 32 workgroups of an XCD share operands in its L2; needs tiles_m % 32 == 0,
// otherwise a plain XCD-bijective order.
__device__ __forceinline__ void tile_of(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
    if ((tiles_m & 31) == 0) {
        const int e = t & 7, l = t >> 3;
        const int band_tiles = 4 * tiles_n;
        const int band = l / band_tiles, r = l - band * band_tiles;
        const int full = (tiles_n >> 3) * 32;
        int tml, tnl;
        if (r < full) {
            const int blk = r >> 5, q = r & 31;
            tnl = blk * 8 + (q & 7);
            tml = q >> 3;
        } else {
            const int rem = tiles_n & 7, r2 = r - full;
            tnl = (tiles_n & ~7) + r2 % rem;
            tml = r2 / rem;
        }
        tm = e * (tiles_m >> 3) + band * 4 + tml;
        tn = tnl;
        return;
    }
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    tm = lin % tiles_m;
    tn = lin / tiles_m;
}

struct Src {
    rsrc_t xr, wr;   // x rows m0.., weight rows n0.. of a tile; rows beyond M / N read as zeros (range check)
};
__device__ __forceinline__ void src_of(Src& s, const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, int M, int N,
                                       int K, int tm, int tn, bool valid) {
    const int m0 = tm * BM, n0 = tn * BN;
    const int rx = !valid ? 0 : (M - m0 < BM ? M - m0 : BM), rw = !valid ? 0 : (N - n0 < BN ? N - n0 : BN);
    s.xr = make_rsrc(x + (int64_t)(valid ? m0 : 0) * K, (uint32_t)rx * (uint32_t)K * 2u);
    s.wr = make_rsrc(w + (int64_t)(valid ? n0 : 0) * K, (uint32_t)rw * (uint32_t)K * 2u);
}

typedef half8 XFrag[2][4];   // [k half][token block]
typedef half8 WFrag[2][2];   // [k half][channel block]

// LDS slot of a unit: (K-tile parity) * 4 + position in the staging sequence A(t), C(t), D(t), B(t+1) -- the order in
// which the phases read them (B(t+1), the next K-tile's first weight sub-tile, is read one phase early into a second
// register set: 8 / 4 / 8 / 4 fragment reads per phase instead of 12 / 4 / 8 / 0)
constexpr int SEQ_A = 0, SEQ_C = 1, SEQ_D = 2, SEQ_B = 3;
template <int XS>
__device__ __forceinline__ void load_x(const char* smem, int par, int wr, int fr, int fq, XFrag& f) {
    const char* u = smem + (par * 4 + (XS ? SEQ_D : SEQ_A)) * UNIT;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) f[kk][j] = *(const half8*)(u + swz(wr * 64 + j * 16 + fr, kk * 4 + fq));
}
// weight sub-tile from the unit in sequence position SEQ of parity `par`
template <int SEQ>
__device__ __forceinline__ void load_w(const char* smem, int par, int wc, int fr, int fq, WFrag& f) {
    const char* u = smem + (par * 4 + SEQ) * UNIT;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) f[kk][i] = *(const half8*)(u + swz(wc * 32 + i * 16 + fr, kk * 4 + fq));
}
// D^T = W . x^T: acc[channel block][token block], a lane owns 4 consecutive channels of a token
template <int XS, int WS>
__device__ __forceinline__ void mfma_quadrant(f32x4 (&acc)[4][8], const WFrag& wf, const XFrag& xf) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[WS * 2 + i][XS * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][i], xf[kk][j], acc[WS * 2 + i][XS * 4 + j], 0, 0, 0);
}

// One wave's two pieces (8 rows x 128 B each) of unit KIND of K-tile kt (of the tile behind `s`) into LDS slot `slot`
template <int KIND>
__device__ __forceinline__ void stage(const Src& s, const uint32_t (&voff)[4][2], char* smem, int slot, int wave, int kt) {
    char* dst = smem + slot * UNIT + wave * 2048;
    const rsrc_t r = (KIND == UA || KIND == UD) ? s.xr : s.wr;
    bufdma16(r, voff[KIND][0], (uint32_t)kt * (BK * 2), dst);
    bufdma16(r, voff[KIND][1], (uint32_t)kt * (BK * 2), dst + 1024);
}

// LDS-free output: the 4 lanes that hold one token's 64 channels transpose their 4 x 4 grid of 8-byte cells with two
// butterfly stages of lane swaps, after which every lane stores 32 contiguous bytes (as gemm8.hip's store_tile_xpose)
__device__ __forceinline__ void store_tile(const f32x4 (&acc)[4][8], uint16_t* __restrict__ y, int M, int N, int m0, int n0,
                                           int wr, int wc, int fr, int fq) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    const int n = n0 + wc * 64 + fq * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t c[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i][0] = mxq_pack_f16(acc[i][j][0], acc[i][j][1]);
            c[i][1] = mxq_pack_f16(acc[i][j][2], acc[i][j][3]);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            u32x2v r;
            r = __builtin_amdgcn_permlane32_swap(c[0][d], c[2][d], false, false); c[0][d] = r[0]; c[2][d] = r[1];
            r = __builtin_amdgcn_permlane32_swap(c[1][d], c[3][d], false, false); c[1][d] = r[0]; c[3][d] = r[1];
            r = __builtin_amdgcn_permlane16_swap(c[0][d], c[1][d], false, false); c[0][d] = r[0]; c[1][d] = r[1];
            r = __builtin_amdgcn_permlane16_swap(c[2][d], c[3][d], false, false); c[2][d] = r[0]; c[3][d] = r[1];
        }
        const int m = m0 + wr * 128 + j * 16 + fr;
        if (m < M && n < N) {
            uint16_t* dst = y + (int64_t)m * N + n;
            __builtin_nontemporal_store((u32x4){c[0][0], c[0][1], c[1][0], c[1][1]}, (u32x4*)dst);
            __builtin_nontemporal_store((u32x4){c[2][0], c[2][1], c[3][0], c[3][1]}, (u32x4*)(dst + 8));
        }
    }
}

struct Regs {
    f32x4 acc[4][8];
    XFrag x0, x1;
    WFrag w0[2], w1;   // w0: by K-tile parity (the next K-tile's is read while this one's is still in use)
};

// Phase p (0..7) of a K-tile PAIR (K-tiles kt0, kt0 + 1 -> slot parities 0, 1).  It reads sequence unit p and stages
// sequence unit p + 6 (into slot (p + 6) % 8; the slot's previous unit p - 2 was read two phases ago), then waits until
// unit p + 1 -- what the next phase reads -- has landed.  LAST: the pair is the tile's last one: K-tiles kt0 + 2, kt0 + 3
// are the NEXT tile's K-tiles 0, 1 (`nxt`; an empty descriptor when there is none: zeros, no traffic).
template <int P, bool LAST>
__device__ __forceinline__ void phase(Regs& R, char* smem, const Src& cur, const Src& nxt, const uint32_t (&voff)[4][2], int kt0,
                                      int wave, int wr, int wc, int fr, int fq) {
    constexpr int PAR = P >> 2, PH = P & 3;
    // ---- fragment reads: the quadrant's new sub-tile (phase 4: the NEXT K-tile's first weight sub-tile)
    constexpr int SEQ = (P + 6) & 3, SLOT = (P + 6) & 7;
    constexpr int DK = ((P + 6) >> 2) + (SEQ == SEQ_B ? 1 : 0);              // its K-tile, relative to kt0
    constexpr int KIND = SEQ == SEQ_A ? UA : SEQ == SEQ_C ? UC : SEQ == SEQ_D ? UD : UB;
    if constexpr (PH == 0) load_x<0>(smem, PAR, wr, fr, fq, R.x0);
    else if constexpr (PH == 1) load_w<SEQ_C>(smem, PAR, wc, fr, fq, R.w1);
    else if constexpr (PH == 2) load_x<1>(smem, PAR, wr, fr, fq, R.x1);
    else load_w<SEQ_B>(smem, PAR, wc, fr, fq, R.w0[PAR ^ 1]);
    D256_FENCE();
    // ---- stage sequence unit P + 6
    if constexpr (LAST && DK >= 2) stage<KIND>(nxt, voff, smem, SLOT, wave, DK - 2);
    else stage<KIND>(cur, voff, smem, SLOT, wave, kt0 + DK);
    D256_FENCE();
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // units <= P + 1 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    D256_FENCE();
    // (s_setprio 1 / 3 around the MFMA cluster: 0.5-1 % slower than none; the staging before the reads: 0.5 % slower)
    if constexpr (PH == 0) mfma_quadrant<0, 0>(R.acc, R.w0[PAR], R.x0);
    else if constexpr (PH == 1) mfma_quadrant<0, 1>(R.acc, R.w1, R.x0);
    else if constexpr (PH == 2) mfma_quadrant<1, 1>(R.acc, R.w1, R.x1);
    else mfma_quadrant<1, 0>(R.acc, R.w0[PAR], R.x1);
    D256_FENCE();
    __builtin_amdgcn_s_barrier();
}

template <bool LAST>
__device__ __forceinline__ void pair(Regs& R, char* smem, const Src& cur, const Src& nxt, const uint32_t (&voff)[4][2], int kt0,
                                     int wave, int wr, int wc, int fr, int fq) {
    phase<0, LAST>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
    phase<1, LAST>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
    phase<2, LAST>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
    phase<3, LAST>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
    phase<4, LAST>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
    phase<5, LAST>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
    phase<6, LAST>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
    phase<7, LAST>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
}

__global__ __launch_bounds__(THREADS) void mxq_dense256_f16_kernel(const uint16_t* __restrict__ x,
                                                                   const uint16_t* __restrict__ w,
                                                                   uint16_t* __restrict__ y, int M, int N, int K,
                                                                   int tiles_m, int tiles_n, int tiles, int grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = K / BK;           // even, >= 2 (launcher)
    const int wr = wave >> 2, wc = wave & 3;
    int ln;
    D256_LANE_ID(ln);
    // per-lane source offsets of the wave's two pieces of each unit kind: piece pc = 2 wave + h holds the unit's rows
    // 8 pc .. 8 pc + 7, a lane's 16 bytes = k-slot (lane & 7) ^ (row & 7) of row (lane >> 3)   [XOR swizzle on the source side]
    uint32_t voff[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int lr = (wave * 2 + h) * 8 + (ln >> 3);                    // row inside the unit, 0..127
        const uint32_t sw = (uint32_t)((ln & 7) ^ ((ln >> 3) & 7)) << 4;
        const int xa = (lr >> 6) * 128 + (lr & 63);                       // A: first 64 tokens of each token half
        const int wb = (lr >> 5) * 64 + (lr & 31);                        // B: first 32 channels of each channel quarter
        voff[UA][h] = (uint32_t)xa * (uint32_t)K * 2u + sw;
        voff[UD][h] = (uint32_t)(xa + 64) * (uint32_t)K * 2u + sw;
        voff[UB][h] = (uint32_t)wb * (uint32_t)K * 2u + sw;
        voff[UC][h] = (uint32_t)(wb + 32) * (uint32_t)K * 2u + sw;
    }
    int tm, tn;
    tile_of(blockIdx.x, tiles_m, tiles_n, tm, tn);
    Src cur, nxt;
    src_of(cur, x, w, M, N, K, tm, tn, true);
    // prologue: B(0) (sequence unit -1, slot 7) and sequence units 0..5 = A(0), C(0), D(0), B(1), A(1), C(1)
    stage<UB>(cur, voff, smem, 7, wave, 0);
    stage<UA>(cur, voff, smem, 0, wave, 0);
    stage<UC>(cur, voff, smem, 1, wave, 0);
    stage<UD>(cur, voff, smem, 2, wave, 0);
    stage<UB>(cur, voff, smem, 3, wave, 1);
    stage<UA>(cur, voff, smem, 4, wave, 1);
    stage<UC>(cur, voff, smem, 5, wave, 1);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // B(0), A(0)
    __builtin_amdgcn_s_barrier();
    if (wr) __builtin_amdgcn_s_barrier();              // group 1 runs one barrier interval behind group 0
    Regs R;
    {
        D256_LANE_ID(ln);
        load_w<SEQ_B>(smem, 1, wc, ln & 15, ln >> 4, R.w0[0]);   // "phase -1": the first K-tile's first weight sub-tile
    }
    for (int tile = blockIdx.x; tile < tiles; tile += grid) {
        D256_LANE_ID(ln);
        const int fr = ln & 15, fq = ln >> 4;
        const int m0 = tm * BM, n0 = tn * BN;
        const bool more = tile + grid < tiles;
        if (more) tile_of(tile + grid, tiles_m, tiles_n, tm, tn);
        src_of(nxt, x, w, M, N, K, tm, tn, more);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) R.acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt0 = 0; kt0 + 2 < NT; kt0 += 2) pair<false>(R, smem, cur, nxt, voff, kt0, wave, wr, wc, fr, fq);
        pair<true>(R, smem, cur, nxt, voff, NT - 2, wave, wr, wc, fr, fq);
        store_tile(R.acc, y, M, N, m0, n0, wr, wc, fr, fq);
        cur = nxt;
    }
    if (!wr) __builtin_amdgcn_s_barrier();             // group 0's matching extra barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (empty-descriptor) pieces still write zeros into this LDS
}

int cu_count8() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8)
            cus = n / 8 * 8;
        else
            cus = 256;
    }
    return cus;
}

}   // namespace

// -> 0 launched; MXQ_NOT_MY_SHAPE: the shape is not this kernel's (K-tile count odd or < 2, rows beyond the 32-bit descriptor offsets, or
// -- unless force -- a tile count that fills the chip too unevenly): the caller takes the 256 x 128 kernel
int mxq_launch_dense256_f16(const void* x, const void* w16, void* y, int M, int N, int K, int force, hipStream_t stream) {
    if (K % (2 * BK) != 0 || K < 2 * BK || (int64_t)BM * K * 2 >= ((int64_t)1 << 32)) return MXQ_NOT_MY_SHAPE;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    const int cus = cu_count8();
    if (!force) {
        // Take this kernel when its ~17 % per-flop advantage (profiles/r03_dense256.txt) outweighs the coarser tile
        // quantisation: share of the launch's (rounds x CUs) tile slots that hold a tile, here and for 256 x 128 tiles.
        const int t128 = tiles_m * ((N + 127) / 128);
        const double e256 = (double)tiles / ((double)((tiles + cus - 1) / cus) * cus);
        const double e128 = (double)t128 / ((double)((t128 + cus - 1) / cus) * cus);
        if (1.17 * e256 < e128) return MXQ_NOT_MY_SHAPE;
    }
    hipError_t e = hipFuncSetAttribute((const void*)mxq_dense256_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    const int grid = tiles < cus ? tiles : cus;
    mxq_dense256_f16_kernel<<<grid, THREADS, SMEM, stream>>>((const uint16_t*)x, (const uint16_t*)w16, (uint16_t*)y, M, N, K,
                                                            tiles_m, tiles_n, tiles, grid);
    return (int)hipGetLastError();
}

#ifdef MXQ_PROFILING
extern "C" int mxq_prof_dense256_f16(const void* x, const void* w16, void* y, int M, int N, int K, void* stream_) {
    return mxq_launch_dense256_f16(x, w16, y, M, N, K, 1, (hipStream_t)stream_);
}
#endif
