// EXPERIMENT RECORD (round 2, second session) -- not built, not part of the product.
// A 256 x 256-tile MFMA kernel for the hoisted-dequant mode (dense fp16 weight).  It was compiled inside csrc/gemm8.hip's
// anonymous namespace (it uses that file's helpers: swz, bufdma16, make_rsrc, tile_of_block, MXQ_FENCE, MXQ_LANE_ID,
// stamp, cu_count) and driven through the profiling entries at the end (tools/ab_gemm.py variants dense256 / dense256e /
// dense256p / dense256s / dense256s2 / dense256s3 / dense256h / (7: priority toggling) / dense256pp = modes 0..8; dense256_clock.py = stamps + clock).
// Every mode is correct on every shape tried (<= 1e-3 against the fp32 product; 2300 / 4096 / 32768 tokens, ragged
// edges, K = 128).  Measurements and verdict: profiles/r02_gemmx_experiment.txt ("dense256").
// ------------------------------------------------------------------------------------------------
// hoisted-dequant mode, 256 x 256 tile ("dense256")
// ------------------------------------------------------------------------------------------------
// For launches with many token tiles the 256 x 128 kernel above pays 128 KB of fragment reads and 48 KB of LDS-DMA per
// 4.2 MFLOP K-step; under sustained load the chip is at its power limit (profiles/r02_coop_experiment.txt), so bytes
// moved per flop are time.  Here: 8 waves, ALL of them MFMA waves with a 128-token x 64-channel tile each (2 x 4 waves,
// 128 accumulator registers, two waves per SIMD at up to 256 VGPRs), every wave also DMAs 4 pieces of the x tile and 4
// of the weight tile per K-step: 192 KB of fragment reads and 64 KB of DMA per 8.4 MFLOP K-step (-25 % / -33 % per
// flop), one barrier per 2048 MFMA cycles instead of per 1024.  LDS: two 32 KB stages each for x and the weight (a
// third does not fit); the DMA of step t+1 is issued in step t, behind the barrier that ended the reads of its slot, and
// has the whole step (>= 2048 cycles) to land.  The MFMAs of (t, kk = 1) run at the head of step t+1, as above.
namespace d256 {
constexpr int BM2 = 256, BN2 = 256, WAVES2 = 8, THREADS2 = WAVES2 * 64;
constexpr int A2 = BM2 * BK * 2, W2 = BN2 * BK * 2;
constexpr int OFF_A2 = 0, OFF_W2 = 2 * A2, SMEM2 = 2 * A2 + 2 * W2;
static_assert(SMEM2 <= 160 * 1024, "LDS budget");
typedef half8 FragW[4];
typedef half8 FragX[8];

struct Dma2 {
    rsrc_t xr, wr;        // x rows m0.., weight rows n0.. of this tile (range-checked: rows beyond M / N read as zeros)
    uint32_t xv[4], wv[4];
};
__device__ __forceinline__ void dma_setup(Dma2& d, const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, int M,
                                          int N, int K, int m0, int n0, int wave, int lane) {
    const int rx = M - m0 < BM2 ? M - m0 : BM2, rw = N - n0 < BN2 ? N - n0 : BN2;
    d.xr = make_rsrc(x + (int64_t)m0 * K, (uint32_t)rx * (uint32_t)K * 2u);
    d.wr = make_rsrc(w + (int64_t)n0 * K, (uint32_t)rw * (uint32_t)K * 2u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const uint32_t off = (uint32_t)row * (uint32_t)K * 2u + ((((uint32_t)lane & 7u) ^ ((uint32_t)row & 7u)) << 4);
        d.xv[i] = off;
        d.wv[i] = off;
    }
}
__device__ __forceinline__ void issue_xs(const Dma2& d, char* smem, int wave, int t) {
    char* dst = smem + OFF_A2 + (t & 1) * A2 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) bufdma16(d.xr, d.xv[i], (uint32_t)t * (BK * 2), dst + i * 1024);
}
__device__ __forceinline__ void issue_ws(const Dma2& d, char* smem, int wave, int t) {
    char* dst = smem + OFF_W2 + (t & 1) * W2 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) bufdma16(d.wr, d.wv[i], (uint32_t)t * (BK * 2), dst + i * 1024);
}
// one piece of step t's tiles (q = 0..3: x, 4..7: weight); soff = t * 128, or an offset beyond the buffers: zeros, no traffic
__device__ __forceinline__ void issue_piece(const Dma2& d, char* smem, int wave, int t, uint32_t soff, int q) {
    if (q < 4) bufdma16(d.xr, d.xv[q], soff, smem + OFF_A2 + (t & 1) * A2 + wave * 4096 + q * 1024);
    else bufdma16(d.wr, d.wv[q - 4], soff, smem + OFF_W2 + (t & 1) * W2 + wave * 4096 + (q - 4) * 1024);
}
__device__ __forceinline__ void load_frags2(const char* smem, int t, int kk, int wm, int wn, int fr, int fq, FragW& wf, FragX& xf) {
    const char* a = smem + OFF_A2 + (t & 1) * A2;
    const char* w = smem + OFF_W2 + (t & 1) * W2;
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *(const half8*)(w + swz(wn * 64 + i * 16 + fr, kk * 4 + fq));
#pragma unroll
    for (int j = 0; j < 8; ++j) xf[j] = *(const half8*)(a + swz(wm * 128 + j * 16 + fr, kk * 4 + fq));
}
template <int I0, int I1>
__device__ __forceinline__ void mfma_rows2(f32x4 (&acc)[4][8], const FragW& wf, const FragX& xf) {
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
}
// the LDS-free output of store_tile_xpose, 8 token blocks per wave
__device__ __forceinline__ void store_tile2(const f32x4 (&acc)[4][8], uint16_t* __restrict__ y, int M, int N, int m0, int n0,
                                            int wm, int wn, int fr, int fq) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    const int n = n0 + wn * 64 + fq * 16;
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
        const int m = m0 + wm * 128 + j * 16 + fr;
        if (m < M && n < N) {
            uint16_t* dst = y + (int64_t)m * N + n;
            *(u32x4*)dst = (u32x4){c[0][0], c[0][1], c[1][0], c[1][1]};
            *(u32x4*)(dst + 8) = (u32x4){c[2][0], c[2][1], c[3][0], c[3][1]};
        }
    }
}

// MODE 3's K loop for one stagger order (FIRST = the wave issues its DMA piece BEFORE each row of 8 MFMAs; its SIMD partner after)
template <bool FIRST, int SLEEP>
__device__ __forceinline__ void steps_staggered(char* smem, const Dma2& cur, int wave, int NT, int wm, int wn, int fr, int fq,
                                                f32x4 (&acc)[4][8], FragW& wf0, FragX& xf0, FragW& wf1, FragX& xf1) {
    for (int t = 1; t < NT; ++t) {
        if constexpr (SLEEP > 0) {      // the stagger: the SIMD partner (wm = 1) starts its step half a row-and-piece period later
            if (wm) __builtin_amdgcn_s_sleep(SLEEP);
        }
        const uint32_t so = t + 1 < NT ? (uint32_t)(t + 1) * (BK * 2) : 0x80000000u;
#define D256_ROW(SET_W, SET_X, R, Q)                                                   \
        if constexpr (FIRST) { issue_piece(cur, smem, wave, t + 1, so, Q); MXQ_FENCE(); } \
        mfma_rows2<R, R + 1>(acc, SET_W, SET_X);                                       \
        MXQ_FENCE();                                                                   \
        if constexpr (!FIRST) { issue_piece(cur, smem, wave, t + 1, so, Q); MXQ_FENCE(); }
        D256_ROW(wf1, xf1, 0, 0)
        load_frags2(smem, t, 0, wm, wn, fr, fq, wf0, xf0);
        MXQ_FENCE();
        D256_ROW(wf1, xf1, 1, 1)
        D256_ROW(wf1, xf1, 2, 2)
        D256_ROW(wf1, xf1, 3, 3)
        load_frags2(smem, t, 1, wm, wn, fr, fq, wf1, xf1);
        MXQ_FENCE();
        D256_ROW(wf0, xf0, 0, 4)
        D256_ROW(wf0, xf0, 1, 5)
        D256_ROW(wf0, xf0, 2, 6)
        D256_ROW(wf0, xf0, 3, 7)
#undef D256_ROW
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}


// ---- MODE 8, "ping-pong": the two waves of a SIMD never multiply at the same time.  Phase 2t: the A waves (wm = 0) run
// step t's 64 MFMAs from registers while the B waves read step t's fragments; phase 2t+1: B multiplies, A reads step
// t+1's fragments and issues ALL of DMA(t+2) (16 pieces per A wave: rows 64 wn .. of both tiles).  One barrier per
// phase.  A multiplying wave has every operand in registers: a bare MFMA stream at the pipe's rate.
struct Dma3 { rsrc_t xr, wr; uint32_t v[8]; };
__device__ __forceinline__ void dma3_setup(Dma3& d, const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, int M,
                                           int N, int K, int m0, int n0, int wn, int lane) {
    const int rx = M - m0 < BM2 ? M - m0 : BM2, rw = N - n0 < BN2 ? N - n0 : BN2;
    d.xr = make_rsrc(x + (int64_t)m0 * K, (uint32_t)rx * (uint32_t)K * 2u);
    d.wr = make_rsrc(w + (int64_t)n0 * K, (uint32_t)rw * (uint32_t)K * 2u);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wn * 64 + i * 8 + (lane >> 3);
        d.v[i] = (uint32_t)row * (uint32_t)K * 2u + ((((uint32_t)lane & 7u) ^ ((uint32_t)row & 7u)) << 4);
    }
}
__device__ __forceinline__ void issue3(const Dma3& d, char* smem, int wn, int t) {
    char* dx = smem + OFF_A2 + (t & 1) * A2 + wn * 8192;
    char* dw = smem + OFF_W2 + (t & 1) * W2 + wn * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) bufdma16(d.xr, d.v[i], (uint32_t)t * (BK * 2), dx + i * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) bufdma16(d.wr, d.v[i], (uint32_t)t * (BK * 2), dw + i * 1024);
}

// Tile order: XCD e (= t & 7) owns the token-tile rows [e * tiles_m / 8, (e + 1) * tiles_m / 8) and walks them in bands of
// 4 rows x blocks of 8 panels, so that the 32 workgroups of an XCD run 4 x tiles against 8 weight tiles (both 32 KB per
// K-step here); needs tiles_m % 32 == 0, otherwise the order of the 256 x 128 kernel.
__device__ __forceinline__ void tile_of_block2(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
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
    tile_of_block(t, tiles_m, tiles_n, tm, tn);
}

template <int MODE>
__global__ __launch_bounds__(THREADS2) void mxq_dense256_f16_kernel(const uint16_t* __restrict__ x,
                                                                    const uint16_t* __restrict__ w,
                                                                    uint16_t* __restrict__ y, int M, int N, int K,
                                                                    int tiles_m, int tiles_n, int tiles, int grid,
                                                                    unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = K / BK;
    const int wm = wave >> 2, wn = wave & 3;
    u64t rt0 = 0, mt0 = 0, stw = 0, stl = 0, stv = 0, stb = 0, stn = 0;
    if (dbg) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
        mt0 = stamp();
    }
    int tm, tn;
    tile_of_block2(blockIdx.x, tiles_m, tiles_n, tm, tn);
    int ln;
    MXQ_LANE_ID(ln);
    if constexpr (MODE == 8) {
        Dma3 cur3;
        if (wm == 0) {
            dma3_setup(cur3, x, w, M, N, K, tm * BM2, tn * BN2, wn, ln);
            issue3(cur3, smem, wn, 0);
            if (NT > 1) issue3(cur3, smem, wn, 1);
        }
        for (int tile = blockIdx.x; tile < tiles; tile += grid) {
            MXQ_LANE_ID(ln);
            const int fr = ln & 15, fq = ln >> 4;
            const int m0 = tm * BM2, n0 = tn * BN2;
            const bool more = tile + grid < tiles;
            if (more) tile_of_block2(tile + grid, tiles_m, tiles_n, tm, tn);
            f32x4 acc[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            FragW wf0, wf1;
            FragX xf0, xf1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // DMA(0), DMA(1) (A waves) and the previous tile's stores
            __builtin_amdgcn_s_barrier();
            if (wm == 0) {
                // phase -1: A reads step 0's fragments
                load_frags2(smem, 0, 0, wm, wn, fr, fq, wf0, xf0);
                load_frags2(smem, 0, 1, wm, wn, fr, fq, wf1, xf1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                for (int t = 0; t < NT; ++t) {
                    // phase 2t: multiply step t
                    mfma_rows2<0, 4>(acc, wf0, xf0);
                    mfma_rows2<0, 4>(acc, wf1, xf1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // DMA(t+1), issued two phases ago, has landed
                    __builtin_amdgcn_s_barrier();
                    // phase 2t+1: stage t & 1 is free (B read it in phase 2t): all of DMA(t+2); step t+1's fragments
                    if (t + 2 < NT) issue3(cur3, smem, wn, t + 2);
                    if (t + 1 < NT) {
                        load_frags2(smem, t + 1, 0, wm, wn, fr, fq, wf0, xf0);
                        load_frags2(smem, t + 1, 1, wm, wn, fr, fq, wf1, xf1);
                    } else {
                        // the last phase of the tile (B multiplies): this tile's output, the next tile's first DMAs
                        if (more) {
                            dma3_setup(cur3, x, w, M, N, K, tm * BM2, tn * BN2, wn, ln);
                            issue3(cur3, smem, wn, 0);
                            if (NT > 1) issue3(cur3, smem, wn, 1);
                        }
                        store_tile2(acc, y, M, N, m0, n0, wm, wn, fr, fq);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            } else {
                __builtin_amdgcn_s_barrier();                               // phase -1: B idles
                for (int t = 0; t < NT; ++t) {
                    // phase 2t: read step t's fragments
                    load_frags2(smem, t, 0, wm, wn, fr, fq, wf0, xf0);
                    load_frags2(smem, t, 1, wm, wn, fr, fq, wf1, xf1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    // phase 2t+1: multiply step t
                    mfma_rows2<0, 4>(acc, wf0, xf0);
                    mfma_rows2<0, 4>(acc, wf1, xf1);
                    __builtin_amdgcn_s_barrier();
                }
                store_tile2(acc, y, M, N, m0, n0, wm, wn, fr, fq);
            }
        }
        return;
    }
    constexpr bool EARLY = MODE == 1;
    Dma2 cur, nxt;
    dma_setup(cur, x, w, M, N, K, tm * BM2, tn * BN2, wave, ln);
    issue_xs(cur, smem, wave, 0);
    issue_ws(cur, smem, wave, 0);
    if constexpr (MODE == 2 || MODE == 6) {
        if (NT > 1) { issue_xs(cur, smem, wave, 1); issue_ws(cur, smem, wave, 1); }
    }
    for (int tile = blockIdx.x; tile < tiles; tile += grid) {
        MXQ_LANE_ID(ln);
        const int fr = ln & 15, fq = ln >> 4;
        const int m0 = tm * BM2, n0 = tn * BN2;
        const bool more = tile + grid < tiles;
        if (more) tile_of_block2(tile + grid, tiles_m, tiles_n, tm, tn);
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        FragW wf0, wf1;
        FragX xf0, xf1;
        if constexpr (MODE == 2) {
            // Two barriers per K-step.  B_a: every wave holds step t's fragments -> stage t & 1 is free, and the DMAs of
            // step t+2 go into it at once; B_b: the DMAs of step t+1 (issued a step and a half ago) have landed
            // (vmcnt(8): only the 8 pieces just issued stay in flight).
            if (NT > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            load_frags2(smem, 0, 0, wm, wn, fr, fq, wf0, xf0);
            load_frags2(smem, 0, 1, wm, wn, fr, fq, wf1, xf1);
            mfma_rows2<0, 2>(acc, wf0, xf0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // B_a(0)
            if (NT > 2) { issue_xs(cur, smem, wave, 2); issue_ws(cur, smem, wave, 2); }
            mfma_rows2<2, 4>(acc, wf0, xf0);
            if (NT > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                   // B_b(0)
            for (int t = 1; t < NT; ++t) {
                const bool issue = t + 2 < NT;
                mfma_rows2<0, 1>(acc, wf1, xf1);
                MXQ_FENCE();
                load_frags2(smem, t, 0, wm, wn, fr, fq, wf0, xf0);
                MXQ_FENCE();
                mfma_rows2<1, 4>(acc, wf1, xf1);
                MXQ_FENCE();
                load_frags2(smem, t, 1, wm, wn, fr, fq, wf1, xf1);
                MXQ_FENCE();
                mfma_rows2<0, 2>(acc, wf0, xf0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                               // B_a(t)
                if (issue) { issue_xs(cur, smem, wave, t + 2); issue_ws(cur, smem, wave, t + 2); }
                MXQ_FENCE();
                mfma_rows2<2, 4>(acc, wf0, xf0);
                if (issue) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                               // B_b(t)
            }
            if (more) {
                dma_setup(nxt, x, w, M, N, K, tm * BM2, tn * BN2, wave, ln);
                issue_xs(nxt, smem, wave, 0);
                issue_ws(nxt, smem, wave, 0);
                if (NT > 1) { issue_xs(nxt, smem, wave, 1); issue_ws(nxt, smem, wave, 1); }
            }
        } else if constexpr (MODE == 6) {
            // Half-step stagger between the two waves of a SIMD (wm = 0 "A", wm = 1 "B"): a K-step is two phases, X1 = the
            // previous step's kk = 1 MFMAs + this step's fragment reads, X2 = this step's kk = 0 MFMAs, one barrier per
            // phase; B runs one phase behind A, so a SIMD always has one wave reading fragments and one only multiplying.
            // Stage t & 1 is free once A (global phase 2t) and B (2t+1) hold step t's fragments; DMA(t+2) is issued by
            // everybody in global phase 2t+2 and awaited at the end of 2t+3, a full K-step later.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (wm) __builtin_amdgcn_s_barrier();              // B idles through global phase 0
            for (int t = 0; t < NT; ++t) {
                // ---- X1(t)
                if (!wm && t >= 1 && t + 1 < NT) { issue_xs(cur, smem, wave, t + 1); issue_ws(cur, smem, wave, t + 1); }
                MXQ_FENCE();
                if (t >= 1) mfma_rows2<0, 1>(acc, wf1, xf1);
                MXQ_FENCE();
                load_frags2(smem, t, 0, wm, wn, fr, fq, wf0, xf0);
                MXQ_FENCE();
                if (t >= 1) mfma_rows2<1, 4>(acc, wf1, xf1);
                MXQ_FENCE();
                load_frags2(smem, t, 1, wm, wn, fr, fq, wf1, xf1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (wm && t >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                // ---- X2(t)
                if (wm && t + 2 < NT) { issue_xs(cur, smem, wave, t + 2); issue_ws(cur, smem, wave, t + 2); }
                MXQ_FENCE();
                mfma_rows2<0, 4>(acc, wf0, xf0);
                if (!wm && t >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(wm && t == NT - 1)) __builtin_amdgcn_s_barrier();
            }
            if (more) {
                dma_setup(nxt, x, w, M, N, K, tm * BM2, tn * BN2, wave, ln);
                issue_xs(nxt, smem, wave, 0);
                issue_ws(nxt, smem, wave, 0);
                if (NT > 1) { issue_xs(nxt, smem, wave, 1); issue_ws(nxt, smem, wave, 1); }
            }
        } else if constexpr (MODE >= 3) {
            // One barrier per K-step (stage (t+1) & 1 is free from the start of step t), but the 8 DMA pieces of step t+1 go
            // out ONE per row of 8 MFMAs, and the two waves of a SIMD (wm = 0 / 1) take "piece, row" and "row, piece" turns:
            // a piece costs its wave ~100+ cycles of issue, during which the partner's MFMAs keep the pipe busy.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (NT > 1) { issue_xs(cur, smem, wave, 1); issue_ws(cur, smem, wave, 1); }
            load_frags2(smem, 0, 0, wm, wn, fr, fq, wf0, xf0);
            load_frags2(smem, 0, 1, wm, wn, fr, fq, wf1, xf1);
            mfma_rows2<0, 4>(acc, wf0, xf0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            steps_staggered<true, (MODE >= 4 ? MODE - 2 : 0)>(smem, cur, wave, NT, wm, wn, fr, fq, acc, wf0, xf0, wf1, xf1);
            if (more) {
                dma_setup(nxt, x, w, M, N, K, tm * BM2, tn * BN2, wave, ln);
                issue_xs(nxt, smem, wave, 0);
                issue_ws(nxt, smem, wave, 0);
            }
        } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // step 0's tiles (and the previous tile's output stores)
        __builtin_amdgcn_s_barrier();
        // step 0: no previous half
        if (NT > 1) {
            issue_xs(cur, smem, wave, 1);
            issue_ws(cur, smem, wave, 1);
        }
        load_frags2(smem, 0, 0, wm, wn, fr, fq, wf0, xf0);
        load_frags2(smem, 0, 1, wm, wn, fr, fq, wf1, xf1);
        mfma_rows2<0, 4>(acc, wf0, xf0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 1; t < NT; ++t) {
            const bool issue = t + 1 < NT;
            const u64t s0 = dbg ? stamp() : 0;
            mfma_rows2<0, 1>(acc, wf1, xf1);
            MXQ_FENCE();
            load_frags2(smem, t, 0, wm, wn, fr, fq, wf0, xf0);
            MXQ_FENCE();
            if (EARLY && issue) issue_xs(cur, smem, wave, t + 1);
            MXQ_FENCE();
            mfma_rows2<1, 2>(acc, wf1, xf1);
            MXQ_FENCE();
            if (issue) { if (EARLY) issue_ws(cur, smem, wave, t + 1); else issue_xs(cur, smem, wave, t + 1); }
            MXQ_FENCE();
            mfma_rows2<2, 4>(acc, wf1, xf1);
            MXQ_FENCE();
            load_frags2(smem, t, 1, wm, wn, fr, fq, wf1, xf1);
            MXQ_FENCE();
            mfma_rows2<0, 2>(acc, wf0, xf0);
            MXQ_FENCE();
            if (!EARLY && issue) issue_ws(cur, smem, wave, t + 1);
            MXQ_FENCE();
            mfma_rows2<2, 4>(acc, wf0, xf0);
            if (dbg) {
                const u64t s1 = stamp();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const u64t s2 = stamp();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const u64t s3 = stamp();
                __builtin_amdgcn_s_barrier();
                const u64t s4 = stamp();
                stw += s1 - s0; stl += s2 - s1; stv += s3 - s2; stb += s4 - s3; stn += 1;
                continue;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // both stages are idle from here on: the next tile's first DMAs fly under this tile's last MFMAs and output
        if (more) {
            dma_setup(nxt, x, w, M, N, K, tm * BM2, tn * BN2, wave, ln);
            issue_xs(nxt, smem, wave, 0);
            issue_ws(nxt, smem, wave, 0);
        }
        }
        mfma_rows2<0, 4>(acc, wf1, xf1);     // (NT-1, kk = 1)
        store_tile2(acc, y, M, N, m0, n0, wm, wn, fr, fq);
        cur = nxt;
    }
    if (dbg && blockIdx.x == 8 && threadIdx.x == 0) {
        u64t rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
        dbg[0] = stamp() - mt0;
        dbg[1] = rt1 - rt0;
    }
    if (dbg && blockIdx.x == 8 && (threadIdx.x & 63) == 0) {
        unsigned long long* d = dbg + 2 + wave * 5;
        d[0] = stw; d[1] = stl; d[2] = stv; d[3] = stb; d[4] = stn;
    }
}

template <int MODE>
int launch(const void* x, const void* w16, void* y, int M, int N, int K, hipStream_t stream, unsigned long long* dbg = nullptr) {
    if ((int64_t)BM2 * K * 2 >= ((int64_t)1 << 32)) return -1;
    hipError_t e = hipFuncSetAttribute((const void*)mxq_dense256_f16_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM2 - 1) / BM2, tiles_n = (N + BN2 - 1) / BN2, tiles = tiles_m * tiles_n;
    const int cus = cu_count() / 8 * 8;
    const int grid = tiles < cus ? tiles : cus;
    mxq_dense256_f16_kernel<MODE><<<grid, THREADS2, SMEM2, stream>>>((const uint16_t*)x, (const uint16_t*)w16, (uint16_t*)y, M, N,
                                                             K, tiles_m, tiles_n, tiles, grid, dbg);
    return (int)hipGetLastError();
}
}   // namespace d256

#ifdef MXQ_PROFILING
extern "C" int mxq_prof_dense256_clock(const void* x, const void* w16, void* y, int M, int N, int K, void* dbg, void* stream_) {
    return d256::launch<1>(x, w16, y, M, N, K, (hipStream_t)stream_, (unsigned long long*)dbg);
}
extern "C" int mxq_prof_dense256_f16(const void* x, const void* w16, void* y, int M, int N, int K, int mode, void* stream_) {
    if (mode == 8) return d256::launch<8>(x, w16, y, M, N, K, (hipStream_t)stream_);
    if (mode == 6) return d256::launch<6>(x, w16, y, M, N, K, (hipStream_t)stream_);
    if (mode == 5) return d256::launch<5>(x, w16, y, M, N, K, (hipStream_t)stream_);
    if (mode == 4) return d256::launch<4>(x, w16, y, M, N, K, (hipStream_t)stream_);
    if (mode == 3) return d256::launch<3>(x, w16, y, M, N, K, (hipStream_t)stream_);
    if (mode == 2) return d256::launch<2>(x, w16, y, M, N, K, (hipStream_t)stream_);
    if (mode == 1) return d256::launch<1>(x, w16, y, M, N, K, (hipStream_t)stream_);
    return d256::launch<0>(x, w16, y, M, N, K, (hipStream_t)stream_);
}


#endif
