// EXPERIMENT RECORD (round 2, second session) -- not built, not part of the product.
// A 256 x 256-tile MFMA kernel for the hoisted-dequant mode (dense fp16 weight), written into csrc/gemm8.hip's anonymous
// namespace (it uses that file's helpers: swz, bufdma16, make_rsrc, tile_of_block, MXQ_FENCE, MXQ_LANE_ID, cu_count) and
// exposed through two profiling entries for tools/ab_gemm.py (variants dense256 / dense256e).  Correct on every shape
// tried (<= 1e-3 against the fp32 product at 4096 and 32768 tokens).  Measurements and verdict:
// profiles/r02_gemmx_experiment.txt ("dense256").
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

template <bool EARLY>
__global__ __launch_bounds__(THREADS2) void mxq_dense256_f16_kernel(const uint16_t* __restrict__ x,
                                                                    const uint16_t* __restrict__ w,
                                                                    uint16_t* __restrict__ y, int M, int N, int K,
                                                                    int tiles_m, int tiles_n, int tiles, int grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = K / BK;
    const int wm = wave >> 2, wn = wave & 3;
    int tm, tn;
    tile_of_block2(blockIdx.x, tiles_m, tiles_n, tm, tn);
    int ln;
    MXQ_LANE_ID(ln);
    Dma2 cur, nxt;
    dma_setup(cur, x, w, M, N, K, tm * BM2, tn * BN2, wave, ln);
    issue_xs(cur, smem, wave, 0);
    issue_ws(cur, smem, wave, 0);
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
            // MFMAs of (t-1, kk = 1) and (t, kk = 0); fragment reads of step t; the DMAs of step t+1 (slot (t+1) & 1 was
            // last read in step t-1), spread behind groups of MFMAs
            const bool issue = t + 1 < NT;
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
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // both stages are idle from here on: the next tile's first DMAs fly under this tile's last MFMAs and output
        if (more) {
            dma_setup(nxt, x, w, M, N, K, tm * BM2, tn * BN2, wave, ln);
            issue_xs(nxt, smem, wave, 0);
            issue_ws(nxt, smem, wave, 0);
        }
        mfma_rows2<0, 4>(acc, wf1, xf1);     // (NT-1, kk = 1)
        store_tile2(acc, y, M, N, m0, n0, wm, wn, fr, fq);
        cur = nxt;
    }
}

template <bool EARLY>
int launch(const void* x, const void* w16, void* y, int M, int N, int K, hipStream_t stream) {
    if ((int64_t)BM2 * K * 2 >= ((int64_t)1 << 32)) return -1;
    hipError_t e = hipFuncSetAttribute((const void*)mxq_dense256_f16_kernel<EARLY>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM2 - 1) / BM2, tiles_n = (N + BN2 - 1) / BN2, tiles = tiles_m * tiles_n;
    const int cus = cu_count() / 8 * 8;
    const int grid = tiles < cus ? tiles : cus;
    mxq_dense256_f16_kernel<EARLY><<<grid, THREADS2, SMEM2, stream>>>((const uint16_t*)x, (const uint16_t*)w16, (uint16_t*)y, M, N,
                                                             K, tiles_m, tiles_n, tiles, grid);
    return (int)hipGetLastError();
}
}   // namespace d256

// profiling entries (inside #ifdef MXQ_PROFILING of gemm8.hip):
// extern "C" int mxq_prof_dense256_f16(...)  { return d256::launch<false>(x, w16, y, M, N, K, stream); }   // DMAs mid-step
// extern "C" int mxq_prof_dense256e_f16(...) { return d256::launch<true>(x, w16, y, M, N, K, stream); }    // DMAs at the head
