// EXPERIMENT RECORD (round 3) -- not built, not part of the product.
// dense256.hip's quadrant-phase ping-pong schedule on a 256-token x 128-channel tile (64 x 64 wave tiles, two phases per
// K-tile, ring of three K-tiles), meant for launches whose 256 x 256 tiles do not fill the chip (2048 tokens).  It was
// compiled inside csrc/dense256.hip's anonymous namespace (it uses that file's helpers) and exported as mxq_dense_f16
// variant 3.  Correct (bit-identical to the 256 x 128 kernel on all of test_dense256_*'s shapes, K-tile counts 2, 3, 64,
// 172), and SLOW: at 2048 tokens 87.2 / 236.0 / 214.6 us against 59.8 / 165.1 / 150.9 for the 256 x 128 kernel with
// dedicated DMA waves (gemm8.hip, dense instantiation) and 81.4 / 173.2 / 194.0 for the 256 x 256 kernel; at 4096 tokens
// 164.5 / 448.9 / 405.5 against 115.7 / 321.9 / 295.0 and 107.1 / 285.7 / 257.9  (4096^2, 11008 x 4096, 4096 x 11008;
// gpurun_out/r3c42).  Per 16 MFMAs a wave here issues 8 fragment reads and 3 DMA pieces (12 + 4 in the even phases)
// where the 256 x 256 kernel issues 6 and 2: the reading wave's segment outlasts its partner's MFMA segment, and the
// ping-pong degenerates into taking turns.  The schedule needs the 128 x 64 wave tile.
// ------------------------------------------------------------------------------------------------
// The same schedule on a 256-token x 128-channel tile, for launches whose 256 x 256 tiles would not fill the chip
// (2048 tokens x 4096 channels = 128 of them).  8 waves = 2 groups x 4; wave (g, q) owns token block64 2 g + (q >> 1) and
// channel half q & 1: 64 x 64 = 4 x 4 accumulator tiles.  A K-tile is TWO phases of 16 MFMAs (token halves of 32); units
// of 16 KB: A = the first 32 tokens of the four token block64s, D = their last 32, W = the 128 channels; a ring of
// THREE K-tiles (144 KB).  Even phase 2T reads A(T), W(T) and stages A(T+2), W(T+2); odd phase 2T+1 reads D(T) and stages
// D(T+2); the waits (vmcnt 10 / 8) retire what the NEXT phase reads, three phases after its issue.
// ------------------------------------------------------------------------------------------------
namespace pp128 {
constexpr int BN1 = 128;
constexpr int SMEM1 = 9 * UNIT;   // 144 KB
enum { KA = 0, KW = 1, KD = 2 };
typedef half8 XF[2][2];   // [k half][token block of the 32-token half]
typedef half8 WF[2][4];   // [k half][channel block]

// XCD-aware tile order (speed only): compact 2-D regions per XCD, as gemm8.hip's tile_of_block
__device__ __forceinline__ void tile_of128(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
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
__device__ __forceinline__ void src_of128(Src& s, const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, int M, int N,
                                          int K, int tm, int tn, bool valid) {
    const int m0 = tm * BM, n0 = tn * BN1;
    const int rx = !valid ? 0 : (M - m0 < BM ? M - m0 : BM), rw = !valid ? 0 : (N - n0 < BN1 ? N - n0 : BN1);
    s.xr = make_rsrc(x + (int64_t)(valid ? m0 : 0) * K, (uint32_t)rx * (uint32_t)K * 2u);
    s.wr = make_rsrc(w + (int64_t)(valid ? n0 : 0) * K, (uint32_t)rw * (uint32_t)K * 2u);
}
template <int KIND>
__device__ __forceinline__ void stage1(const Src& s, const uint32_t (&voff)[3][2], char* smem, int slot, int wave, int kt) {
    char* dst = smem + slot * UNIT + wave * 2048;
    const rsrc_t r = KIND == KW ? s.wr : s.xr;
    bufdma16(r, voff[KIND][0], (uint32_t)kt * (BK * 2), dst);
    bufdma16(r, voff[KIND][1], (uint32_t)kt * (BK * 2), dst + 1024);
}
struct Regs1 {
    f32x4 acc[4][4];
    XF x0, x1;
    WF w;
};
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
// K-tile kt of the current tile, in ring position R (compile time); stages K-tile kt + 2 (the next tile's kt + 2 - NT
// once the current one is exhausted; `nxt` is an empty descriptor when there is none) into ring position (R + 2) % 3
template <int R>
__device__ __forceinline__ void ktile(Regs1& Q, char* smem, const Src& cur, const Src& nxt, const uint32_t (&voff)[3][2], int kt,
                                      int NT, int wave, int mb, int nb, int fr, int fq) {
    constexpr int RS = (R + 2) % 3;
    const bool wrap = kt + 2 >= NT;
    const int skt = wrap ? kt + 2 - NT : kt + 2;
    Src s;
    s.xr = wrap ? nxt.xr : cur.xr;
    s.wr = wrap ? nxt.wr : cur.wr;
    const char* ua = smem + (R * 3 + KA) * UNIT;
    const char* uw = smem + (R * 3 + KW) * UNIT;
    const char* ud = smem + (R * 3 + KD) * UNIT;
    // ---- even phase: A, W
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) Q.w[kk][i] = *(const half8*)(uw + swz(nb * 64 + i * 16 + fr, kk * 4 + fq));
    D256_FENCE();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) Q.x0[kk][j] = *(const half8*)(ua + swz(mb * 32 + j * 16 + fr, kk * 4 + fq));
    D256_FENCE();
    stage1<KA>(s, voff, smem, RS * 3 + KA, wave, skt);
    stage1<KW>(s, voff, smem, RS * 3 + KW, wave, skt);
    D256_FENCE();
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // D of this K-tile has landed (read in the odd phase)
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    D256_FENCE();
    __builtin_amdgcn_s_setprio(1);
    mfma_half<0>(Q.acc, Q.w, Q.x0);
    __builtin_amdgcn_s_setprio(0);
    D256_FENCE();
    __builtin_amdgcn_s_barrier();
    // ---- odd phase: D
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) Q.x1[kk][j] = *(const half8*)(ud + swz(mb * 32 + j * 16 + fr, kk * 4 + fq));
    D256_FENCE();
    stage1<KD>(s, voff, smem, RS * 3 + KD, wave, skt);
    D256_FENCE();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // A, W of the next K-tile have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    D256_FENCE();
    __builtin_amdgcn_s_setprio(1);
    mfma_half<1>(Q.acc, Q.w, Q.x1);
    __builtin_amdgcn_s_setprio(0);
    D256_FENCE();
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void store_tile1(const f32x4 (&acc)[4][4], uint16_t* __restrict__ y, int M, int N, int m0, int n0,
                                            int mb, int nb, int fr, int fq) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    const int n = n0 + nb * 64 + fq * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
        const int m = m0 + mb * 64 + j * 16 + fr;
        if (m < M && n < N) {
            uint16_t* dst = y + (int64_t)m * N + n;
            __builtin_nontemporal_store((u32x4){c[0][0], c[0][1], c[1][0], c[1][1]}, (u32x4*)dst);
            __builtin_nontemporal_store((u32x4){c[2][0], c[2][1], c[3][0], c[3][1]}, (u32x4*)(dst + 8));
        }
    }
}

__global__ __launch_bounds__(THREADS) void mxq_dense128pp_f16_kernel(const uint16_t* __restrict__ x,
                                                                     const uint16_t* __restrict__ w,
                                                                     uint16_t* __restrict__ y, int M, int N, int K,
                                                                     int tiles_m, int tiles_n, int tiles, int grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = K / BK;           // >= 2 (launcher)
    const int g = wave >> 2, q = wave & 3;
    const int mb = g * 2 + (q >> 1), nb = q & 1;
    int ln;
    D256_LANE_ID(ln);
    uint32_t voff[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int lr = (wave * 2 + h) * 8 + (ln >> 3);                    // row inside the unit, 0..127
        const uint32_t sw = (uint32_t)((ln & 7) ^ ((ln >> 3) & 7)) << 4;
        const int xa = (lr >> 5) * 64 + (lr & 31);                        // A: first 32 tokens of each token block64
        voff[KA][h] = (uint32_t)xa * (uint32_t)K * 2u + sw;
        voff[KD][h] = (uint32_t)(xa + 32) * (uint32_t)K * 2u + sw;
        voff[KW][h] = (uint32_t)lr * (uint32_t)K * 2u + sw;
    }
    int tm, tn;
    tile_of128(blockIdx.x, tiles_m, tiles_n, tm, tn);
    Src cur, nxt;
    src_of128(cur, x, w, M, N, K, tm, tn, true);
    // prologue: K-tiles 0 and 1 in issue order A, W, D
    stage1<KA>(cur, voff, smem, 0 * 3 + KA, wave, 0);
    stage1<KW>(cur, voff, smem, 0 * 3 + KW, wave, 0);
    stage1<KD>(cur, voff, smem, 0 * 3 + KD, wave, 0);
    stage1<KA>(cur, voff, smem, 1 * 3 + KA, wave, 1);
    stage1<KW>(cur, voff, smem, 1 * 3 + KW, wave, 1);
    stage1<KD>(cur, voff, smem, 1 * 3 + KD, wave, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A(0), W(0)
    __builtin_amdgcn_s_barrier();
    if (g) __builtin_amdgcn_s_barrier();               // group 1 runs one barrier interval behind group 0
    Regs1 Q;
    int ring = 0;                                       // ring position of the current K-tile
    for (int tile = blockIdx.x; tile < tiles; tile += grid) {
        D256_LANE_ID(ln);
        const int fr = ln & 15, fq = ln >> 4;
        const int m0 = tm * BM, n0 = tn * BN1;
        const bool more = tile + grid < tiles;
        if (more) tile_of128(tile + grid, tiles_m, tiles_n, tm, tn);
        src_of128(nxt, x, w, M, N, K, tm, tn, more);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Q.acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < NT; ++kt) {
            if (ring == 0) ktile<0>(Q, smem, cur, nxt, voff, kt, NT, wave, mb, nb, fr, fq);
            else if (ring == 1) ktile<1>(Q, smem, cur, nxt, voff, kt, NT, wave, mb, nb, fr, fq);
            else ktile<2>(Q, smem, cur, nxt, voff, kt, NT, wave, mb, nb, fr, fq);
            ring = ring == 2 ? 0 : ring + 1;
        }
        store_tile1(Q.acc, y, M, N, m0, n0, mb, nb, fr, fq);
        cur = nxt;
    }
    if (!g) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
}   // namespace pp128

// the 256 x 128-tile version of the schedule: any M, N; K >= 128
int mxq_launch_dense128pp_f16(const void* x, const void* w16, void* y, int M, int N, int K, hipStream_t stream) {
    if (K < 2 * BK || (int64_t)BM * K * 2 >= ((int64_t)1 << 32)) return -2;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + pp128::BN1 - 1) / pp128::BN1, tiles = tiles_m * tiles_n;
    const int cus = cu_count8();
    hipError_t e = hipFuncSetAttribute((const void*)pp128::mxq_dense128pp_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       pp128::SMEM1);
    if (e != hipSuccess) return (int)e;
    const int grid = tiles < cus ? tiles : cus;
    pp128::mxq_dense128pp_f16_kernel<<<grid, THREADS, pp128::SMEM1, stream>>>((const uint16_t*)x, (const uint16_t*)w16, (uint16_t*)y,
                                                                              M, N, K, tiles_m, tiles_n, tiles, grid);
    return (int)hipGetLastError();
}

