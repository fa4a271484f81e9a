// W2/4 x A16 dequant-GEMM, v6 = gemm5 (wave-specialised, split dequant) + hybrid stream-K tail.
//
// gemm5 runs one workgroup per 256x128 output tile and one workgroup per CU, so a launch costs
// ceil(tiles / 256) tile times: Llama's gate/up projection at M = 2048 (8 x 86 = 688 tiles) pays 3
// rounds for 2.69 rounds of work, and any launch with fewer than 256 tiles leaves CUs idle.  Here the
// tiles beyond the last full round ("tail") are not given to workgroups whole: their K-steps are
// dealt evenly to one stream-K workgroup per CU, XCD by XCD (each XCD's 32 units share that XCD's
// tail tiles, so the operands stay in its L2).  A unit's K range covers the end of one tile and the
// start of the next; each piece ("segment") runs gemm5's pipeline on a shifted K window.  A segment
// that does not cover its tile's whole K leaves its fp32 accumulators in a workspace slot and bumps
// a per-(tile, wave) K-step counter; the wave whose bump completes the count sums the slots in unit
// order (its own from registers) -- a fixed order, so the result does not depend on which wave
// finishes -- writes fp16 y and re-zeroes the counter.  Nobody ever waits on another workgroup.
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

// consumer-side dequant of one 2-bit group (16 weights) of chunk t: packed LDS copy -> W16[t & 1]
__device__ __forceinline__ void cons_dequant(char* smem, int t, int d_row, int g) {
    const uint32_t* blk = (const uint32_t*)(smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE + (d_row >> 4) * BP_BLK);
    const int r = d_row & 15;
    uint32_t o[8];
    const uint32_t scw = ((const uint16_t*)blk)[mxq_sc_u16(r)];
    mxq_deq2x16(blk[mxq_c2(g, r)],
                mxq_scale(__uint_as_float(blk[mxq_qq(g)]), __uint_as_float(blk[mxq_qq(g) + 1]), (scw >> (4 * g)) & 15u),
                __uint_as_float(blk[mxq_z2(g, r)]), o);
    char* wt = smem + OFF_W + (t & 1) * W_STAGE;
    *(u32x4*)(wt + swz(d_row, g * 2)) = (u32x4){o[0], o[1], o[2], o[3]};
    *(u32x4*)(wt + swz(d_row, g * 2 + 1)) = (u32x4){o[4], o[5], o[6], o[7]};
}

// One stream-K segment's bookkeeping (all wave-uniform).  Unit u of XCD e owns the K-steps
// [bound(u), bound(u+1)) of that XCD's tail tiles laid end to end (NT steps per tile).
struct SkSeg {
    float* ws;        // partial slots: [unit = 8u+e][2][BM*BN] fp32
    int* cnt;         // K-step counters: [tail tile = 8j+e][N_CONS waves]
    int u, e, units;  // this unit, its XCD, units per XCD
    int S;            // K-steps in one XCD's tail = tail tiles per XCD * NT
    int j;            // tile index inside the XCD's tail
    int first;        // 1: the segment starts at the unit's range start (slot 0), else slot 1
};

// Partial accumulators cross XCDs (one L2 each).  A __threadfence() would make that safe but costs an
// L2-wide write-back + invalidate per call (measured: +140 us on a 200 us launch); instead the slot
// traffic itself is agent-scope: relaxed atomic 8-byte stores / loads (sc1: written through to, and
// read from, the device coherence point), ordered against the counter bump by s_waitcnt vmcnt(0).
typedef unsigned long long u64;
__device__ __forceinline__ void st_agent(float* slot, int f, int lane, f32x4 v) {
    union { f32x4 v4; u64 q[2]; } c;
    c.v4 = v;
    u64* p = (u64*)slot + (f * 2) * 64 + lane;
    __hip_atomic_store(p, c.q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 64, c.q[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 ld_agent(const float* slot, int f, int lane) {
    union { f32x4 v4; u64 q[2]; } c;
    const u64* p = (const u64*)slot + (f * 2) * 64 + lane;
    c.q[0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.q[1] = __hip_atomic_load(p + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return c.v4;
}

__device__ __forceinline__ int sk_bound(int u, int S, int units) { return (int)((uint32_t)u * (uint32_t)S / (uint32_t)units); }

// The same dequant cut into stages of <= 6 VALU ops, so that consumer_step() can place one stage behind
// every group of four MFMAs: a 16x16x32 MFMA keeps the SIMD's vector issue port for 8 of its 16 cycles,
// the other 8 take two VALU ops for free, whereas the whole dequant issued in one piece after the MFMAs
// (what hipcc emits for a plain call: ~45 VALU ops in a row) leaves the matrix pipe idle while both
// consumer waves of a SIMD do it in lockstep.  Arithmetic and rounding are mxq_deq2x16's.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct DeqStage {
    int off_c2, off_sc, off_qq, w_off0, w_off1, sh;   // per-thread constants
    uint32_t d, scw, p01, p23, lut_lo, lut_hi, o[4];
    float z, s;
    f32x2 qq;
};
__device__ __forceinline__ void deq_init(DeqStage& q, int d_row, int g) {
    const int blk = d_row >> 4, r = d_row & 15;
    q.off_c2 = blk * BP_BLK + mxq_c2(g, r) * 4;            // Z2 sits (MXQ_OFF_Z2 - MXQ_OFF_C2) dwords further
    q.off_sc = blk * BP_BLK + mxq_sc_u16(r) * 2;
    q.off_qq = blk * BP_BLK + mxq_qq(g) * 4;
    q.w_off0 = swz(d_row, g * 2);
    q.w_off1 = swz(d_row, g * 2 + 1);
    q.sh = 4 * g;
}
__device__ __forceinline__ void deq_load(DeqStage& q, const char* smem, int t) {
    const char* b = smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE;
    q.d = *(const uint32_t*)(b + q.off_c2);
    q.z = *(const float*)(b + q.off_c2 + (MXQ_OFF_Z2 - MXQ_OFF_C2) * 4);
    q.scw = *(const uint16_t*)(b + q.off_sc);
    q.qq = *(const f32x2*)(b + q.off_qq);
}
__device__ __forceinline__ void deq_scale(DeqStage& q) { q.s = mxq_scale(q.qq[0], q.qq[1], (q.scw >> q.sh) & 15u); }
__host__ __device__ __forceinline__ void deq_pair01(DeqStage& q) { q.p01 = mxq_pack_f16(q.s * (0.0f - q.z), q.s * (1.0f - q.z)); }
__host__ __device__ __forceinline__ void deq_pair23(DeqStage& q) { q.p23 = mxq_pack_f16(q.s * (2.0f - q.z), q.s * (3.0f - q.z)); }
__host__ __device__ __forceinline__ void deq_lut(DeqStage& q) {
    q.lut_lo = MXQ_PERM(q.p23, q.p01, 0x06040200u);
    q.lut_hi = MXQ_PERM(q.p23, q.p01, 0x07050301u);
}
template <int J>
__host__ __device__ __forceinline__ void deq_select(DeqStage& q) {   // elements 4J .. 4J+3 -> o[2(J&1)], o[2(J&1)+1]
    const uint32_t m = (q.d >> (2 * J)) & 0x03030303u;
    const uint32_t lo = MXQ_PERM(0u, q.lut_lo, m), hi = MXQ_PERM(0u, q.lut_hi, m);
    q.o[2 * (J & 1)] = MXQ_PERM(hi, lo, 0x05010400u);
    q.o[2 * (J & 1) + 1] = MXQ_PERM(hi, lo, 0x07030602u);
}
template <int H>
__device__ __forceinline__ void deq_store(const DeqStage& q, char* smem, int t) {   // half H: elements 8H .. 8H+7
    char* wt = smem + OFF_W + (t & 1) * W_STAGE;
    *(u32x4*)(wt + (H ? q.w_off1 : q.w_off0)) = (u32x4){q.o[0], q.o[1], q.o[2], q.o[3]};
}

// One K-step t >= 1 of a consumer wave: MFMAs of (t-1, kk=1) and (t, kk=0), fragment loads of step t, and --
// DEQ -- the staged dequant of this thread's group of chunk t+1, one stage behind every four MFMAs.
#define MXQ_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int ABL>
__device__ __forceinline__ void consumer_step(char* smem, int t, bool deq, int wm, int wn, int fr, int fq,
                                              f32x4 (&acc)[4][4], Frag4& wf0, Frag4& xf0, Frag4& wf1, Frag4& xf1,
                                              DeqStage& q) {
    // `deq` is wave-uniform.  The packed words are read first (LDS returns in order: they are in before the
    // fragments); the arithmetic runs after the step's last MFMA has been issued.
    if (deq) deq_load(q, smem, t + 1);
    MXQ_FENCE();
    mfma_rows<0, 1, ABL>(acc, wf1, xf1);
    MXQ_FENCE();
    if constexpr (!(ABL & 8)) load_frags(smem, t, 0, wm, wn, fr, fq, wf0, xf0);
    MXQ_FENCE();
    mfma_rows<1, 3, ABL>(acc, wf1, xf1);
    MXQ_FENCE();
    if (deq) { deq_scale(q); deq_pair01(q); }
    MXQ_FENCE();
    mfma_rows<3, 4, ABL>(acc, wf1, xf1);
    MXQ_FENCE();
    if (deq) { deq_pair23(q); deq_lut(q); }
    if constexpr (!(ABL & 8)) load_frags(smem, t, 1, wm, wn, fr, fq, wf1, xf1);
    MXQ_FENCE();
    mfma_rows<0, 1, ABL>(acc, wf0, xf0);
    MXQ_FENCE();
    if (deq) deq_select<0>(q);
    MXQ_FENCE();
    mfma_rows<1, 2, ABL>(acc, wf0, xf0);
    MXQ_FENCE();
    if (deq) { deq_select<1>(q); deq_store<0>(q, smem, t + 1); }
    MXQ_FENCE();
    mfma_rows<2, 3, ABL>(acc, wf0, xf0);
    MXQ_FENCE();
    if (deq) deq_select<2>(q);
    MXQ_FENCE();
    mfma_rows<3, 4, ABL>(acc, wf0, xf0);
    MXQ_FENCE();
    if (deq) { deq_select<3>(q); deq_store<1>(q, smem, t + 1); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void store_tile(const f32x4 (&acc)[4][4], uint16_t* __restrict__ y, int M, int N, int m0,
                                           int n0, int wm, int wn, int fr, int fq) {
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

// The same tile through LDS: a lane's accumulators are 4 channels x 16 tokens per fragment, i.e. a direct store
// writes 32 B into each of 16 rows (8 KB apart) per instruction -- 16.8 MB of such pieces cost 13 us at the end
// of a 4096^2 launch (profiling build without the stores: 82 -> 69 us).  Staged in the wave's own 9 KB of the
// (now idle) x ring as [token][channel] fp16 rows of 144 B, the tile leaves as full 128-B lines, 16 B per lane.
// Only for workgroups that run a single whole tile: a stream-K unit's DMA waves may already be filling the
// ring for the next segment.
template <bool NO_GLOBAL = false>
__device__ __forceinline__ void store_tile_staged(const f32x4 (&acc)[4][4], char* smem, uint16_t* __restrict__ y, int M,
                                                  int N, int m0, int n0, int wave, int lane) {
    constexpr int ROW = 144;   // 128 B of channels + 16 B: keeps the b128 reads aligned and spreads the banks
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
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
        if constexpr (NO_GLOBAL) { if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u) y[0] = 1; }   // timing probe: LDS pass only
        else if (m < M && n < N) *(u32x4*)(y + (int64_t)m * N + n) = v;
    }
}

template <int ABL>
__device__ __forceinline__ void consumer(char* smem, int wave, int lane, int NT, uint16_t* __restrict__ y, int M, int N,
                                         int m0, int n0, int NT_tile, const SkSeg& sk, bool lds_free) {
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    const bool has_deq = wave < 6;                      // wave-uniform
    const int d_row = (wave & 1) * 64 + lane, d_g = wave >> 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag4 wf0, xf0, wf1, xf1;

    __builtin_amdgcn_s_barrier();   // prologue barrier 1: x tiles 0,1 and packed blocks 0..2 landed
    if (has_deq) cons_dequant(smem, 0, d_row, d_g);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // prologue barrier 2: W16(0) written

    // step 0: no previous half
    load_frags(smem, 0, 0, wm, wn, fr, fq, wf0, xf0);
    load_frags(smem, 0, 1, wm, wn, fr, fq, wf1, xf1);
    mfma_rows<0, 4, ABL>(acc, wf0, xf0);
    if (has_deq && NT > 1) cons_dequant(smem, 1, d_row, d_g);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // (wf1, xf1) = fragments of (t-1, kk=1), waited for at the end of the previous step.  Steps 1 .. NT-2
    // dequantise chunk t+1 on the way; the last step has nothing left to dequantise.
    DeqStage q;
    deq_init(q, d_row, d_g);
    const bool deq_wave = has_deq && !(ABL & (4 | 32));
    for (int t = 1; t < NT; ++t)
        consumer_step<ABL>(smem, t, deq_wave && t + 1 < NT, wm, wn, fr, fq, acc, wf0, xf0, wf1, xf1, q);
    mfma_rows<0, 4, ABL>(acc, wf1, xf1);   // (NT-1, kk=1)

    if (NT != NT_tile) {
        // partial segment: park the accumulators in this unit's slot and go on; the K-step count is
        // bumped (and the tile possibly finished) in sk_settle() after the unit's last segment, when
        // these stores have long landed
        float* mine = sk.ws + ((int64_t)((sk.u * 8 + sk.e) * 2 + (sk.first ? 0 : 1)) * (BM * BN)) + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) st_agent(mine, i * 4 + j, lane, acc[i][j]);
        return;
    }
    if constexpr (!(ABL & 256)) {   // 256: no output (timing probe)
        if (lds_free) store_tile_staged<(ABL & 512) != 0>(acc, smem, y, M, N, m0, n0, wave, lane);
        else store_tile(acc, y, M, N, m0, n0, wm, wn, fr, fq);
    }
    else {   // keep every accumulator alive without writing the tile
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) y[0] = 1;
    }
}

// The wave that completed a tile's K-step count: sum every contributor's slot in unit order (its own
// included, re-read from the workspace, so the order never depends on who finishes) and write y.
__device__ __forceinline__ void sk_finish(const SkSeg& sk, int j, int NT_tile, int wave, int lane, char* smem,
                                          uint16_t* __restrict__ y, int M, int N, int m0, int n0) {
    const int lo = j * NT_tile, hi = lo + NT_tile;
    int uf = 0;
    while (uf + 1 < sk.units && sk_bound(uf + 1, sk.S, sk.units) <= lo) ++uf;
    f32x4 acc[4][4];
    bool any = false;
    for (int v = uf; v < sk.units && sk_bound(v, sk.S, sk.units) < hi; ++v) {
        const int vb = sk_bound(v, sk.S, sk.units);
        if (sk_bound(v + 1, sk.S, sk.units) <= (vb > lo ? vb : lo)) continue;   // empty range: no slot was written
        const float* src = sk.ws + ((int64_t)((v * 8 + sk.e) * 2 + (vb >= lo ? 0 : 1)) * (BM * BN)) + wave * 4096;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // 8 fragments (16 loads) in flight at a time
            f32x4 p[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) p[i][jj] = ld_agent(src, (h * 2 + i) * 4 + jj, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[h * 2 + i][jj] = any ? acc[h * 2 + i][jj] + p[i][jj] : p[i][jj];
            __builtin_amdgcn_sched_barrier(0);
        }
        any = true;
    }
    if (lane == 0)   // ready for the next launch
        __hip_atomic_store(sk.cnt + (j * 8 + sk.e) * N_CONS + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    store_tile_staged(acc, smem, y, M, N, m0, n0, wave, lane);   // runs after the unit's last segment: LDS is idle
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

template <int ABL = 0>
__device__ __forceinline__ void issue_a(const Prod& c, int t) {
    // producer p fills rows 64p .. 64p+63 of the x slot: DMA i covers rows 64p + 8i .. +7
    char* dst = c.smem + OFF_A + (t % A_SLOTS) * A_STAGE + c.p * 8192;
#pragma unroll
    for (int i = 0; i < ((ABL & 16) ? 4 : 8); ++i) glds16(c.a_src[i] + t * BK, dst + i * 1024);   // ABL 16: half the x DMAs (timing probe)
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

// producer dequant: the 4-bit arm only; thread -> (W row d_row, half d_qp of the 16 four-bit weights)
template <int LAYOUT>
__device__ __forceinline__ void dequant(const Prod& c, int t) {
    const uint32_t* blk = (const uint32_t*)(c.smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE + c.d_blk * BP_BLK);
    uint32_t o[4];
    mxq_deq4x8(blk[mxq_c4(c.d_qp, c.d_r)], c.s4, c.z4, o);
    char* wt = c.smem + OFF_W + (t & 1) * W_STAGE;
    *(u32x4*)(wt + swz(c.d_row, 6 + c.d_qp)) = (u32x4){o[0], o[1], o[2], o[3]};
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
        if constexpr (!(ABL & 1)) issue_a<ABL>(c, t + 2);
        issue_bp<LAYOUT>(c, t + 3);
        if constexpr (!(ABL & (4 | 64))) dequant<LAYOUT>(c, t + 1);
        if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if constexpr (ABL & 16) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
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

// producer side of one segment: K-steps [kt0, kt0 + nsteps) of the tile that block id tile_bid maps to
template <int ABL, int LAYOUT>
__device__ __forceinline__ void produce_segment(char* smem, int wave, int lane, int tid, const uint16_t* __restrict__ x,
                                                const uint32_t* __restrict__ qweight,
                                                const float4* __restrict__ rowmeta, int M, int N, int K, int m0, int n0,
                                                int kt0, int nsteps) {
    const int NT_tile = K / BK;
    Prod c;
    c.smem = smem;
    c.p = wave - N_CONS;
    c.lane = lane;
    c.NT = nsteps;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = c.p * 64 + i * 8 + (lane >> 3);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        c.a_src[i] = x + (int64_t)gm * K + kt0 * BK + (((lane & 7) ^ (row & 7)) << 3);
    }
    constexpr int BLK_DW = LAYOUT == MXQ_LAYOUT_W4ROW ? 128 : 144;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        int rb = (n0 >> 4) + c.p * 2 + b;
        rb = rb < (N >> 4) ? rb : (N >> 4) - 1;
        c.bp_src[b] = (const char*)(qweight + ((int64_t)rb * NT_tile + kt0) * BLK_DW) + lane * 16;
    }
    const int ptid = tid - N_CONS * 64;   // 0..255
    c.d_row = ptid & 127;
    c.d_qp = __builtin_amdgcn_readfirstlane(ptid >> 7);   // wave-uniform: producers 0,1 -> 0; 2,3 -> 1
    c.d_blk = c.d_row >> 4;
    c.d_r = c.d_row & 15;
    {
        int gn = n0 + c.d_row;
        gn = gn < N ? gn : N - 1;
        const float4 m = rowmeta[gn];
        c.s4 = mxq_scale(m.z, m.w, (uint32_t)m.y);
        c.z4 = m.x;
    }
    producer<ABL, LAYOUT>(c);
}

// grid = dp_blocks (one whole tile each, gemm5's schedule) + 8 * units stream-K workgroups
template <int ABL, int LAYOUT>
__global__ __launch_bounds__(THREADS) void mxq_gemm6_f16_kernel(const uint16_t* __restrict__ x,
                                                               const uint32_t* __restrict__ qweight,
                                                               const float4* __restrict__ rowmeta,
                                                               uint16_t* __restrict__ y, int M, int N, int K,
                                                               int tiles_m, int tiles_n, int dp_blocks, int tail,
                                                               int units, float* __restrict__ ws,
                                                               int* __restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NT = K / BK;
    const int bid = blockIdx.x;
    SkSeg sk;
    sk.ws = ws;
    sk.cnt = cnt;
    sk.units = units;
    sk.S = 0;
    sk.u = sk.e = sk.j = sk.first = 0;
    // a data-parallel workgroup is the degenerate unit: one tile, its whole K range
    int base = bid, b0 = 0, b1 = NT;
    if (bid >= dp_blocks) {
        const int s = bid - dp_blocks;
        sk.e = s & 7;
        sk.u = s >> 3;
        base = dp_blocks + sk.e;
        sk.S = ((tail + 7 - sk.e) >> 3) * NT;   // tail tile t belongs to XCD t & 7: the first tail % 8 XCDs hold one more
        b0 = sk_bound(sk.u, sk.S, units);
        b1 = sk_bound(sk.u + 1, sk.S, units);
    }
    // every wave walks the same segment list, so the barrier counts of the two roles stay matched; after
    // a segment's last barrier nobody touches LDS any more, so the next segment may start at once
    // (two copies of the loop, one per wave-uniform role: each role keeps only its own loop invariants live)
    if (wave < N_CONS) {
        __builtin_amdgcn_s_setprio(3);   // MFMA waves win issue arbitration against the DMA / dequant wave of their SIMD
        // a unit has at most two partial segments: the one its range starts in and the one it ends in
        int pj0 = -1, pn0 = 0, pj1 = -1, pn1 = 0;
        for (int pos = b0; pos < b1;) {
            sk.j = pos / NT;
            const int end = b1 < (sk.j + 1) * NT ? b1 : (sk.j + 1) * NT;
            sk.first = pos == b0;
            int tm, tn;
            tile_of_block(base + sk.j * 8, tiles_m, tiles_n, tm, tn);
            // lane id recomputed per segment and made opaque: nothing lane-derived is hoisted (and spilled) across the loop
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            consumer<ABL>(smem, wave, ln, end - pos, y, M, N, tm * BM, tn * BN, NT, sk, bid < dp_blocks);
            if (end - pos != NT) {
                if (sk.first) { pj0 = sk.j; pn0 = end - pos; }
                else { pj1 = sk.j; pn1 = end - pos; }
            }
            pos = end;
        }
        if (pj0 >= 0 || pj1 >= 0) {
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            // every slot store of this wave has reached the coherence point before any count moves
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int old0 = 0, old1 = 0;
            if (ln == 0) {   // both bumps in flight together
                if (pj0 >= 0) old0 = __hip_atomic_fetch_add(cnt + (pj0 * 8 + sk.e) * N_CONS + wave, pn0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pj1 >= 0) old1 = __hip_atomic_fetch_add(cnt + (pj1 * 8 + sk.e) * N_CONS + wave, pn1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            old0 = __builtin_amdgcn_readfirstlane(old0);
            old1 = __builtin_amdgcn_readfirstlane(old1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // compiler ordering only: the slot loads are agent-scope themselves
            if (pj0 >= 0 && old0 + pn0 == NT) {
                int tm, tn;
                tile_of_block(base + pj0 * 8, tiles_m, tiles_n, tm, tn);
                sk_finish(sk, pj0, NT, wave, ln, smem, y, M, N, tm * BM, tn * BN);
            }
            if (pj1 >= 0 && old1 + pn1 == NT) {
                int tm, tn;
                tile_of_block(base + pj1 * 8, tiles_m, tiles_n, tm, tn);
                sk_finish(sk, pj1, NT, wave, ln, smem, y, M, N, tm * BM, tn * BN);
            }
        }
    } else {
        for (int pos = b0; pos < b1;) {
            const int j = pos / NT;
            const int end = b1 < (j + 1) * NT ? b1 : (j + 1) * NT;
            int tm, tn;
            tile_of_block(base + j * 8, tiles_m, tiles_n, tm, tn);
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            produce_segment<ABL, LAYOUT>(smem, wave, ln, wave * 64 + ln, x, qweight, rowmeta, M, N, K, tm * BM, tn * BN,
                                         pos - j * NT, end - pos);
            pos = end;
        }
    }
}

int cu_count() {
    static int cus = 0;   // one device model per process on this platform
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else
            cus = 256;
    }
    return cus;
}

constexpr size_t CNT_BYTES = 64 * 1024;   // K-step counters at the head of the workspace (>= 8*units*N_CONS ints)

template <int ABL, int LAYOUT = MXQ_LAYOUT_MIXED>
static int launch6(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   void* workspace, size_t ws_bytes, bool force, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)mxq_gemm6_f16_kernel<ABL, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    const int NT = K / BK;
    const int cus = cu_count() / 8 * 8, units = cus / 8;
    int dp_blocks = tiles, tail = 0;
    // stream-K tail: only when the workspace is there and a unit gets >= 4 K-steps
    if (workspace && tiles % cus != 0 && units * 8 * N_CONS * sizeof(int) <= CNT_BYTES &&
        ws_bytes >= CNT_BYTES + (size_t)cus * 2 * BM * BN * sizeof(float)) {
        const int t8 = (tiles % cus) / 8;   // tail tiles per XCD (the first tail % 8 XCDs hold one more)
        // Splitting the tail costs ~20 us (every unit parks 128 KB of fp32 partials, the finishers read them
        // back: measured on 256 CUs) and saves the idle share of one tile time, (1 - tail/CUs) * NT K-steps
        // of ~1 us: worth it from ~24 idle K-steps (e.g. M = 512: 60 -> 37 us at 4096^2, 148 -> 63 us at
        // K = 11008; NOT for Llama's gate/up at M = 2048, tail 176/256 and NT = 64, where it measured +-0).
        const bool pays = (int64_t)(cus - tiles % cus) * NT >= (int64_t)24 * cus;
        if ((force || pays) && (int64_t)t8 * NT >= (int64_t)units * 4) {
            tail = tiles % cus;
            dp_blocks = tiles - tail;
        }
    }
    const int grid = dp_blocks + (tail ? cus : 0);
    mxq_gemm6_f16_kernel<ABL, LAYOUT><<<grid, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n,
        dp_blocks, tail, units, (float*)((char*)workspace + CNT_BYTES), (int*)workspace);
    return (int)hipGetLastError();
}

}   // namespace

size_t mxq_gemm6_workspace_bytes() { return CNT_BYTES + (size_t)(cu_count() / 8 * 8) * 2 * BM * BN * sizeof(float); }

int mxq_launch_gemm6_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                         void* workspace, size_t ws_bytes, int force, hipStream_t stream) {
    return launch6<0>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, force != 0, stream);
}

#ifdef MXQ_PROFILING
// Built only into libmxq_hip_prof.so (make prof; tools/): parts of the kernel removed to time the rest.
// WRONG RESULTS by construction -- never part of libmxq_hip.so or of include/mxq_hip.h.
// 1 = no x DMAs, 2 = no MFMA, 4 = no dequant at all, 16 = half the x DMAs,
// 32 = no 2-bit (consumer-side) dequant, 64 = no 4-bit (producer-side) dequant, 256 = no output stores,
// 512 = output staged through LDS but not written
extern "C" int mxq_prof_gemm6_ablate_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N,
                                         int K, int abl, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    switch (abl) {
        case 1: return launch6<1>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 2: return launch6<2>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 4: return launch6<4>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 5: return launch6<5>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 16: return launch6<16>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 32: return launch6<32>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 64: return launch6<64>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 256: return launch6<256>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);   // no y stores
        case 512: return launch6<512>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);   // LDS staging, no global stores
    }
    return -1;   // MXQ_E_SHAPE: not an ablation this build carries
}
#endif   // MXQ_PROFILING
