// W2/4 x A16 dequant-GEMM for prefill (any token count > 32): the activations never touch LDS.
//
// Counterpart of the reference's (never built) AWQ tensor-core GEMM
// mxq_quant/cuda_kernel/csrc/quantization/gemm_cuda_gen.cu:28-218 and of the implicit nn.Linear on the
// fake-quant weight (mxq_quant/main.py:85); arithmetic contract x16 . fp16(scale * (q - zero))^T of
// lib/quantizer.py:19-20 + mxqgpt.py:448, fp32 accumulation.
//
// What bounded the previous structure (8 MFMA waves of 64 x 64 + 4 dequant waves, x tile through a 3-slot LDS ring
// filled by LDS-DMA) was the SIMDs' vector-issue port: per K-step and SIMD 64 MFMAs hold it 8 cycles each (512), the
// dequant wave's ~90 VALU ops 4 each (360), and every 1-KiB LDS-DMA piece costs its issuer 60-185 cycles
// (MI355X_MICROARCH.md, cycle constants) -- 10 pieces per SIMD and step, 8 of them for x.  ~1450 cycles per step
// against the MFMA pipe's 1024.  So:
//   * waves 0-7 "MFMA waves", two per SIMD, each 32 tokens x ALL 128 channels of the tile (8 x 2 accumulator
//     fragments = 64 VGPRs): the x operand of v_mfma_f32_16x16x32_f16 (8 consecutive k of one token per lane) is
//     exactly a 16-byte global load, so the x fragments come STRAIGHT from global memory / L2 into registers, one
//     K-step ahead (4 buffer_load_dwordx4 per wave and step, rows beyond M zero through the descriptor's range
//     check): no x tile in LDS, no x DMA pieces, no x fragment reads (-64 KiB of LDS reads and -32 KiB of LDS writes
//     per step), and no token is fetched by two waves.  The W operand (shared by the 8 waves) still comes from the
//     fp16 W16 tile in LDS: 16 ds_read_b128 per wave and step, 4 at a time, each issued one 8-MFMA phase before use
//     (the SIMD's other MFMA wave fills the pipe meanwhile).
//   * waves 8-11 "dequant waves" (one per SIMD): unchanged -- the packed blocks' LDS-DMA (2 pieces per wave and
//     step, 4-slot ring) and the whole dequant of chunk t+1 into the W16 double buffer, at raised issue priority,
//     scalar fp32 ops (no SLP packing: Makefile).
//   * one raw s_barrier per K-step (W16 hand-over); the MFMAs of a step's last phase run behind the barrier, so the
//     first W reads of the next step are covered by them.  50 KiB of LDS.  D^T = W . x^T: a lane owns 4 consecutive
//     channels of a token; the output leaves without an LDS round trip (store_tile_xpose).
//
// Grid = min(tiles, CUs) persistent workgroups dealing whole tiles round-robin + (with a workspace) one stream-K
// workgroup per CU for the tiles beyond the last full round ("tail"): their K-steps are dealt evenly, XCD by XCD
// (tail tile t belongs to XCD t & 7; an XCD's 32 units share its tail tiles so the operands stay in that L2).  A
// unit's K range covers the end of one tile and the start of the next; each piece ("segment") runs the same pipeline
// on a shifted K window.  A segment that does not cover its tile's whole K leaves its fp32 accumulators in a
// workspace slot and, after the unit's last segment, bumps a per-(tile, wave) K-step counter; the wave whose bump
// completes the count sums the slots in unit order (its own re-read from the slot) -- a fixed order, so the
// result does not depend on which wave finishes -- writes fp16 y and re-zeroes the counter.  Nobody ever waits on
// another workgroup.  Slots and counters cross XCDs (one L2 each): slot traffic is agent-scope relaxed atomics
// (global_store / load ... sc1), ordered against the counter bump by s_waitcnt vmcnt(0).
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int N_MMA = 8, N_DEQ = 4, THREADS = (N_MMA + N_DEQ) * 64;
constexpr int TB = BM / 16 / N_MMA;              // 16-token blocks per MFMA wave (2: a wave owns 32 tokens x all 128 channels)
constexpr int A_STAGE = BM * BK * 2, A_SLOTS = 3;
constexpr int BP_BLK = MXQ_BLK_BYTES;            // 576 B: stride of 144 dwords keeps blocks on distinct banks
constexpr int BP_STAGE = (BN / 16) * BP_BLK, BP_SLOTS = 4;
constexpr int W_STAGE = BN * BK * 2;
constexpr int OFF_A = 0;
constexpr int OFF_BP = OFF_A + A_SLOTS * A_STAGE;
constexpr int OFF_W = (OFF_BP + BP_SLOTS * BP_STAGE + 255) / 256 * 256;
constexpr int SMEM_BYTES = OFF_W + 2 * W_STAGE;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
// hoisted-dequant mode: the packed ring and the W16 double buffer make room for a 3-slot ring of fp16 weight tiles
constexpr int OFF_WD = OFF_BP, WD_SLOTS = 3;
static_assert(OFF_WD + WD_SLOTS * W_STAGE <= SMEM_BYTES, "dense weight ring fits the same LDS");
constexpr int LAYOUT_DENSE16 = 100;   // internal: qweight is a dense fp16 [N, K] matrix (never part of the C ABI)

// profiling-only switches (template parameter ABL; product build = 0): no x loads / no MFMA / no dequant / no stores
constexpr int ABL_NO_MFMA = 2, ABL_NO_DEQ = 4, ABL_NO_XLD = 1, ABL_NO_STORE = 256;
// scheduling experiments (correct results): issue priorities of the two roles
constexpr int EXP_NO_PRIO = 1024, EXP_MMA_PRIO = 2048;

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

// LDS-DMA through a buffer descriptor: 16 B per lane, LDS destination = wave-uniform base + 16 * lane;
// global source = descriptor base + voff (per lane) + soff (scalar); out-of-range sources deliver zeros
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ void bufdma16(rsrc_t rsrc, uint32_t voff, uint32_t soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

// XCD-aware tile order (speed only): tiles are dealt to the 8 XCDs as compact 2-D blocks (4 x 2 regions)
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
// stream-K bookkeeping (protocol: header)
// ------------------------------------------------------------------------------------------------
struct SkSeg {
    float* ws;        // partial slots: [unit = 8u+e][2][BM*BN] fp32
    int* cnt;         // K-step counters: [tail tile = 8j+e][N_MMA waves]
    int u, e, units;  // this unit, its XCD, units per XCD
    int S;            // K-steps in one XCD's tail = tail tiles per XCD * NT
    int j;            // tile index inside the XCD's tail
    int first;        // 1: the segment starts at the unit's range start (slot 0), else slot 1
};
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

// ------------------------------------------------------------------------------------------------
// MFMA waves
// ------------------------------------------------------------------------------------------------
typedef half8 Frag4[4];

#define MXQ_FENCE() __builtin_amdgcn_sched_barrier(0)

// W fragments of channel blocks 4h .. 4h+3, k slice kk, of chunk t.  DENSE: the weight tile is a 3-slot ring of fp16
// tiles filled by LDS-DMA (hoisted-dequant mode, below) instead of the double buffer the dequant waves write.
template <bool DENSE, int kk, int h>
__device__ __forceinline__ void load_w(const char* smem, int t, int fr, int fq, Frag4& wf) {
    const char* w_base = DENSE ? smem + OFF_WD + (t % WD_SLOTS) * W_STAGE : smem + OFF_W + (t & 1) * W_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *(const half8*)(w_base + swz((h * 4 + i) * 16 + fr, kk * 4 + fq));
}

// 4 x TB MFMAs: channel blocks 4h .. 4h+3 x the wave's token blocks, one k slice
template <int ABL, int h, int I0 = 0, int I1 = 4>
__device__ __forceinline__ void mfma_phase(f32x4 (&acc)[8][TB], const Frag4& wf, const half8 (&xf)[TB]) {
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            if constexpr (ABL & ABL_NO_MFMA) asm volatile("" ::"v"(wf[i]), "v"(xf[j]));
            else acc[h * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[h * 4 + i][j], 0, 0, 0);
        }
}

// The wave's x rows (tokens 32 w .. 32 w + 31 of the tile) live in a wave-PRIVATE part of a 3-slot LDS ring: the
// wave fills it itself by LDS-DMA (4 pieces of 8 full 128-B rows per K-step, two steps ahead; lane -> (row 8 i + lane / 8,
// 16-B chunk lane % 8), chunk XOR-swizzled against the row so that the fragment reads are conflict-free) and is the
// only reader, so the x path needs no barrier -- only the wave's own counted vmcnt wait.  (Loading the x fragments
// straight from global memory into registers was tried first: 30 us of 72 at 2048 x 4096^2 against 6 us for the DMA
// ring -- kept as abtmp / DESIGN.md record only.)
struct XDma {
    rsrc_t rsrc;         // x rows m0 .. of this tile (range-checked: rows beyond M read as zeros)
    uint32_t voff[4];    // per piece: the lane's row offset + swizzled chunk; 0x80000000 = nothing to copy
    uint32_t k0;         // byte offset of the segment's first K-step inside a row
};
__device__ __forceinline__ void xdma_setup(XDma& xd, const uint16_t* __restrict__ x, int M, int K, int m0, int kt0,
                                           int wave, int lane) {
    const int rows = M - m0 < BM ? M - m0 : BM;                       // live rows of this tile
    xd.rsrc = make_rsrc(x + (int64_t)m0 * K, (uint32_t)rows * (uint32_t)K * 2u);
    xd.k0 = (uint32_t)kt0 * (BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        xd.voff[i] = (uint32_t)row * (uint32_t)K * 2u + ((((uint32_t)lane & 7u) ^ ((uint32_t)row & 7u)) << 4);
    }
}
__device__ __forceinline__ void xdma_none(XDma& xd, const uint16_t* __restrict__ x) {   // every piece out of range: zeros, no traffic
    xd.rsrc = make_rsrc(x, 0u);
    xd.k0 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) xd.voff[i] = 0x80000000u;
}
// pieces I0 .. I1-1 of step t of descriptor d into ring slot g % 3 (g: the wave's running step count across tiles)
template <int ABL, int I0, int I1>
__device__ __forceinline__ void issue_x(const XDma& d, char* smem, int wave, int g, int t) {
    if constexpr (ABL & ABL_NO_XLD) return;
    char* dst = smem + OFF_A + (g % A_SLOTS) * A_STAGE + wave * 4096;
#pragma unroll
    for (int i = I0; i < I1; ++i) bufdma16(d.rsrc, d.voff[i], d.k0 + (uint32_t)t * (BK * 2), dst + i * 1024);
}
// x fragments of k slice kk from slot g % 3: token 32 w + 16 j + fr, k = 32 kk + 8 fq .. + 7
template <int kk>
__device__ __forceinline__ void load_x(const char* smem, int wave, int g, int fr, int fq, half8 (&xf)[TB]) {
    const char* a_base = smem + OFF_A + (g % A_SLOTS) * A_STAGE;
#pragma unroll
    for (int j = 0; j < TB; ++j) xf[j] = *(const half8*)(a_base + swz(wave * (TB * 16) + j * 16 + fr, kk * 4 + fq));
}

// One K-step t (the wave's g-th step overall).  On entry the wave's x(t) has landed in slot g % 3, and -- unless
// FIRST -- wb = W(t-1, k slice 1, channel blocks 4-7) and x1 = x(t-1, k slice 1), whose MFMAs run first, behind the
// previous barrier, so that this step's first reads are covered.  The DMA of the step two ahead (descriptor xn, step
// tn: normally this segment's t + 2; near a tile's end the NEXT tile's step 0 / 1, or nothing) is issued in two
// halves between the phases.  Every fragment read is issued one phase before its use, into registers whose MFMAs
// have just been issued.
template <int ABL, bool FIRST, bool DENSE>
__device__ __forceinline__ void mma_step(char* smem, int wave, int t, int g, int fr, int fq, f32x4 (&acc)[8][TB],
                                         Frag4& wa, Frag4& wb, half8 (&x0)[TB], half8 (&x1)[TB], const XDma& xn,
                                         int tn) {
    // (each DMA piece sits in the MIDDLE of a phase's MFMAs, away from the fragment reads: a piece issued next to
    // ds_reads costs its wave 2-3 x the issue time of one issued among bare MFMAs)
    load_w<DENSE, 0, 0>(smem, t, fr, fq, wa);
    load_x<0>(smem, wave, g, fr, fq, x0);
    MXQ_FENCE();
    if constexpr (!FIRST) {
        mfma_phase<ABL, 1, 0, 2>(acc, wb, x1);
        MXQ_FENCE();
        issue_x<ABL, 0, 1>(xn, smem, wave, g + 2, tn);     // slot (g+2) % 3 held x(g-1): read out one step ago
        MXQ_FENCE();
        mfma_phase<ABL, 1, 2, 4>(acc, wb, x1);
    } else {
        issue_x<ABL, 0, 1>(xn, smem, wave, g + 2, tn);
    }
    MXQ_FENCE();
    load_w<DENSE, 0, 1>(smem, t, fr, fq, wb);
    load_x<1>(smem, wave, g, fr, fq, x1);
    MXQ_FENCE();
    mfma_phase<ABL, 0, 0, 2>(acc, wa, x0);
    MXQ_FENCE();
    issue_x<ABL, 1, 2>(xn, smem, wave, g + 2, tn);
    MXQ_FENCE();
    mfma_phase<ABL, 0, 2, 4>(acc, wa, x0);
    MXQ_FENCE();
    load_w<DENSE, 1, 0>(smem, t, fr, fq, wa);
    MXQ_FENCE();
    mfma_phase<ABL, 1, 0, 2>(acc, wb, x0);
    MXQ_FENCE();
    issue_x<ABL, 2, 3>(xn, smem, wave, g + 2, tn);
    MXQ_FENCE();
    mfma_phase<ABL, 1, 2, 4>(acc, wb, x0);
    MXQ_FENCE();
    load_w<DENSE, 1, 1>(smem, t, fr, fq, wb);
    MXQ_FENCE();
    mfma_phase<ABL, 0, 0, 2>(acc, wa, x1);
    MXQ_FENCE();
    issue_x<ABL, 3, 4>(xn, smem, wave, g + 2, tn);
    MXQ_FENCE();
    mfma_phase<ABL, 0, 2, 4>(acc, wa, x1);
    // x(g+1) -- the previous step's 4 pieces -- has landed; this step's 4 stay in flight.  Every read of this step's
    // W tile has returned before the barrier hands the buffer back to the dequant waves.
    if constexpr (ABL & ABL_NO_XLD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// A lane's accumulators are 4 channels (8 B as fp16) of channel block i for each of 4 token blocks j; the 4 lanes
// {fr, fr+16, fr+32, fr+48} hold one token's 4 x 16 channels of a 64-channel half as a 4 x 4 grid of 8-byte cells
// (block i, quarter fq).  Two butterfly stages of lane swaps (v_permlane32_swap: lanes +-32 <-> blocks +-2;
// v_permlane16_swap: lanes +-16 <-> blocks +-1) transpose the grid, after which lane fq owns block fq whole: 32
// contiguous bytes = two 16-byte stores.  No LDS involved.
__device__ __forceinline__ void store_tile_xpose(const f32x4 (&acc)[8][TB], uint16_t* __restrict__ y, int M, int N,
                                                 int m0, int n0, int wm, int fr, int fq) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + h * 64 + fq * 16;
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            uint32_t c[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c[i][0] = mxq_pack_f16(acc[h * 4 + i][j][0], acc[h * 4 + i][j][1]);
                c[i][1] = mxq_pack_f16(acc[h * 4 + i][j][2], acc[h * 4 + i][j][3]);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                u32x2v r;
                r = __builtin_amdgcn_permlane32_swap(c[0][d], c[2][d], false, false); c[0][d] = r[0]; c[2][d] = r[1];
                r = __builtin_amdgcn_permlane32_swap(c[1][d], c[3][d], false, false); c[1][d] = r[0]; c[3][d] = r[1];
                r = __builtin_amdgcn_permlane16_swap(c[0][d], c[1][d], false, false); c[0][d] = r[0]; c[1][d] = r[1];
                r = __builtin_amdgcn_permlane16_swap(c[2][d], c[3][d], false, false); c[2][d] = r[0]; c[3][d] = r[1];
            }
            const int m = m0 + wm * (TB * 16) + j * 16 + fr;
            if (m < M && n < N) {
                uint16_t* dst = y + (int64_t)m * N + n;
                *(u32x4*)dst = (u32x4){c[0][0], c[0][1], c[1][0], c[1][1]};
                *(u32x4*)(dst + 8) = (u32x4){c[2][0], c[2][1], c[3][0], c[3][1]};
            }
        }
    }
}

// One segment = NT K-steps of one tile on an MFMA wave; g0: the wave's running step count at its start (ring slot
// phase).  pre: x(0) and x(1) are already on their way (issued during the previous tile's last two steps).  nxt: what
// to copy during the LAST two steps -- the next tile's steps 0 and 1 (persistent loop) or nothing.
template <int ABL, bool DENSE>
__device__ __forceinline__ void mma_segment(char* smem, int wave, int lane, int NT, int g0, const XDma& xd, const XDma& nxt,
                                            bool pre, uint16_t* __restrict__ y, int M, int N, int m0, int n0,
                                            int NT_tile, const SkSeg& sk) {
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[8][TB];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag4 wa, wb;
    half8 x0[TB], x1[TB];
    if (!pre) {
        issue_x<ABL, 0, 4>(xd, smem, wave, g0, 0);
        issue_x<ABL, 0, 4>(NT > 1 ? xd : nxt, smem, wave, g0 + 1, NT > 1 ? 1 : 0);
    }
    // x(0) landed (x(1) may still fly); behind a previous tile its output stores sit in between: a full wait
    if (!pre) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // prologue barrier 1: packed blocks 0..3 landed (dequant waves)
    __builtin_amdgcn_s_barrier();   // prologue barrier 2: W16(0) written by the dequant waves

    // the step two ahead of step t: this segment's t + 2, or step t + 2 - NT of nxt (a select, not a branch)
    auto ahead = [&](int t, XDma& d) -> int {
        const bool over = t + 2 >= NT;
        d.rsrc = over ? nxt.rsrc : xd.rsrc;
        d.k0 = over ? nxt.k0 : xd.k0;
#pragma unroll
        for (int i = 0; i < 4; ++i) d.voff[i] = over ? nxt.voff[i] : xd.voff[i];
        return over ? t + 2 - NT : t + 2;
    };
    XDma d;
    int tn = ahead(0, d);
    mma_step<ABL, true, DENSE>(smem, wave, 0, g0, fr, fq, acc, wa, wb, x0, x1, d, tn);
    for (int t = 1; t < NT; ++t) {
        tn = ahead(t, d);
        mma_step<ABL, false, DENSE>(smem, wave, t, g0 + t, fr, fq, acc, wa, wb, x0, x1, d, tn);
    }
    mfma_phase<ABL, 1>(acc, wb, x1);   // (NT-1, k slice 1, channel blocks 4-7)

    if (NT != NT_tile) {
        // partial segment: park the accumulators in this unit's slot; counted in after the unit's last segment
        float* mine = sk.ws + ((int64_t)((sk.u * 8 + sk.e) * 2 + (sk.first ? 0 : 1)) * (BM * BN)) + wave * (BM * BN / N_MMA);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) st_agent(mine, i * TB + j, lane, acc[i][j]);
        return;
    }
    if constexpr (!(ABL & ABL_NO_STORE)) {
        store_tile_xpose(acc, y, M, N, m0, n0, wave, fr, fq);
    } else {   // keep every accumulator alive without writing the tile
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) s_ += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (s_ == 123.456f) y[0] = 1;
    }
}

// The wave that completed a tile's K-step count: sum every contributor's slot in unit order and write y.
__device__ __forceinline__ void sk_finish(const SkSeg& sk, int j, int NT_tile, int wave, int lane,
                                          uint16_t* __restrict__ y, int M, int N, int m0, int n0) {
    const int lo = j * NT_tile, hi = lo + NT_tile;
    int uf = 0;
    while (uf + 1 < sk.units && sk_bound(uf + 1, sk.S, sk.units) <= lo) ++uf;
    f32x4 acc[8][TB];
    bool any = false;
    for (int v = uf; v < sk.units && sk_bound(v, sk.S, sk.units) < hi; ++v) {
        const int vb = sk_bound(v, sk.S, sk.units);
        if (sk_bound(v + 1, sk.S, sk.units) <= (vb > lo ? vb : lo)) continue;   // empty range: no slot was written
        const float* src = sk.ws + ((int64_t)((v * 8 + sk.e) * 2 + (vb >= lo ? 0 : 1)) * (BM * BN)) + wave * (BM * BN / N_MMA);
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // 8 fragments (16 loads) in flight at a time
            f32x4 p[4][TB];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < TB; ++jj) p[i][jj] = ld_agent(src, (h * 4 + i) * TB + jj, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < TB; ++jj) acc[h * 4 + i][jj] = any ? acc[h * 4 + i][jj] + p[i][jj] : p[i][jj];
            __builtin_amdgcn_sched_barrier(0);
        }
        any = true;
    }
    if (lane == 0)   // ready for the next launch
        __hip_atomic_store(sk.cnt + (j * 8 + sk.e) * N_MMA + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    store_tile_xpose(acc, y, M, N, m0, n0, wave, lane & 15, lane >> 4);
}

// ------------------------------------------------------------------------------------------------
// dequant waves
// ------------------------------------------------------------------------------------------------
struct Deq {
    char* smem;
    rsrc_t rsrc;          // this tile's 8 row-blocks of packed weights (range-checked at the N edge)
    uint32_t voff[2];     // lane's 16-B piece inside row-block 2d + b of the tile (the range check is on this offset:
                          // a row-block beyond the weight's last one reads as zeros, whatever the K offset)
    uint32_t k0;          // byte offset of the segment's first K-step inside a row-block's run
    int d, lane, NT;
    int row, r, h;        // W row of this thread (0..127), its row inside the block, column half (wave-uniform)
    int off_blk;          // byte offset of the row's block inside a packed slot
    float s4, z4;
    float4 rm;            // the row's 4-bit-arm parameters as loaded (rowmeta)
};

template <int LAYOUT>
__device__ __forceinline__ void issue_bp(const Deq& c, int t) {
    // dequant wave d copies packed blocks 2d, 2d+1 (rows 32d .. 32d+31), 36 lanes each (32 for W4ROW);
    // the LDS stride stays 576 B for every layout (bank-conflict-free block spacing)
    constexpr int BYTES = LAYOUT == MXQ_LAYOUT_W4ROW ? 512 : LAYOUT == MXQ_LAYOUT_MIXEDC ? MXQC_BLK_BYTES : MXQ_BLK_BYTES;
    char* dst = c.smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE + c.d * 2 * BP_BLK;
    if (c.lane < BYTES / 16) {
        const uint32_t so = c.k0 + (uint32_t)t * BYTES;
        bufdma16(c.rsrc, c.voff[0], so, dst);
        bufdma16(c.rsrc, c.voff[1], so, dst + BP_BLK);
    }
}

__device__ __forceinline__ void put8(char* wt, int row, int slot, const uint32_t* o) {
    *(u32x4*)(wt + swz(row, slot)) = (u32x4){o[0], o[1], o[2], o[3]};
}

// The packed words one thread needs for one chunk, read from the LDS copy ONE K-step before they are used, so that
// the dequant arithmetic never waits for an LDS read.  h = 0: 2-bit groups 0, 1 (columns 0..31); h = 1: group 2 and
// the 4-bit quarter (columns 32..63).  W2G16: h = 0 groups 0, 1; h = 1 groups 2, 3.  W4ROW: 4 code words each.
struct Pk {
    uint32_t c[4];    // code words
    uint32_t z[2];    // 2-bit zero-points (fp32 bits)
    uint32_t scw;     // the row's scale codes
    f32x2 qq[2];      // (qs, qz) of the thread's 2-bit groups
};

template <int LAYOUT>
__device__ __forceinline__ void load_pk(const Deq& c, int t, Pk& k) {
    const uint32_t* blk = (const uint32_t*)(c.smem + OFF_BP + (t % BP_SLOTS) * BP_STAGE + c.off_blk);
    if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
#pragma unroll
        for (int i = 0; i < 4; ++i) k.c[i] = blk[mxq_w4_c4(c.h * 2 + (i >> 1), i & 1, c.r)];
        return;
    }
    constexpr bool MIXED = LAYOUT == MXQ_LAYOUT_MIXED || LAYOUT == MXQ_LAYOUT_MIXEDC, COMPACT = LAYOUT == MXQ_LAYOUT_MIXEDC;
    k.scw = ((const uint16_t*)blk)[COMPACT ? mxqc_sc_u16(c.r) : mxq_sc_u16(c.r)];
    const int g0 = c.h * 2;   // first 2-bit group of this thread
    auto zero_of = [&](int g) -> uint32_t {   // fp32 bits of the group's zero-point (compact: widened from fp16 here)
        if constexpr (COMPACT) {
            const uint16_t zh = ((const uint16_t*)blk)[mxqc_z2_u16(0, c.r) + g * 16];
            return __float_as_uint((float)__builtin_bit_cast(_Float16, zh));
        } else {
            return blk[(MIXED ? mxq_z2(0, c.r) : mxq_w2_z2(0, c.r)) + g * 16];
        }
    };
    constexpr int QQ0 = COMPACT ? MXQC_OFF_QQ : MXQ_OFF_QQ;
    k.c[0] = blk[(MIXED ? mxq_c2(0, c.r) : mxq_w2_c2(0, c.r)) + g0 * 16];
    k.z[0] = zero_of(g0);
    k.qq[0] = *(const f32x2*)(blk + QQ0 + g0 * 2);
    if (LAYOUT == MXQ_LAYOUT_W2G16 || c.h == 0) {
        k.c[1] = blk[(MIXED ? mxq_c2(1, c.r) : mxq_w2_c2(1, c.r)) + g0 * 16];
        k.z[1] = zero_of(g0 + 1);
        k.qq[1] = *(const f32x2*)(blk + QQ0 + g0 * 2 + 2);
    } else {
        k.c[2] = blk[mxq_c4(0, c.r)];
        k.c[3] = blk[mxq_c4(1, c.r)];
    }
}

// chunk t: preloaded packed words -> fp16 W16[t & 1]
template <int LAYOUT>
__device__ __forceinline__ void dequant_pk(const Deq& c, int t, const Pk& k) {
    char* wt = c.smem + OFF_W + (t & 1) * W_STAGE;
    uint32_t o[8];
    if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            mxq_deq4x8(k.c[2 * q], c.s4, c.z4, o);
            mxq_deq4x8(k.c[2 * q + 1], c.s4, c.z4, o + 4);
            put8(wt, c.row, (c.h * 2 + q) * 2, o);
            put8(wt, c.row, (c.h * 2 + q) * 2 + 1, o + 4);
        }
        return;
    }
    const int g0 = c.h * 2;
    mxq_deq2x16(k.c[0], mxq_scale(k.qq[0][0], k.qq[0][1], (k.scw >> (4 * g0)) & 15u), __uint_as_float(k.z[0]), o);
    put8(wt, c.row, g0 * 2, o);
    put8(wt, c.row, g0 * 2 + 1, o + 4);
    if (LAYOUT == MXQ_LAYOUT_W2G16 || c.h == 0) {
        mxq_deq2x16(k.c[1], mxq_scale(k.qq[1][0], k.qq[1][1], (k.scw >> (4 * g0 + 4)) & 15u), __uint_as_float(k.z[1]), o);
        put8(wt, c.row, g0 * 2 + 2, o);
        put8(wt, c.row, g0 * 2 + 3, o + 4);
    } else {
        mxq_deq4x8(k.c[2], c.s4, c.z4, o);
        mxq_deq4x8(k.c[3], c.s4, c.z4, o + 4);
        put8(wt, c.row, 6, o);
        put8(wt, c.row, 7, o + 4);
    }
}

template <int LAYOUT>
__device__ __forceinline__ void deq_setup(Deq& c, char* smem, int wave, int lane, const uint32_t* __restrict__ qweight,
                                          int N, int K, int n0, int kt0, int nsteps) {
    constexpr int BLK_B = LAYOUT == MXQ_LAYOUT_W4ROW ? 512 : LAYOUT == MXQ_LAYOUT_MIXEDC ? MXQC_BLK_BYTES : MXQ_BLK_BYTES;
    const int NT_tile = K / BK;
    c.smem = smem;
    c.d = wave - N_MMA;
    c.lane = lane;
    c.NT = nsteps;
    const int rb0 = n0 >> 4, rbs = (N >> 4) - rb0 < BN / 16 ? (N >> 4) - rb0 : BN / 16;   // live row-blocks
    const uint32_t blk_stride = (uint32_t)NT_tile * BLK_B;   // bytes between consecutive row-blocks
    c.rsrc = make_rsrc((const char*)qweight + (int64_t)rb0 * NT_tile * BLK_B, (uint32_t)rbs * blk_stride);
    c.voff[0] = (uint32_t)lane * 16u + (uint32_t)(c.d * 2) * blk_stride;
    c.voff[1] = c.voff[0] + blk_stride;
    c.k0 = (uint32_t)kt0 * BLK_B;
    const int dt = c.d * 64 + lane;   // 0..255
    c.row = dt & 127;
    c.h = __builtin_amdgcn_readfirstlane(dt >> 7);   // wave-uniform: dequant waves 0,1 -> 0; 2,3 -> 1
    c.r = c.row & 15;
    c.off_blk = (c.row >> 4) * BP_BLK;
}
// a segment's prologue DMAs (packed blocks of its first BP_SLOTS K-steps: needs the whole packed ring idle) and
// the row's 4-bit-arm parameters; the values are first used behind prologue barrier 1
template <int LAYOUT>
__device__ __forceinline__ void deq_prologue_issue(Deq& c, const float4* __restrict__ rowmeta, int N, int n0) {
    for (int t = 0; t < BP_SLOTS && t < c.NT; ++t) issue_bp<LAYOUT>(c, t);
    int gn = n0 + c.row;
    gn = gn < N ? gn : N - 1;
    c.rm = rowmeta[gn];
}

// One segment on the dequant waves; pre / next as in mma_segment.
template <int ABL, int LAYOUT, class Next>
__device__ __forceinline__ void deq_segment(Deq& c, int wave, int lane, const float4* __restrict__ rowmeta, int N,
                                            int n0, bool pre, Next&& next) {
    if (!pre) deq_prologue_issue<LAYOUT>(c, rowmeta, N, n0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    c.s4 = mxq_scale(c.rm.z, c.rm.w, (uint32_t)c.rm.y);
    c.z4 = c.rm.x;
    Pk cur = {}, nxt = {};
    if constexpr (!(ABL & ABL_NO_DEQ)) {
        load_pk<LAYOUT>(c, 0, cur);
        dequant_pk<LAYOUT>(c, 0, cur);
        if (c.NT > 1) load_pk<LAYOUT>(c, 1, cur);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // K-step t: read the packed words of chunk t+2 (landed one step ago), issue the DMA of chunk t+4 (its slot held
    // chunk t, read into registers two steps ago), dequantise chunk t+1 from registers into W16[(t+1) & 1]
    int t = 0;
    for (; t + 4 < c.NT; ++t) {   // steady state: everything unconditional
        if constexpr (!(ABL & ABL_NO_DEQ)) load_pk<LAYOUT>(c, t + 2, nxt);
        issue_bp<LAYOUT>(c, t + 4);
        if constexpr (!(ABL & ABL_NO_DEQ)) dequant_pk<LAYOUT>(c, t + 1, cur);
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");   // this step's 2 DMAs stay in flight
        cur = nxt;
        __builtin_amdgcn_s_barrier();
    }
    for (; t < c.NT; ++t) {
        if constexpr (!(ABL & ABL_NO_DEQ)) {
            if (t + 2 < c.NT) load_pk<LAYOUT>(c, t + 2, nxt);
            if (t + 1 < c.NT) dequant_pk<LAYOUT>(c, t + 1, cur);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        cur = nxt;
        __builtin_amdgcn_s_barrier();
    }
    next();   // the rings are idle from here on
}

// ------------------------------------------------------------------------------------------------
// hoisted-dequant mode: waves 8-11 are plain DMA waves for an fp16 weight tile
// ------------------------------------------------------------------------------------------------
// When a launch covers many token tiles (M >= 8192), dequantising the same 128 x 64 weight tile once per 256 tokens
// is the dominant avoidable cost: the dequant is then HOISTED out of the token loop -- the bit-exact dequant kernel
// (pack.hip) writes fp16 weights into a scratch buffer once, and this kernel's waves 8-11 stream its tiles into a
// 3-slot LDS ring (4 DMAs of 8 full 128-B rows per wave and K-step, source chunks XOR-swizzled like the x tile) two
// steps ahead, exactly as the MFMA waves stream x.  Same MFMA loop, same barriers, same epilogue; the products and
// their summation order are those of the fused mode, so the two modes agree bit for bit.
struct WDma {
    rsrc_t rsrc;         // weight rows n0 .. of this tile (range-checked: rows beyond N read as zeros)
    uint32_t voff[4];
    uint32_t k0;
    int d, NT;
};
__device__ __forceinline__ void wdma_setup(WDma& w, const uint16_t* __restrict__ w16, int N, int K, int n0, int kt0,
                                           int nsteps, int wave, int lane) {
    const int rows = N - n0 < BN ? N - n0 : BN;
    w.rsrc = make_rsrc(w16 + (int64_t)n0 * K, (uint32_t)rows * (uint32_t)K * 2u);
    w.k0 = (uint32_t)kt0 * (BK * 2);
    w.d = wave - N_MMA;
    w.NT = nsteps;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = w.d * 32 + i * 8 + (lane >> 3);
        w.voff[i] = (uint32_t)row * (uint32_t)K * 2u + ((((uint32_t)lane & 7u) ^ ((uint32_t)row & 7u)) << 4);
    }
}
__device__ __forceinline__ void issue_w(const WDma& w, char* smem, int t) {
    char* dst = smem + OFF_WD + (t % WD_SLOTS) * W_STAGE + w.d * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) bufdma16(w.rsrc, w.voff[i], w.k0 + (uint32_t)t * (BK * 2), dst + i * 1024);
}
__device__ __forceinline__ void wdma_prologue_issue(const WDma& w, char* smem) {
    issue_w(w, smem, 0);
    if (w.NT > 1) issue_w(w, smem, 1);
}
// barrier-for-barrier the twin of deq_segment (prologue barriers 1, 2, then one per K-step)
template <class Next>
__device__ __forceinline__ void wdma_segment(const WDma& w, char* smem, bool pre, Next&& next) {
    if (!pre) wdma_prologue_issue(w, smem);
    if (w.NT > 1 && !pre) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    int t = 0;
    for (; t + 2 < w.NT; ++t) {
        issue_w(w, smem, t + 2);                                   // slot (t+2) % 3 was last read in step t-1
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");           // this step's 4 DMAs stay in flight
        __builtin_amdgcn_s_barrier();
    }
    for (; t < w.NT; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    next();
}

#define MXQ_LANE_ID(ln) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln))

// grid = dp_grid persistent workgroups, which deal the first dp_tiles tiles round-robin (tile = block + k * dp_grid:
// blocks b and b + 8 share an XCD and dp_grid is a multiple of 8 or the tile count itself, so a workgroup's tiles
// keep its XCD's label) and overlap one tile's output with the next one's first DMAs, + 8 * units stream-K
// workgroups for the `tail` tiles beyond them.
template <int ABL, int LAYOUT>
__global__ __launch_bounds__(THREADS) void mxq_gemmx_f16_kernel(const uint16_t* __restrict__ x,
                                                               const uint32_t* __restrict__ qweight,
                                                               const float4* __restrict__ rowmeta,
                                                               uint16_t* __restrict__ y, int M, int N, int K,
                                                               int tiles_m, int tiles_n, int dp_tiles, int dp_grid,
                                                               int tail, int units, float* __restrict__ ws,
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
    auto nothing = [] {};
    if (wave < N_MMA) {
        if constexpr ((ABL & EXP_MMA_PRIO) != 0) __builtin_amdgcn_s_setprio(3);   // experiment: MFMA waves first
    } else {
        // the dequant chain is the longer one of a K-step: its VALU ops go first whenever they are ready (the MFMAs
        // lose a 4-cycle issue slot each time, the chain would lose up to 16)
        if constexpr (!(ABL & (EXP_NO_PRIO | EXP_MMA_PRIO))) __builtin_amdgcn_s_setprio(3);
    }

    if (bid < dp_grid) {
        // ---- persistent data-parallel workgroup: whole tiles bid, bid + dp_grid, ...
        int tm, tn;
        tile_of_block(bid, tiles_m, tiles_n, tm, tn);
        if (wave < N_MMA) {
            int ln;
            MXQ_LANE_ID(ln);
            XDma cur, nxt;
            xdma_setup(cur, x, M, K, tm * BM, 0, wave, ln);
            const bool chain = NT >= 2;    // a tile's last two steps copy the next tile's first two (needs two steps)
            if (chain) {
                issue_x<ABL, 0, 4>(cur, smem, wave, 0, 0);
                issue_x<ABL, 0, 4>(cur, smem, wave, 1, 1);
            }
            int g = 0;
            for (int tile = bid; tile < dp_tiles; tile += dp_grid) {
                MXQ_LANE_ID(ln);   // recomputed per tile and opaque: nothing lane-derived is hoisted (and spilled) across the loop
                const int m0 = tm * BM, n0 = tn * BN;
                const bool more = tile + dp_grid < dp_tiles;
                XDma none;
                xdma_none(none, x);
                if (more) {
                    tile_of_block(tile + dp_grid, tiles_m, tiles_n, tm, tn);
                    xdma_setup(nxt, x, M, K, tm * BM, 0, wave, ln);
                } else {
                    nxt = none;
                }
                mma_segment<ABL, LAYOUT == LAYOUT_DENSE16>(smem, wave, ln, NT, g, cur, chain ? nxt : none, chain, y, M, N, m0, n0, NT, sk);
                g = (g + NT) % A_SLOTS;
                cur = nxt;
            }
        } else if constexpr (LAYOUT == LAYOUT_DENSE16) {
            int ln;
            MXQ_LANE_ID(ln);
            WDma cur, nxt;
            wdma_setup(cur, (const uint16_t*)qweight, N, K, tn * BN, 0, NT, wave, ln);
            wdma_prologue_issue(cur, smem);
            for (int tile = bid; tile < dp_tiles; tile += dp_grid) {
                MXQ_LANE_ID(ln);
                const bool more = tile + dp_grid < dp_tiles;
                if (more) tile_of_block(tile + dp_grid, tiles_m, tiles_n, tm, tn);
                wdma_segment(cur, smem, true, [&] {
                    if (more) {
                        wdma_setup(nxt, (const uint16_t*)qweight, N, K, tn * BN, 0, NT, wave, ln);
                        wdma_prologue_issue(nxt, smem);
                    }
                });
                cur = nxt;
            }
        } else {
            int ln;
            MXQ_LANE_ID(ln);
            Deq cur, nxt;
            deq_setup<LAYOUT>(cur, smem, wave, ln, qweight, N, K, tn * BN, 0, NT);
            deq_prologue_issue<LAYOUT>(cur, rowmeta, N, tn * BN);
            for (int tile = bid; tile < dp_tiles; tile += dp_grid) {
                MXQ_LANE_ID(ln);
                const int n0 = tn * BN;
                const bool more = tile + dp_grid < dp_tiles;
                if (more) tile_of_block(tile + dp_grid, tiles_m, tiles_n, tm, tn);
                deq_segment<ABL, LAYOUT>(cur, wave, ln, rowmeta, N, n0, true, [&] {
                    if (more) {
                        deq_setup<LAYOUT>(nxt, smem, wave, ln, qweight, N, K, tn * BN, 0, NT);
                        deq_prologue_issue<LAYOUT>(nxt, rowmeta, N, tn * BN);
                    }
                });
                cur = nxt;
            }
        }
        return;
    }

    // ---- stream-K unit u of XCD e: K-steps [b0, b1) of that XCD's tail tiles laid end to end
    const int su = bid - dp_grid;
    sk.e = su & 7;
    sk.u = su >> 3;
    const int base = dp_tiles + sk.e;
    sk.S = ((tail + 7 - sk.e) >> 3) * NT;   // tail tile t belongs to XCD t & 7: the first tail % 8 XCDs hold one more
    const int b0 = sk_bound(sk.u, sk.S, units), b1 = sk_bound(sk.u + 1, sk.S, units);
    // every wave walks the same segment list, so the barrier counts of the two roles stay matched
    if (wave < N_MMA) {
        int pj0 = -1, pn0 = 0, pj1 = -1, pn1 = 0;
        for (int pos = b0; pos < b1;) {
            sk.j = pos / NT;
            const int end = b1 < (sk.j + 1) * NT ? b1 : (sk.j + 1) * NT;
            sk.first = pos == b0;
            int tm, tn;
            tile_of_block(base + sk.j * 8, tiles_m, tiles_n, tm, tn);
            int ln;
            MXQ_LANE_ID(ln);
            XDma xd, none;
            xdma_setup(xd, x, M, K, tm * BM, pos - sk.j * NT, wave, ln);
            xdma_none(none, x);
            mma_segment<ABL, LAYOUT == LAYOUT_DENSE16>(smem, wave, ln, end - pos, 0, xd, none, false, y, M, N, tm * BM, tn * BN, NT, sk);
            if (end - pos != NT) {
                if (sk.first) { pj0 = sk.j; pn0 = end - pos; }
                else { pj1 = sk.j; pn1 = end - pos; }
            }
            pos = end;
        }
        if (pj0 >= 0 || pj1 >= 0) {
            int ln;
            MXQ_LANE_ID(ln);
            // every slot store of this wave has reached the coherence point before any count moves
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int old0 = 0, old1 = 0;
            if (ln == 0) {   // both bumps in flight together
                if (pj0 >= 0) old0 = __hip_atomic_fetch_add(cnt + (pj0 * 8 + sk.e) * N_MMA + wave, pn0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pj1 >= 0) old1 = __hip_atomic_fetch_add(cnt + (pj1 * 8 + sk.e) * N_MMA + wave, pn1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            old0 = __builtin_amdgcn_readfirstlane(old0);
            old1 = __builtin_amdgcn_readfirstlane(old1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // compiler ordering only: the slot loads are agent-scope themselves
            if (pj0 >= 0 && old0 + pn0 == NT) {
                int tm, tn;
                tile_of_block(base + pj0 * 8, tiles_m, tiles_n, tm, tn);
                sk_finish(sk, pj0, NT, wave, ln, y, M, N, tm * BM, tn * BN);
            }
            if (pj1 >= 0 && old1 + pn1 == NT) {
                int tm, tn;
                tile_of_block(base + pj1 * 8, tiles_m, tiles_n, tm, tn);
                sk_finish(sk, pj1, NT, wave, ln, y, M, N, tm * BM, tn * BN);
            }
        }
    } else {
        for (int pos = b0; pos < b1;) {
            const int j = pos / NT;
            const int end = b1 < (j + 1) * NT ? b1 : (j + 1) * NT;
            int tm, tn;
            tile_of_block(base + j * 8, tiles_m, tiles_n, tm, tn);
            int ln;
            MXQ_LANE_ID(ln);
            if constexpr (LAYOUT == LAYOUT_DENSE16) {
                WDma w;
                wdma_setup(w, (const uint16_t*)qweight, N, K, tn * BN, pos - j * NT, end - pos, wave, ln);
                wdma_segment(w, smem, false, nothing);
            } else {
                Deq c;
                deq_setup<LAYOUT>(c, smem, wave, ln, qweight, N, K, tn * BN, pos - j * NT, end - pos);
                deq_segment<ABL, LAYOUT>(c, wave, ln, rowmeta, N, tn * BN, false, nothing);
            }
            pos = end;
        }
    }
}

static int cu_count() {
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

constexpr size_t CNT_BYTES = 64 * 1024;   // K-step counters at the head of the workspace (>= 8*units*N_MMA ints)

template <int ABL, int LAYOUT = MXQ_LAYOUT_MIXED>
static int launchx(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   void* workspace, size_t ws_bytes, bool force, hipStream_t stream) {
    // the DMA descriptors address one tile's rows with 32-bit offsets: 256 rows of x, 8 row-blocks of packed weights
    if ((int64_t)BM * K * 2 >= ((int64_t)1 << 32) || (int64_t)(BN / 16) * (K / BK) * MXQ_BLK_BYTES >= ((int64_t)1 << 32))
        return -1;   // MXQ_E_SHAPE
    hipError_t e = hipFuncSetAttribute((const void*)mxq_gemmx_f16_kernel<ABL, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    const int NT = K / BK;
    const int cus = cu_count() / 8 * 8, units = cus / 8;
    int dp_tiles = tiles, tail = 0;
    if (workspace && tiles % cus != 0 && units * 8 * N_MMA * sizeof(int) <= CNT_BYTES &&
        ws_bytes >= CNT_BYTES + (size_t)cus * 2 * BM * BN * sizeof(float)) {
        const int t8 = (tiles % cus) / 8;   // tail tiles per XCD (the first tail % 8 XCDs hold one more)
        // Splitting the tail costs ~20 us (every unit parks 128 KB of fp32 partials, the finishers read them back)
        // and saves the idle share of one tile time, (1 - tail/CUs) * NT K-steps of ~1 us: worth it from ~24 idle
        // K-steps per CU (M = 512: 60 -> 37 us at 4096^2; NOT Llama's gate/up at M = 2048, tail 176 / 256, NT = 64)
        const bool pays = (int64_t)(cus - tiles % cus) * NT >= (int64_t)24 * cus;
        if ((force || pays) && (int64_t)t8 * NT >= (int64_t)units * 4) {
            tail = tiles % cus;
            dp_tiles = tiles - tail;
        }
    }
    const int dp_grid = dp_tiles < cus ? dp_tiles : cus;   // persistent: at most one data-parallel workgroup per CU
    const int grid = dp_grid + (tail ? cus : 0);
    mxq_gemmx_f16_kernel<ABL, LAYOUT><<<grid, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n,
        dp_tiles, dp_grid, tail, units, (float*)((char*)workspace + CNT_BYTES), (int*)workspace);
    return (int)hipGetLastError();
}

}   // namespace

size_t mxq_gemmx_workspace_bytes() { return CNT_BYTES + (size_t)(cu_count() / 8 * 8) * 2 * BM * BN * sizeof(float); }

int mxq_launch_gemmx_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                         void* workspace, size_t ws_bytes, int force, hipStream_t stream) {
    return launchx<0>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, force != 0, stream);
}

int mxq_launch_gemmx_layout_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                int layout, void* workspace, size_t ws_bytes, hipStream_t stream) {
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launchx<0, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, false, stream);
        case MXQ_LAYOUT_W2G16: return launchx<0, MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, false, stream);
        case MXQ_LAYOUT_W4ROW: return launchx<0, MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, false, stream);
        case MXQ_LAYOUT_MIXEDC: return launchx<0, MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, false, stream);
    }
    return -1;
}

// hoisted-dequant mode: w16 = dense fp16 [N, K] weight (the dequant kernel's output); same tiles, no stream-K tail
int mxq_launch_gemmx_dense_f16(const void* x, const void* w16, void* y, int M, int N, int K, hipStream_t stream) {
    return launchx<0, LAYOUT_DENSE16>(x, w16, nullptr, y, M, N, K, nullptr, 0, false, stream);
}

#ifdef MXQ_PROFILING
// Built only into libmxq_hip_prof.so (make prof; tools/): parts of the kernel removed to time the rest.
// WRONG RESULTS by construction -- never part of libmxq_hip.so or of include/mxq_hip.h.
// 1 = no x loads, 2 = no MFMA, 4 = no dequant, 256 = no output stores (sums); 1024 / 2048: issue priorities (correct)
extern "C" int mxq_prof_gemmx_ablate_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N,
                                         int K, int abl, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    switch (abl) {
        case 0: return launchx<0>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 1: return launchx<1>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 2: return launchx<2>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 4: return launchx<4>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 5: return launchx<5>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 6: return launchx<6>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 256: return launchx<256>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 1024: return launchx<1024>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);   // correct results
        case 2048: return launchx<2048>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);   // correct results
    }
    return -1;   // MXQ_E_SHAPE: not an ablation this build carries
}
#endif   // MXQ_PROFILING
