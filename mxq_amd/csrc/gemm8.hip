// W2/4 x A16 dequant-GEMM for prefill (any token count > 8): MFMA waves own the activation stream, dedicated waves
// own ALL of the dequant; persistent over tiles, hybrid stream-K tail.
//
// Counterpart of the reference's (never built) AWQ tensor-core GEMM
// mxq_quant/cuda_kernel/csrc/quantization/gemm_cuda_gen.cu:28-218 and of the implicit nn.Linear on the
// fake-quant weight (mxq_quant/main.py:85); arithmetic contract x16 . fp16(scale * (q - zero))^T of
// lib/quantizer.py:19-20 + mxqgpt.py:448, fp32 accumulation.
//
// Tile 256 tokens x 128 channels x K-step 64 (= one MXQ chunk); 12 waves with fixed roles:
//   * waves 0-7  "MFMA waves" (one 64 x 64 sub-tile = 4 x 4 v_mfma_f32_16x16x32_f16 each): fragment reads + 32 MFMAs
//     per K-step, and the x tile's LDS-DMA (4 x 1 KiB buffer_load ... lds each per K-step, addressed by one
//     per-lane offset VGPR each + a scalar K offset; rows beyond M read as zeros through the buffer descriptor's
//     range check) -- no VALU work in the loop at all.  (Round 1's kernel put the 2-bit dequant, ~48 VALU ops per
//     K-step, INSIDE the MFMA waves' instruction streams: an in-order wave that carries VALU chains between its
//     MFMAs stalls its MFMAs on them, and both MFMA waves of a SIMD do so in lockstep.)
//   * waves 8-11 "dequant waves" (one per SIMD): the whole dequant (2-bit LUT / v_perm groups and the 4-bit arm)
//     into the fp16 W16 tile.  The packed words come straight from global memory into registers (6-7 dword loads
//     per thread and chunk through the tile's buffer descriptor), a GROUP of 3 chunks at a time; a burst converts the
//     whole group into result registers and issues the next group's loads, the 3 K-steps after it only write one
//     chunk each into the W16 double buffer (deq_segment_h: why bursts).  Raised issue priority, scalar fp32 ops
//     (no SLP packing: Makefile).
//   * one raw s_barrier per K-step; x ring 3 slots (DMA two steps ahead, counted vmcnt), W16 double buffer
//     (hoisted-dequant mode: a 3-slot ring of fp16 weight tiles instead): 144 KiB of LDS.  D^T = W . x^T: a lane
//     owns 4 consecutive channels of a token.
//   * the output leaves without an LDS round trip (store_tile_xpose), so a workgroup that runs several tiles issues
//     the next tile's first DMAs behind the last barrier of this one and they fly under its epilogue.
//
// Grid = min(tiles, CUs) persistent workgroups dealing whole tiles round-robin; with a workspace, the tiles beyond the
// last full round ("tail") are split along K over the SAME workgroups (hybrid stream-K; the reference launcher's
// split_k_iters, gemm_cuda_gen.cu:429-478): workgroup 8u + e is also unit u of XCD e (tail tile t belongs to XCD t & 7; an
// XCD's units share its tail tiles so the operands stay in that L2) and runs K-steps [bound(u), bound(u+1)) of that XCD's
// tail tiles laid end to end -- up to one piece that ENDS a tile, whole tiles, and one piece that STARTS a tile.  Each
// piece ("segment") runs the same pipeline on a shifted K window.  The unit that holds a tile's LAST K-step is the
// tile's OWNER.  A unit runs its pieces in DESCENDING K order: first the piece that starts a tile (it cannot be the
// owner's): its fp32 accumulators go to a workspace slot and -- once those stores have retired, which the next
// segment's first DMA wait covers for free -- a per-(tile, wave) K-step counter is bumped; last the piece that ends a
// tile: the owner keeps ITS accumulators in registers, waits until the counter shows every other contributor's steps,
// adds their slots in unit order (a fixed order: results do not depend on timing), writes fp16 y and re-zeroes the
// counter.  (Round 2-3: every contributor parked, the last arriver read them all back -- 2-3 slots of 128 KB per
// workgroup, all at the end of the launch: ~20 us; now the parks happen mid-launch and an owner reads 1-2 slots.)
// An owner waits only for units with a LOWER index, which never wait before their parked piece: no cycle; every
// workgroup of the grid (<= one per CU, 144 KB of LDS each) is resident, and the spin is bounded.  Slots and counters
// cross XCDs (one L2 each): slot traffic is agent-scope sc1 accesses, ordered against the counter by vmcnt.
#include <hip/hip_runtime.h>

#include <type_traits>

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

// The token height of the tile is a build parameter: this file is compiled twice.  gemm8.hip itself: 256 tokens, 8 MFMA
// waves (2 per SIMD) -- the prefill kernel.  gemm8h.hip (#defines MXQ_G8_BM 128 and includes this file): 128 tokens, 4 MFMA
// waves (one per SIMD, next to the SIMD's dequant wave) -- twice the conversion work per flop, half the tile: for launches
// whose 256-token tiles would leave most of the chip to the stream-K fix-up (<= ~1k tokens; profiles/r04_gemm8h.txt).
// Everything else -- the dequant waves, the 128-channel W tile, the K pipeline, the stream-K protocol -- is the same code.
#ifndef MXQ_G8_BM
#define MXQ_G8_BM 256
#endif
#ifndef MXQ_G8_BN
#define MXQ_G8_BN 128
#endif
// MXQ_G8_AWQ (gemm8a.hip, gemm8aq.hip): the same kernel with dequant waves for the OPERANDS of the reference's
// gemm_forward_cuda (K-major 4-bit words, group-wise fp16 scales, packed integer zero-points: "AWQ layout" below) instead of
// the MXQ block formats -- one more build of this file, whose kernels take three more arguments.
#if defined(MXQ_G8_AWQ) && MXQ_G8_BM == 256
#define G8_NAME(stem) mxq_##stem##gemm8a
#elif defined(MXQ_G8_AWQ) && MXQ_G8_BM == 128
#define G8_NAME(stem) mxq_##stem##gemm8ah
#elif defined(MXQ_G8_AWQ)
#define G8_NAME(stem) mxq_##stem##gemm8aq
#elif MXQ_G8_BM == 256
#define G8_NAME(stem) mxq_##stem##gemm8
#elif MXQ_G8_BM == 128 && MXQ_G8_BN == 128
#define G8_NAME(stem) mxq_##stem##gemm8h
#elif MXQ_G8_BM == 128
#define G8_NAME(stem) mxq_##stem##gemm8n
#elif MXQ_G8_BN == 128
#define G8_NAME(stem) mxq_##stem##gemm8q
#else
#define G8_NAME(stem) mxq_##stem##gemm8m
#endif
#define G8_CAT_(a, b) a##b
#define G8_CAT(a, b) G8_CAT_(a, b)
#define G8_SYM(stem, suffix) G8_CAT(G8_NAME(stem), suffix)
#define G8_KERNEL G8_SYM(, _f16_kernel)
constexpr int BM = MXQ_G8_BM, BN = MXQ_G8_BN, BK = 64;
static_assert(BN == 128 || (BN == 64 && BM <= 128), "tile width");
static_assert(BM == 256 || BM == 128 || BM == 64, "tile height");
// Dequant waves: a thread converts one PART of one row of the 128 x 64 weight tile per K-step.  256-token tile: 4 waves,
// parts = column halves (32 weights).  128-token tile: the MFMA side of a K-step is half as long, and the MFMA waves were
// found waiting ~30 % of it for the conversion (tools/gemm_stamps.py --half: 1261 cycles per step, 922 with the conversion
// switched off) -- so there the tile is converted by 8 waves, parts = 16-column quarters (3 waves per SIMD, as gemm8).
#ifndef MXQ_G8_NDEQ
#define MXQ_G8_NDEQ (MXQ_G8_BM == 256 ? 4 : MXQ_G8_BN == 128 ? 8 : 4)
#endif
// MFMA waves: a WGM (tokens) x WGN (channels) grid of sub-tiles of 16 NJ tokens x 64 channels.  128-channel tiles: 64 x 64
// sub-tiles, 2 waves across the channels; the 64-channel tile (gemm8n.hip: 128 x 64, for launches whose fp32 partial tiles are
// the cost) has one wave across and four of 32 x 64 down the tokens.
constexpr int WGN = BN / 64, N_MMA = BM / 32, WGM = N_MMA / WGN;
constexpr int NJ = BM / (16 * WGM);           // 16-token blocks per MFMA wave (4 | 2)
constexpr int FRAGS = 4 * NJ;                 // accumulator fragments per MFMA wave: f = i * NJ + j (W block i, token block j)
constexpr int WSLOT = FRAGS * 1024;           // bytes of a wave's share of a partial-tile slot (16 B per lane and fragment)
constexpr int N_DEQ = MXQ_G8_NDEQ, THREADS = (N_MMA + N_DEQ) * 64;
static_assert(BM / (8 * N_MMA) == 4, "every MFMA wave stages 32 rows of the x tile per K-step");
constexpr int DEQ_PARTS = N_DEQ * 64 / BN;    // parts of a row per thread: 2 (halves) or 4 (quarters)
constexpr int DEQ_NRES = 8 / DEQ_PARTS;       // 16-byte W16 slots a thread writes per chunk
constexpr int A_STAGE = BM * BK * 2, A_SLOTS = 3;
constexpr int W_STAGE = BN * BK * 2;
constexpr int OFF_A = 0;
constexpr int OFF_W = OFF_A + A_SLOTS * A_STAGE;
// hoisted-dequant mode: a 3-slot ring of fp16 weight tiles in the place of the W16 double buffer
constexpr int OFF_WD = OFF_W, WD_SLOTS = 3;
#if defined(MXQ_G8_AWQ)
// AWQ layout: the K-major code words reach the dequant waves through LDS -- one 1-KiB LDS-DMA per dequant wave
// and K-step brings the step's 64 rows x 64 bytes in whole row segments, the threads then read their 8 words (k = 8 ko ..
// 8 ko + 7 of one channel octet) with ds_read_b32.  (The same words fetched straight into registers, 8 row-strided dword
// loads per thread and K-step: 91 us at 2048 x 4096^2 against 65 with the loads switched off -- each wave instruction
// touched 8 rows x 16 bytes.)  A ring of three 4-KiB slots behind the weight-tile buffers: one group of DEQ_R = 3 chunks in
// the 256-token build; chunk q in slot q % 3, fetched two steps ahead, in the small-tile builds (DEQ_R = 1).
#define MXQ_AWQ_RAW 1
constexpr int OFF_RAW = OFF_WD + WD_SLOTS * W_STAGE, RAW_SLOT = 4096;
constexpr int SMEM_BYTES = OFF_RAW + 3 * RAW_SLOT;
#else
constexpr int SMEM_BYTES = OFF_WD + WD_SLOTS * W_STAGE;
#endif
static_assert(OFF_W + 2 * W_STAGE <= SMEM_BYTES && SMEM_BYTES <= 160 * 1024, "LDS budget");
static_assert(N_MMA * WSLOT <= SMEM_BYTES, "the stream-K owner stages a wave's share of a slot per MFMA wave in the idle rings");
constexpr int LAYOUT_DENSE16 = 100;   // internal: qweight is a dense fp16 [N, K] matrix (never part of the C ABI)
constexpr int LAYOUT_AWQ = 101;       // internal (MXQ_G8_AWQ builds): qweight = int32 [K, N / 8] words of gemm_forward_cuda
#ifdef MXQ_G8_AWQ
static_assert(MXQ_G8_BN == 128, "the AWQ dequant waves cover 16 channel octets");
// the rest of gemm_forward_cuda's operands (mxq_quant/cuda_kernel/csrc/quantization/gemm_cuda.h:3-4): passed by value
struct AwqOps {
    const uint16_t* scales;   // fp16 [K / G, N]
    const uint32_t* zeros;    // int32 [K / G, N / 8], nibble order of the code words
    uint32_t gmul;            // floor(2^32 / G) + 1: k / G = mulhi(k, gmul) for every k < 2^20 (G < 2^12)
    int groups;               // K / G
};
#define G8_AWQ_PARAM , AwqOps awq
#define G8_AWQ_ARG , awq
// the launch's uniform facts, built once from kernel arguments and handed down BY REFERENCE (never copied into the per-tile
// Deq, never selected between tiles: descriptors that pass through a select live in VGPRs and every load through them
// gets a waterfall loop -- and here the copies even went to scratch)
struct AwqU {
    __amdgpu_buffer_rsrc_t rs_q, rs_s, rs_z;   // code words [K, N / 8], scales [K / G, N], zeros [K / G, N / 8]: whole tensors
    uint32_t gmul, row_bytes;                  // row_bytes = N / 2: one k row of code words = one group row of zeros
    int last_kt;                               // K / 64 - 1
};
#define G8_AWQU_PARAM , const AwqU& au, int& raw_slot
#define G8_AWQU_ARG , au, raw_slot
#else
#define G8_AWQ_PARAM
#define G8_AWQ_ARG
#define G8_AWQU_PARAM
#define G8_AWQU_ARG
#endif

// profiling-only switches (template parameter ABL; the product library instantiates ABL = 0 only, and the stamp / ballast
// code exists only under MXQ_PROFILING: libmxq_hip_prof.so)
[[maybe_unused]] constexpr int ABL_NO_MFMA = 2, ABL_NO_DEQ = 4, ABL_NO_XDMA = 1, ABL_NO_STORE = 256;
// 16: the dequant waves skip the 4-bit arm (its loads and its conversion); 32: the OLDER MFMA wave of each SIMD (waves 0-3)
// carries 30 independent VALU ops per K-step -- what taking that arm over would cost it.  Timing only.
[[maybe_unused]] constexpr int ABL_NO_Q4 = 16, ABL_MMA_VALU = 32;
// scheduling experiments (correct results): issue priorities of the two roles
[[maybe_unused]] constexpr int EXP_NO_PRIO = 1024, EXP_MMA_PRIO = 2048, EXP_STAMPS = 4096;
// r04 "halving" ablations (timing only; the operands stay RANDOM, unlike the remove-it-all ablations whose constant LDS
// contents let the chip clock higher): 512 = the MFMA waves issue 2 of their 4 x pieces per step (the other rows of the
// slot keep an earlier step's x); 8192 = a burst converts only the first of its 3 chunks (the other two W16 writes carry
// the previous burst's weights)
[[maybe_unused]] constexpr int ABL_HALF_XDMA = 512, ABL_THIRD_DEQ = 8192;
// stream-K diagnostics (profiling builds, correct results): 16384 = wall-clock stamps of the fix-up phases (100 MHz
// s_memrealtime, MFMA wave 0 of every workgroup -> 8 u64 per workgroup at byte 32768 of the workspace head);
// 32768 = the all-contributors reduction from 3 contributors per tile on (product: SK_DIST_MIN)
[[maybe_unused]] constexpr int EXP_SKSTAMPS = 16384, EXP_SK_DIST3 = 32768;
// 65536 = fault injection for the expiry test (tests/test_gpu_parity.py, profiling library only): unit 0 never counts its
// parked pieces and every wait gives up after 4096 polls -- the launch must end with the workspace's status words set and
// the starved tiles poisoned, not with silently wrong sums
[[maybe_unused]] constexpr int EXP_SK_WITHHOLD = 65536;

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
    int owner;        // 1: a partial segment that ENDS its tile: finish the tile here (sk_finish_owner) ...
    int dist;         // ... unless the tile has >= SK_DIST_MIN contributors: then every piece is parked (sk_reduce_distributed)
    int pend_j, pend_n;   // a parked piece whose counter bump is due as soon as its stores have retired (-1: none)
    float* park;      // slices mode (below): this workgroup's slab -- every piece is parked there, nothing is counted
};
// A wave's 16 KB slot, 16 bytes per lane and fragment: ONE sc1 (write-through / L1-bypassing) 16-byte access per fragment,
// 1 KB contiguous per instruction.  (Round 2 began with two 8-byte agent-scope atomics per fragment: 8-byte accesses run at
// 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md; nothing here needs atomicity -- a slot is read only after the
// K-step count that its writer bumped behind s_waitcnt vmcnt(0) is complete.)
__device__ __forceinline__ void st_agent(float* slot, int f, int lane, f32x4 v) {
    const rsrc_t r = make_rsrc(slot, (uint32_t)WSLOT);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (uint32_t)(f * 64 + lane) * 16u, 0u, 16);
}
__device__ __forceinline__ f32x4 ld_agent(const float* slot, int f, int lane) {
    const rsrc_t r = make_rsrc(slot, (uint32_t)WSLOT);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)(f * 64 + lane) * 16u, 0u, 16));
}
__device__ __forceinline__ int sk_bound(int u, int S, int units) { return (int)((uint32_t)u * (uint32_t)S / (uint32_t)units); }

// ------------------------------------------------------------------------------------------------
// MFMA waves
// ------------------------------------------------------------------------------------------------
typedef half8 Frag4[4];
typedef half8 FragX[NJ];
typedef f32x4 AccT[4][NJ];

// DENSE: the weight tile is a 3-slot ring of fp16 tiles filled by LDS-DMA (hoisted-dequant mode, below) instead of the
// double buffer the dequant waves write
template <bool DENSE>
__device__ __forceinline__ void load_frags(const char* smem, int t, int kk, int wm, int wn, int fr, int fq, Frag4& wf,
                                           FragX& xf) {
    const char* a_base = smem + OFF_A + (t % A_SLOTS) * A_STAGE;
    const char* w_base = DENSE ? smem + OFF_WD + (t % WD_SLOTS) * W_STAGE : smem + OFF_W + (t & 1) * W_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *(const half8*)(w_base + swz(wn * 64 + i * 16 + fr, kk * 4 + fq));
#pragma unroll
    for (int j = 0; j < NJ; ++j) xf[j] = *(const half8*)(a_base + swz(wm * (16 * NJ) + j * 16 + fr, kk * 4 + fq));
}

// ABL_FILL (profiling builds): FILLN independent v_add_f32 behind every MFMA, pinned there by sched_group_barrier -- what
// vector-ALU work costs when it sits IN the MFMA waves' own streams, one or two ops per MFMA gap (r04 probe)
struct Fill { float f[4]; };
template <int I0, int I1, int ABL>
__device__ __forceinline__ void mfma_rows(AccT& acc, const Frag4& wf, const FragX& xf, Fill* fl = nullptr) {
    constexpr int FILLN = (ABL >> 6) & 3;
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (ABL & ABL_NO_MFMA) asm volatile("" ::"v"(wf[i]), "v"(xf[j]));
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            if constexpr (FILLN > 0) {
#pragma unroll
                for (int q = 0; q < FILLN; ++q) fl->f[(j * FILLN + q) & 3] += 1.5f;
            }
        }
    if constexpr (FILLN > 0) {
#pragma unroll
        for (int n = 0; n < (I1 - I0) * NJ; ++n) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, FILLN, 0);   // FILLN VALU ops
        }
    }
}

// the x tile of K-step t: wave w fills rows 32w .. 32w+31 of slot t % 3 with 4 DMAs of 8 full 128-B rows;
// lane -> (row 8i + lane / 8, 16-B chunk lane % 8), source chunk XOR-swizzled so that the fragment reads are
// conflict-poor (swz above).  voff[i] = that row's byte offset from the tile's first row + the chunk.
struct XDma {
    rsrc_t rsrc;         // x rows m0 .. of this tile (range-checked: rows beyond M read as zeros)
    uint32_t voff[4];
    uint32_t k0;         // byte offset of the segment's first K-step inside a row
};
template <int I0, int I1>
__device__ __forceinline__ void issue_x(const XDma& d, char* smem, int wave, int t) {
    char* dst = smem + OFF_A + (t % A_SLOTS) * A_STAGE + wave * 4096;
#pragma unroll
    for (int i = I0; i < I1; ++i) bufdma16(d.rsrc, d.voff[i], d.k0 + (uint32_t)t * (BK * 2), dst + i * 1024);
}

#define MXQ_FENCE() __builtin_amdgcn_sched_barrier(0)

// diagnostic builds only (libmxq_hip_prof.so, EXP_STAMPS): cycle stamps around the phases of a K-step; sums leave through the workspace
typedef unsigned long long u64t;
struct Stamps { u64t work, wait, bar, n; };
#ifdef MXQ_PROFILING
__device__ __forceinline__ u64t stamp() {
    u64t t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MXQ_STAMPS(ABL) (((ABL) & EXP_STAMPS) != 0)
#else
__device__ __forceinline__ u64t stamp() { return 0; }
#define MXQ_STAMPS(ABL) false      /* the product build carries no stamp code */
#endif
#ifdef MXQ_PROFILING
#define MXQ_SKSTAMPS(ABL) (((ABL) & EXP_SKSTAMPS) != 0)
#else
#define MXQ_SKSTAMPS(ABL) false
#endif
// phase stamp i of this workgroup (wave 0, lane 0 only; overwritten by later pieces: the LAST piece's phases remain)
template <int ABL>
__device__ __forceinline__ void sk_stamp(int* cnt, int wave, int lane, int i) {
    if constexpr (MXQ_SKSTAMPS(ABL)) {
        u64t t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (wave == 0 && lane == 0 && cnt) ((u64t*)((char*)cnt + 32768))[(int64_t)blockIdx.x * 8 + i] = t;
    }
}

// One K-step t >= 1: MFMAs of (t-1, kk=1) and (t, kk=0), fragment reads of step t, and -- ISSUE -- the x DMAs of
// step t+2 (slot (t+2) % 3 was last read in step t-1), spread behind groups of MFMAs.
template <int ABL, bool ISSUE, bool DENSE>
__device__ __forceinline__ void mma_step(char* smem, int t, int wave, int wm, int wn, int fr, int fq, const XDma& xd,
                                         AccT& acc, Frag4& wf0, FragX& xf0, Frag4& wf1, FragX& xf1,
                                         Stamps& st) {
    u64t t0 = 0, t1 = 0, t2 = 0;
    [[maybe_unused]] Fill fl = {{(float)t, (float)t + 1.f, (float)t + 2.f, (float)t + 3.f}};
    if constexpr (MXQ_STAMPS(ABL)) t0 = stamp();
    mfma_rows<0, 1, ABL>(acc, wf1, xf1, &fl);
    MXQ_FENCE();
    load_frags<DENSE>(smem, t, 0, wm, wn, fr, fq, wf0, xf0);
    MXQ_FENCE();
    mfma_rows<1, 2, ABL>(acc, wf1, xf1, &fl);
    MXQ_FENCE();
    if constexpr (ISSUE && !(ABL & ABL_NO_XDMA)) issue_x<0, 2>(xd, smem, wave, t + 2);
    MXQ_FENCE();
    mfma_rows<2, 4, ABL>(acc, wf1, xf1, &fl);
    MXQ_FENCE();
    load_frags<DENSE>(smem, t, 1, wm, wn, fr, fq, wf1, xf1);
    MXQ_FENCE();
    mfma_rows<0, 2, ABL>(acc, wf0, xf0, &fl);
    MXQ_FENCE();
#ifdef MXQ_PROFILING
    if constexpr ((ABL & ABL_MMA_VALU) != 0) {
        if (wave < 4) {
            float b0 = (float)t, b1 = b0 + 1.f, b2 = b0 + 2.f, b3 = b0 + 3.f, b4 = b0 + 4.f;
#pragma unroll
            for (int q = 0; q < 6; ++q)
                asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_add_f32 %4, %4, %4"
                             : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4));
            asm volatile("" ::"v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4));
        }
    }
#endif
    MXQ_FENCE();
    if constexpr (ISSUE && !(ABL & (ABL_NO_XDMA | ABL_HALF_XDMA))) issue_x<2, 4>(xd, smem, wave, t + 2);
    MXQ_FENCE();
    mfma_rows<2, 4, ABL>(acc, wf0, xf0, &fl);
    if constexpr (((ABL >> 6) & 3) != 0) asm volatile("" ::"v"(fl.f[0]), "v"(fl.f[1]), "v"(fl.f[2]), "v"(fl.f[3]));
    if constexpr (MXQ_STAMPS(ABL)) t1 = stamp();
    // this step's 4 DMAs stay in flight across the barrier; the previous step's (x of step t+1) have landed
    if constexpr (ISSUE && (ABL & ABL_HALF_XDMA) != 0) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (ISSUE && !(ABL & ABL_NO_XDMA)) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if constexpr (MXQ_STAMPS(ABL)) t2 = stamp();
    __builtin_amdgcn_s_barrier();
    if constexpr (MXQ_STAMPS(ABL)) {
        const u64t t3 = stamp();
        st.work += t1 - t0; st.wait += t2 - t1; st.bar += t3 - t2; st.n += 1;
    }
}

// The tile WITHOUT an LDS round trip: a lane's accumulators are 4 channels (8 B as fp16) of W-fragment block i for
// each of 4 token blocks j; the 4 lanes {fr, fr+16, fr+32, fr+48} hold one token's 64 channels as a 4 x 4 grid of
// 8-byte cells (block i, quarter fq).  Two butterfly stages of lane swaps (v_permlane32_swap: lanes +-32 <-> blocks
// +-2; v_permlane16_swap: lanes +-16 <-> blocks +-1) transpose the grid, after which lane fq owns block fq whole:
// 32 contiguous bytes = two 16-byte stores, and the 4 lanes together write the token's full 128-B line.  Needs no
// LDS, so the x ring can be refilled for the NEXT tile while this one is still being written (persistent loop).
// ... one token block j of a wave's sub-tile: c4[i] = the lane's 4 channels of W-fragment block i for token j * 16 + fr
__device__ __forceinline__ void store_block_xpose(const f32x4 (&c4)[4], uint16_t* __restrict__ y, int M, int N, int m0, int n0,
                                                  int wm, int wn, int j, int fr, int fq) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    const int n = n0 + wn * 64 + fq * 16;
    uint32_t c[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[i][0] = mxq_pack_f16(c4[i][0], c4[i][1]);
        c[i][1] = mxq_pack_f16(c4[i][2], c4[i][3]);
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        u32x2v r;
        r = __builtin_amdgcn_permlane32_swap(c[0][d], c[2][d], false, false); c[0][d] = r[0]; c[2][d] = r[1];
        r = __builtin_amdgcn_permlane32_swap(c[1][d], c[3][d], false, false); c[1][d] = r[0]; c[3][d] = r[1];
        r = __builtin_amdgcn_permlane16_swap(c[0][d], c[1][d], false, false); c[0][d] = r[0]; c[1][d] = r[1];
        r = __builtin_amdgcn_permlane16_swap(c[2][d], c[3][d], false, false); c[2][d] = r[0]; c[3][d] = r[1];
    }
    const int m = m0 + wm * (16 * NJ) + j * 16 + fr;
    if (m < M && n < N) {
        uint16_t* dst = y + (int64_t)m * N + n;
        // streaming stores: y is written once and not read by this kernel, so it should not displace x / W
        // lines in L2 (+0.7-1.5 % at 32768 tokens, neutral at 2048: profiles/r03_nt_stores.txt)
        __builtin_nontemporal_store((u32x4){c[0][0], c[0][1], c[1][0], c[1][1]}, (u32x4*)dst);
        __builtin_nontemporal_store((u32x4){c[2][0], c[2][1], c[3][0], c[3][1]}, (u32x4*)(dst + 8));
    }
}
__device__ __forceinline__ void store_tile_xpose(const AccT& acc, uint16_t* __restrict__ y, int M, int N,
                                                 int m0, int n0, int wm, int wn, int fr, int fq) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 c4[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
        store_block_xpose(c4, y, M, N, m0, n0, wm, wn, j, fr, fq);
    }
}

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
// a segment's prologue DMAs: x of its steps 0 and 1 (needs the whole x ring idle: every read of the previous
// segment's slots lies before that segment's last barrier)
template <int ABL>
__device__ __forceinline__ void mma_prologue_issue(const XDma& xd, char* smem, int wave, int NT) {
    if constexpr (!(ABL & ABL_NO_XDMA)) {
        issue_x<0, 4>(xd, smem, wave, 0);
        if (NT > 1) issue_x<0, 4>(xd, smem, wave, 1);
    }
}

// contributors of tail tile j of this XCD (units whose K range meets it, in unit order): their number, and this unit's
// place among them (-1: none of its steps)
// timing-only ablation (WRONG results): every stream-K piece reads its operands from K-step 0 on, i.e. all units of an XCD
// walk the same K window at the same time -- the upper bound of what K-aligned pieces could share through L2
#ifdef MXQ_SK_ABL_KOFF0
#define SK_KOFF(k) 0
#else
#define SK_KOFF(k) (k)
#endif
constexpr int SK_SPIN_BOUND = 1 << 22;      // polls (each one agent-scope load + s_sleep: several seconds in all)
template <int ABL> constexpr int sk_spin_bound() { return (ABL & EXP_SK_WITHHOLD) ? 4096 : SK_SPIN_BOUND; }
// A wait that gives up must not pass for a result.  Status words at the END of the workspace's 64-KiB head (ints
// SK_STATUS_OFF .. +3; the K-step counts, the "done" counts and the profiling stamps all lie below byte 49152):
//   [0] code (0 = fine; 1 = an owner's wait expired, 2 = a wait of the all-contributors reduction expired)
//   [1] tail-tile index, [2] workgroup, [3] the count it saw.
// The first failure's details stay (later ones only keep [0] non-zero); the tile (or token block) whose operands never
// arrived is written as NaN.  The counters are then inconsistent: the host re-zeroes the head before the workspace is used
// again (mxq_workspace_status reports, packing.workspace_status raises and resets).
// HOST MAILBOX (round 6): ints SK_MAILBOX_OFF, +1 of the head hold a 64-bit address of 4 ints of PINNED HOST memory (0: none;
// written by the caller that owns the workspace, include/mxq_hip.h).  The first failure also stores its four status ints
// there with system-scope stores -- the code last -- so the host sees an expired wait by reading its own memory, without
// synchronising with the device: the product's next call on that workspace raises (mxq_amd/packing.py).  Costs nothing
// unless a wait expires.
constexpr int SK_STATUS_OFF = 16380, SK_MAILBOX_OFF = 16376;
__device__ __forceinline__ void sk_fail(int* cnt, int code, int j, int seen, int lane) {
    if (lane == 0) {
        int* st = cnt + SK_STATUS_OFF;
        if (__hip_atomic_exchange(st, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __hip_atomic_store(st + 1, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st + 2, (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st + 3, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long lo = (unsigned)__hip_atomic_load(cnt + SK_MAILBOX_OFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = (unsigned)__hip_atomic_load(cnt + SK_MAILBOX_OFF + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int* mb = (int*)(lo | (hi << 32));
            if (mb != nullptr) {
                __hip_atomic_store(mb + 1, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(mb + 2, (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(mb + 3, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(mb, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
#ifndef MXQ_SK_DIST_MIN
#define MXQ_SK_DIST_MIN 6
#endif
constexpr int SK_DIST_MIN = MXQ_SK_DIST_MIN;       // from this many contributors on, ALL of them reduce the tile (sk_reduce_distributed)
constexpr int SK_DONE_OFF = 4096;    // ints: per-tile "done" counts behind the per-(tile, wave) K-step counts
// (unit u holds step k iff bound(u) <= k < bound(u + 1), i.e. u = ceil((k + 1) units / S) - 1; every unit of an XCD with a tail
//  holds at least one step -- launcher: S >= 4 units -- so the contributors are the contiguous range uf .. uf + C - 1.  Two
//  integer divisions; a scan over the units costs 64 of them, ~5 us of vector-ALU time, and ran once per piece and per task
//  in the first build)
__device__ __forceinline__ int sk_contributors(const SkSeg& sk, int j, int NT_tile, int& uf) {
    const uint32_t lo = (uint32_t)j * (uint32_t)NT_tile, hi = lo + (uint32_t)NT_tile;
    const uint32_t U = (uint32_t)sk.units, S = (uint32_t)sk.S;
    uf = (int)(((lo + 1u) * U + S - 1u) / S) - 1;
    const int ul = (int)((hi * U + S - 1u) / S) - 1;
    return ul - uf + 1;
}

// (same-process A/B, profiles/r04_streamk.txt: all-contributors reduction wins where a tile has 8 contributors that finish
//  together -- gate/up at 640 / 768 / 1536 tokens: 82.1 -> 77.0, 83.5 -> 78.1, 145.2 -> 140.9 us -- and loses 1-8 us at 3-4
//  contributors, where the owner's one or two extra round trips cost less than everybody's park + sync: threshold 6)
// A tile with >= SK_DIST_MIN contributors: EVERY contributor has parked its piece (the one that ends the tile too) and every
// one of them sums a share -- the 32 (producing wave, token block) tasks of the tile are dealt over the contributors' 8 C
// waves; a task waits until the producing wave's K-step count is complete, brings the C slots' four fragments through LDS
// (up to four slots per batch: 16 sc1 LDS-DMA pieces in flight = one round trip), adds them in unit order and writes the
// token block.  The contributors of such a tile finish together (equal shares of K-steps), so an owner alone would read
// their slots one round trip after the other -- 7 of them at 640 tokens x 11008 (258 tiles: 2 tail tiles over 8 units
// each): 84.9 us against 65 of K-steps.  Any contributor may have to wait for any other here: every workgroup of the launch
// is resident (grid <= CUs, one per CU) and the spin is bounded.
template <int ABL>
__device__ __forceinline__ void sk_reduce_distributed(const SkSeg& sk, int j, int C, int uf, int NT_tile, int wave, int lane,
                                                      char* smem, uint16_t* __restrict__ y, int M, int N, int m0, int n0) {
    const int lo = j * NT_tile, self = sk.u - uf;
    // only the FIRST contributor's range can have started before the tile (its piece here is then its second: slot 1)
    const int first_slot = sk_bound(uf, sk.S, sk.units) >= lo ? 0 : 1;
    int* cw = sk.cnt + (j * 8 + sk.e) * N_MMA;
    char* stage = smem + wave * WSLOT;               // room for NJ contributors' four fragments per batch
    for (int t = self * N_MMA + wave; t < NJ * N_MMA; t += C * N_MMA) {
        const int ws = t / NJ, jj = t % NJ;
        bool arrived = false;
        int seen = 0;
        for (int spin = 0; spin < sk_spin_bound<ABL>(); ++spin) {
            int v = 0;
            if (lane == 0) v = __hip_atomic_load(cw + ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = __builtin_amdgcn_readfirstlane(v);
            if (seen >= NT_tile) { arrived = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // compiler ordering only: the slot loads are sc1
        sk_stamp<ABL>(sk.cnt, wave, lane, 4);
        f32x4 c4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (!arrived) {   // the wait gave up: flag the workspace, poison the token block (never a silent partial sum)
            sk_fail(sk.cnt, 2, j, seen, lane);
            const float qnan = __builtin_nanf("");
#pragma unroll
            for (int i = 0; i < 4; ++i) c4[i] = f32x4{qnan, qnan, qnan, qnan};
            store_block_xpose(c4, y, M, N, m0, n0, ws / WGN, ws % WGN, jj, lane & 15, lane >> 4);
            continue;
        }
        int staged = 0;
        auto flush = [&]() {                                     // (sums start from +0: unit order, fixed)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int b = 0; b < staged; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) c4[i] = c4[i] + *(const f32x4*)(stage + b * 4096 + i * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next batch's DMA overwrites the stage
            staged = 0;
        };
        for (int k = 0; k < C; ++k) {
            const int v = uf + k;
            const float* src = sk.ws + ((int64_t)((v * 8 + sk.e) * 2 + (k == 0 ? first_slot : 0)) * (BM * BN)) + ws * (WSLOT / 4);
            const rsrc_t r = make_rsrc(src, (uint32_t)WSLOT);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(stage + staged * 4096 + i * 1024),
                                                         16, (uint32_t)lane * 16u, (uint32_t)(i * NJ + jj) * 1024u, 0, 16);
            if (++staged == NJ) flush();
        }
        if (staged) flush();
        store_block_xpose(c4, y, M, N, m0, n0, ws / WGN, ws % WGN, jj, lane & 15, lane >> 4);
    }
    sk_stamp<ABL>(sk.cnt, wave, lane, 5);
    // this wave is through with the tile; the last of the contributors' 8 C waves re-zeroes its counters for the next launch
    if (lane == 0) {
        int* done = sk.cnt + SK_DONE_OFF + j * 8 + sk.e;
        if (__hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == C * N_MMA - 1) {
#pragma unroll
            for (int w = 0; w < N_MMA; ++w) __hip_atomic_store(cw + w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// a parked piece's K-steps enter its tile's count (pre: the slot stores have retired -- the caller's vmcnt wait)
template <int ABL = 0>
__device__ __forceinline__ void sk_bump_pending(SkSeg& sk, int wave, int lane) {
    if (sk.pend_j >= 0) {
        if (lane == 0 && !((ABL & EXP_SK_WITHHOLD) && sk.u == 0))   // (fault injection: unit 0's counts never arrive)
            __hip_atomic_fetch_add(sk.cnt + (sk.pend_j * 8 + sk.e) * N_MMA + wave, sk.pend_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sk.pend_j = -1;
    }
}

// The owner of tile sk.j (the unit whose piece holds the tile's last K-step; `own` steps, accumulators in registers): wait
// until the wave's counter shows the other NT_tile - own steps, add the contributors' slots in unit order, write y and
// re-zero the counter.  Contributors are units with a lower index; their piece of this tile is the FIRST thing they run.
template <int ABL>
__device__ __forceinline__ void sk_finish_owner(const SkSeg& sk, int own, int NT_tile, int wave, int lane, char* smem,
                                                AccT& acc, uint16_t* __restrict__ y, int M, int N, int m0, int n0) {
    const int j = sk.j, lo = j * NT_tile;
    int* c = sk.cnt + (j * 8 + sk.e) * N_MMA + wave;
    const int need = NT_tile - own;
    sk_stamp<ABL>(sk.cnt, wave, lane, 7);   // (marks the workgroup as an owner)
    bool arrived = false;
    int seen = 0;
    for (int spin = 0; spin < sk_spin_bound<ABL>(); ++spin) {
        int v = 0;
        if (lane == 0) v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = __builtin_amdgcn_readfirstlane(v);
        if (seen >= need) { arrived = true; break; }
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // compiler ordering only: the slot loads are agent-scope themselves
    sk_stamp<ABL>(sk.cnt, wave, lane, 2);
    if (!arrived) {   // the wait gave up: flag the workspace and write the tile as NaN (never a silent partial sum)
        sk_fail(sk.cnt, 1, j, seen, lane);
        const float qnan = __builtin_nanf("");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = f32x4{qnan, qnan, qnan, qnan};
        store_tile_xpose(acc, y, M, N, m0, n0, wave / WGN, wave % WGN, lane & 15, lane >> 4);
        return;
    }
    int uf;
    sk_contributors(sk, j, NT_tile, uf);
    // A contributor's slot comes through LDS, not through registers: 16 LDS-DMA pieces (sc1: the bytes were written by
    // another XCD) into this wave's 16 KB of the idle rings, all in flight together = ONE loaded-memory round trip per
    // slot and no second accumulator set (64 more VGPRs spilled at the 168-register cap).  The rings are idle: this is the
    // unit's last segment, every LDS read of it lies before its last barrier, and the dequant waves have passed theirs.
    char* stage = smem + wave * WSLOT;
    for (int v = uf; v < sk.u; ++v) {
        const int vb = sk_bound(v, sk.S, sk.units);
        if (sk_bound(v + 1, sk.S, sk.units) <= (vb > lo ? vb : lo)) continue;   // empty range: no slot was written
        const float* src = sk.ws + ((int64_t)((v * 8 + sk.e) * 2 + (vb >= lo ? 0 : 1)) * (BM * BN)) + wave * (WSLOT / 4);
        const rsrc_t r = make_rsrc(src, (uint32_t)WSLOT);
#pragma unroll
        for (int f = 0; f < FRAGS; ++f)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(stage + f * 1024), 16,
                                                     (uint32_t)lane * 16u, (uint32_t)f * 1024u, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = acc[i][jj] + *(const f32x4*)(stage + (i * NJ + jj) * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next slot's DMA overwrites the stage
    }
    sk_stamp<ABL>(sk.cnt, wave, lane, 3);
    if (lane == 0)   // ready for the next launch
        __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    store_tile_xpose(acc, y, M, N, m0, n0, wave / WGN, wave % WGN, lane & 15, lane >> 4);
}

// One segment = NT K-steps of one tile.  pre: its prologue DMAs are already in flight (issued by the caller behind
// the previous segment's last barrier; other VMEM traffic of this wave -- the previous tile's output stores -- may
// sit in between, so the first wait is a full one).  After the last barrier, BEFORE the final 16 MFMAs and the
// output, `next(...)` runs: the persistent loop issues the next tile's prologue DMAs there, which then fly under
// this tile's epilogue.
template <int ABL, bool DENSE, class Next>
__device__ __forceinline__ void mma_segment(char* smem, int wave, int lane, int NT, const XDma& xd, bool pre,
                                            uint16_t* __restrict__ y, int M, int N, int m0, int n0, int NT_tile,
                                            SkSeg& sk, Next&& next) {
    const int wm = wave / WGN, wn = wave % WGN, fr = lane & 15, fq = lane >> 4;
    AccT acc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag4 wf0, wf1;
    FragX xf0, xf1;

    if (!pre) mma_prologue_issue<ABL>(xd, smem, wave, NT);
    if (NT > 1 && !pre) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // vmcnt retires in order: the wait for this segment's first x tile has also retired every older operation of the
    // wave -- the slot stores of the piece it parked before.  Its K-step count moves now, for free.
    sk_bump_pending<ABL>(sk, wave, lane);
    __builtin_amdgcn_s_barrier();   // prologue barrier 1: x tile 0 and packed blocks 0..3 landed
    __builtin_amdgcn_s_barrier();   // prologue barrier 2: W16(0) written by the dequant waves

    // step 0: no previous half
    load_frags<DENSE>(smem, 0, 0, wm, wn, fr, fq, wf0, xf0);
    load_frags<DENSE>(smem, 0, 1, wm, wn, fr, fq, wf1, xf1);
    if (NT > 2) {
        if constexpr (!(ABL & ABL_NO_XDMA)) issue_x<0, 4>(xd, smem, wave, 2);
        mfma_rows<0, 4, ABL & ~192>(acc, wf0, xf0);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else {
        mfma_rows<0, 4, ABL & ~192>(acc, wf0, xf0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    int t = 1;
    Stamps st = {0, 0, 0, 0};
    u64t rt0 = 0;
    if constexpr (MXQ_STAMPS(ABL)) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
    for (; t + 2 < NT; ++t) mma_step<ABL, true, DENSE>(smem, t, wave, wm, wn, fr, fq, xd, acc, wf0, xf0, wf1, xf1, st);
    if constexpr (MXQ_STAMPS(ABL)) {   // this workgroup's wave: {work, wait, barrier, steps} cycle sums
        u64t rt1;                              // + the same span on the constant 100 MHz clock (high half of word 3): the core clock held
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
        if (lane == 0 && sk.ws) {
            u64t* d = (u64t*)sk.ws + ((int64_t)blockIdx.x * (N_MMA + N_DEQ) + wave) * 4;
            d[0] = st.work; d[1] = st.wait; d[2] = st.bar; d[3] = st.n | ((rt1 - rt0) << 32);
        }
    }
    for (; t < NT; ++t) mma_step<ABL, false, DENSE>(smem, t, wave, wm, wn, fr, fq, xd, acc, wf0, xf0, wf1, xf1, st);
    next();                                  // the ring is idle from here on
    mfma_rows<0, 4, ABL & ~192>(acc, wf1, xf1);     // (NT-1, kk=1)
    // (three ways out from here -- store, park, owner's reduction: without this pin the register allocator renames the
    //  accumulators across the last 16 MFMAs to suit one of them and spills 28 registers)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(acc[i][j]));

    if (NT != NT_tile) {
        sk_stamp<ABL>(sk.cnt, wave, lane, 1);
        if (sk.owner && !sk.dist) {   // the piece that ends its tile: the others' pieces were parked long ago (they run first)
            sk_finish_owner<ABL>(sk, NT, NT_tile, wave, lane, smem, acc, y, M, N, m0, n0);
            return;
        }
        // a piece that starts its tile: park the accumulators in this unit's slot; the count moves once the stores retired
        float* mine = (sk.park ? sk.park : sk.ws + (int64_t)((sk.u * 8 + sk.e) * 2 + (sk.first ? 0 : 1)) * (BM * BN)) + wave * (WSLOT / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) st_agent(mine, i * NJ + j, lane, acc[i][j]);
        sk.pend_j = sk.park ? -1 : sk.j;
        sk.pend_n = NT;
        sk_stamp<ABL>(sk.cnt, wave, lane, 2);
        return;
    }
    if constexpr (!(ABL & ABL_NO_STORE)) {
        store_tile_xpose(acc, y, M, N, m0, n0, wm, wn, fr, fq);
    } else {   // keep every accumulator alive without writing the tile
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) s_ += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (s_ == 123.456f) y[0] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// dequant waves
// ------------------------------------------------------------------------------------------------
struct Deq {
    char* smem;
    rsrc_t rsrc;          // the WHOLE packed weight (one descriptor per launch: a per-tile one would have to be
                          // selected between tiles, and a selected descriptor lives in VGPRs -- a waterfall loop around
                          // every load).  Row-blocks are the outermost dimension, so a row-block beyond the weight's
                          // last one lies beyond the buffer: zeros, whatever the K offset
    uint32_t voff_blk;    // byte offset of the thread's row-block inside the packed weight; 0x80000000 = nothing to
                          // load (the range check is on this offset; launcher: the weight is smaller than 2^31 bytes)
    uint32_t k0;          // byte offset of the segment's first K-step inside a row-block's run
    int d, lane, NT;
    int row, r, h;        // W row of this thread (0..127), its row inside the block, column half (wave-uniform)
    float s4, z4;
    float4 rm;            // the row's 4-bit-arm parameters as loaded (rowmeta)
#ifdef MXQ_G8_AWQ
    // AWQ layout: the thread converts (channel octet co, k octet ko) of the 64 x 128 weight tile for DEQ_NRES channels of the
    // octet (convert_pk: which ones).
    // voff_blk = byte offset of word (k = 8 ko, column n0 / 8 + co) from row k = 0; k0 = the segment's first K-STEP.
    uint32_t zoff, soff;  // byte offsets of the thread's zero word / first scale pair inside a group's row
    uint32_t raw_voff;    // LDS-staged words (MXQ_AWQ_RAW): byte offset of the 16 bytes this lane's DMA fetches, from row k = 0
    uint32_t raw_rd;      // ... and of the thread's word of row k = 8 ko inside a ring slot
    int co, ko, sh;       // sh = X * 0x00010001: added to the v_perm selectors that pick the thread's byte X of a word
#endif
};

__device__ __forceinline__ void put8(char* wt, int row, int slot, const uint32_t* o) {
    *(u32x4*)(wt + swz(row, slot)) = (u32x4){o[0], o[1], o[2], o[3]};
}

// The packed words one thread needs for one chunk.  h = 0: 2-bit groups 0, 1 (columns 0..31); h = 1: group 2 and the
// 4-bit quarter (columns 32..63).  W2G16: h = 0 groups 0, 1; h = 1 groups 2, 3.  W4ROW: 4 code words each.
struct Pk {
    uint32_t c[4];    // code words
    uint32_t z[2];    // 2-bit zero-points (fp32 bits; compact metadata: the fp16 halfword until widen_pk)
    uint32_t scw;     // the row's scale codes
    f32x2 qq[2];      // (qs, qz) of the thread's 2-bit groups
};
struct Pk4 {          // W4ROW: code words only (scale / zero come from rowmeta)
    uint32_t c[4];
};
struct PkA {          // AWQ layout: 8 code words (k = 8 ko .. 8 ko + 7 of one channel octet), its group's zero word and scale pairs
    uint32_t q[8];
    uint32_t z;
    uint32_t s[2];    // scale dwords: channels (c0, c0 + 1) and (c0 + 2, c0 + 3) of the thread's first channel c0
};
template <int LAYOUT>
struct PkOf { typedef Pk type; };
template <>
struct PkOf<MXQ_LAYOUT_W4ROW> { typedef Pk4 type; };
template <>
struct PkOf<LAYOUT_AWQ> { typedef PkA type; };

// They come STRAIGHT from global memory into registers, three K-steps before they are used (6-7 dword loads per
// thread and chunk through the tile's buffer descriptor; 16 lanes of a row-block read 64 consecutive bytes).  Round 2
// first had them copied into a 4-slot LDS ring by LDS-DMA and read back from there: the two DMA pieces per wave and
// step cost their issuer 100-185 cycles apiece next to MFMAs, on the one wave per SIMD whose ~90-op chain is the
// critical path of a K-step (-2..4 % per launch without them; bit-identical results).
template <int LAYOUT, int H, bool NOQ4 = false, int PARTS = DEQ_PARTS>   // (PARTS a parameter: the other mode's branches are not instantiated)
__device__ __forceinline__ void load_pk(const Deq& c, int t, typename PkOf<LAYOUT>::type& k G8_AWQU_PARAM) {
#ifdef MXQ_G8_AWQ
    if constexpr (LAYOUT == LAYOUT_AWQ) {
        // K-step c.k0 + t, clamped to the tensor (a burst loads whole groups of R chunks: the ones past the segment's end are
        // never written to LDS, but their loads must stay inside the allocation -- the K offset rides in the scalar offset,
        // which a raw buffer does not range-check)
        int kt = __builtin_amdgcn_readfirstlane((int)c.k0 + t);
        kt = kt < au.last_kt ? kt : au.last_kt;
        const uint32_t kbase = (uint32_t)kt * BK;
        const uint32_t so = kbase * au.row_bytes;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            k.q[i] = __builtin_amdgcn_raw_buffer_load_b32(au.rs_q, c.voff_blk, so + (uint32_t)i * au.row_bytes, 0);
        const uint32_t g = __umulhi(kbase + 8u * (uint32_t)c.ko, au.gmul);          // group of this thread's k octet
        k.z = __builtin_amdgcn_raw_buffer_load_b32(au.rs_z, c.zoff + g * au.row_bytes, 0, 0);
        if constexpr (DEQ_NRES == 2) {   // channels c0 and c0 + 2: c0 may be odd -- halfword loads (a dword load would be misaligned)
            k.s[0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(au.rs_s, c.soff + g * (4u * au.row_bytes), 0, 0);
            k.s[1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(au.rs_s, c.soff + g * (4u * au.row_bytes) + 4u, 0, 0);
        } else {
            k.s[0] = __builtin_amdgcn_raw_buffer_load_b32(au.rs_s, c.soff + g * (4u * au.row_bytes), 0, 0);
            k.s[1] = __builtin_amdgcn_raw_buffer_load_b32(au.rs_s, c.soff + g * (4u * au.row_bytes) + 4u, 0, 0);
        }
        return;
    } else {
#endif
    constexpr int BYTES = LAYOUT == MXQ_LAYOUT_W4ROW ? 512 : LAYOUT == MXQ_LAYOUT_MIXEDC ? MXQC_BLK_BYTES : MXQ_BLK_BYTES;
    // wave-uniform by construction; said explicitly, or a K offset selected between two tiles' descriptors counts as
    // divergent and every load below gets a waterfall loop around it
    const uint32_t so = __builtin_amdgcn_readfirstlane(c.k0 + (uint32_t)t * BYTES);
    auto dw = [&](int idx) { return __builtin_amdgcn_raw_buffer_load_b32(c.rsrc, c.voff_blk + (uint32_t)idx * 4u, so, 0); };
    auto hw = [&](int idx) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(c.rsrc, c.voff_blk + (uint32_t)idx * 2u, so, 0); };
    if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
        if constexpr (PARTS == 4) {
            k.c[0] = dw(mxq_w4_c4(H, 0, c.r));
            k.c[1] = dw(mxq_w4_c4(H, 1, c.r));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) k.c[i] = dw(mxq_w4_c4(H * 2 + (i >> 1), i & 1, c.r));
        }
    } else if constexpr (PARTS == 4) {
        // quarter H: 2-bit group H (mixed: H < 3; W2G16: any H), or the 4-bit quarter (mixed, H = 3)
        constexpr bool MIXED = LAYOUT == MXQ_LAYOUT_MIXED || LAYOUT == MXQ_LAYOUT_MIXEDC, COMPACT = LAYOUT == MXQ_LAYOUT_MIXEDC;
        if constexpr (MIXED && H == 3) {
            if constexpr (!NOQ4) {
                k.c[2] = dw(mxq_c4(0, c.r));
                k.c[3] = dw(mxq_c4(1, c.r));
            }
        } else {
            constexpr int QQ0 = COMPACT ? MXQC_OFF_QQ : MXQ_OFF_QQ;
            k.scw = hw(COMPACT ? mxqc_sc_u16(c.r) : mxq_sc_u16(c.r));
            k.c[0] = dw((MIXED ? mxq_c2(0, c.r) : mxq_w2_c2(0, c.r)) + H * 16);
            if constexpr (COMPACT) k.z[0] = hw(mxqc_z2_u16(0, c.r) + H * 16);
            else k.z[0] = dw((MIXED ? mxq_z2(0, c.r) : mxq_w2_z2(0, c.r)) + H * 16);
            k.qq[0] = (f32x2){__uint_as_float(dw(QQ0 + H * 2)), __uint_as_float(dw(QQ0 + H * 2 + 1))};
        }
    } else {
    constexpr bool MIXED = LAYOUT == MXQ_LAYOUT_MIXED || LAYOUT == MXQ_LAYOUT_MIXEDC, COMPACT = LAYOUT == MXQ_LAYOUT_MIXEDC;
    k.scw = hw(COMPACT ? mxqc_sc_u16(c.r) : mxq_sc_u16(c.r));
    const int g0 = H * 2;   // first 2-bit group of this thread
    auto zero_of = [&](int g) -> uint32_t {
        if constexpr (COMPACT) return hw(mxqc_z2_u16(0, c.r) + g * 16);
        else return dw((MIXED ? mxq_z2(0, c.r) : mxq_w2_z2(0, c.r)) + g * 16);
    };
    constexpr int QQ0 = COMPACT ? MXQC_OFF_QQ : MXQ_OFF_QQ;
    k.c[0] = dw((MIXED ? mxq_c2(0, c.r) : mxq_w2_c2(0, c.r)) + g0 * 16);
    k.z[0] = zero_of(g0);
    k.qq[0] = (f32x2){__uint_as_float(dw(QQ0 + g0 * 2)), __uint_as_float(dw(QQ0 + g0 * 2 + 1))};
    if (LAYOUT == MXQ_LAYOUT_W2G16 || H == 0) {
        k.c[1] = dw((MIXED ? mxq_c2(1, c.r) : mxq_w2_c2(1, c.r)) + g0 * 16);
        k.z[1] = zero_of(g0 + 1);
        k.qq[1] = (f32x2){__uint_as_float(dw(QQ0 + g0 * 2 + 2)), __uint_as_float(dw(QQ0 + g0 * 2 + 3))};
    } else if constexpr (!NOQ4) {
        k.c[2] = dw(mxq_c4(0, c.r));
        k.c[3] = dw(mxq_c4(1, c.r));
    }
    }
#ifdef MXQ_G8_AWQ
    }
#endif
}

template <int LAYOUT>
__device__ __forceinline__ void widen_pk(typename PkOf<LAYOUT>::type& k) {   // compact zero-points arrive as fp16 halfwords
    if constexpr (LAYOUT == MXQ_LAYOUT_MIXEDC) {
        k.z[0] = __float_as_uint((float)__builtin_bit_cast(_Float16, (uint16_t)k.z[0]));
        k.z[1] = __float_as_uint((float)__builtin_bit_cast(_Float16, (uint16_t)k.z[1]));
    }
}

// chunk t: preloaded packed words -> fp16 W16[t & 1]
// chunk's packed words -> the thread's 32 fp16 weights (4 x 16 bytes: W16 slots s0 .. s0+3 of its row)
template <int LAYOUT, int H, bool NOQ4 = false, int PARTS = DEQ_PARTS>
__device__ __forceinline__ void convert_pk(const Deq& c, const typename PkOf<LAYOUT>::type& k, u32x4 (&res)[8 / PARTS]) {
#ifdef MXQ_G8_AWQ
    if constexpr (LAYOUT == LAYOUT_AWQ) {
        // The reference's arithmetic (dequantize.cuh:15-78, gemm_cuda_gen.cu:134-141) in packed fp16 -- a 4-bit field OR-ed into
        // 0x6400 reads 1024 + q (low nibble of a byte) or 1024 + 16 q (high nibble); (1024 + q) - (1024 + z) and
        // (1024 + 16 q) / 16 - (64 + z) are both q - z EXACTLY, and the product with the scale is rounded ONCE -- but packed
        // over TWO CONSECUTIVE k OF ONE CHANNEL instead of the reference's two channels at one k: one v_perm_b32 puts the
        // byte that holds a channel's nibble in words k and k + 1 side by side, the masked result IS the output dword, and the
        // K-major -> [channel][8 k] transpose costs nothing further.  A word's byte b holds channels (0,2 | 4,6 | 1,3 | 5,7)[b] in
        // its (low, high) nibbles (dequantize.cuh:35-51); the thread's bytes are a per-lane v_perm selector (c.sh), not a code path.
        // 4 channels per thread: bytes (hq, 2 + hq) -> channels 4 hq + (0, 1, 2, 3) = (X.lo, Y.lo, X.hi, Y.hi);
        // 2 channels per thread: byte hq -> channels (c0, c0 + 2) = (X.lo, X.hi).
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        constexpr uint32_t LO = 0x000f000fu, HI = 0x00f000f0u;
        // (x & mask) | 0x64006400 as ONE v_and_or_b32: a VOP3 instruction takes no literal on gfx9, and left to itself the
        // compiler emits v_and_b32 + v_or_b32 with a literal each -- 40 extra VALU ops per chunk on the critical wave.  The
        // mask rides in an SGPR, the magic constant in a VGPR.
        uint32_t magic;
        asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));
        auto and_or = [&](uint32_t x, uint32_t mask) {
            uint32_t r;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(magic));
            return r;
        };
        constexpr int NB = DEQ_NRES / 2;                       // bytes per thread: 2 (X, Y) or 1 (X)
        const h2 sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f}, zero2 = {(_Float16)0.f, (_Float16)0.f};
        const uint32_t selx = 0x0c040c00u + (uint32_t)c.sh;    // [lo.byte X, 0, hi.byte X, 0]; c.sh = X * 0x00010001
        h2 zlo[NB], zhi[NB], slo[NB], shi[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t zz = __builtin_amdgcn_perm(k.z, k.z, selx - 0x00040000u + 0x00020002u * b);       // [z.byte, 0, z.byte, 0]
            zlo[b] = __builtin_bit_cast(h2, and_or(zz, LO));                                           // 1024 + z
            zhi[b] = __builtin_elementwise_fma(__builtin_bit_cast(h2, and_or(zz, HI)), sixteenth, zero2);   // 64 + z, exact
        }
        if constexpr (NB == 2) {
            slo[0] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(k.s[0], k.s[0], 0x01000100u));   // channel c0 (X.lo)
            slo[1] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(k.s[0], k.s[0], 0x03020302u));   // c0 + 1 (Y.lo)
            shi[0] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(k.s[1], k.s[1], 0x01000100u));   // c0 + 2 (X.hi)
            shi[1] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(k.s[1], k.s[1], 0x03020302u));   // c0 + 3 (Y.hi)
        } else {
            slo[0] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(k.s[0], k.s[0], 0x01000100u));   // channel c0 (X.lo)
            shi[0] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(k.s[1], k.s[1], 0x01000100u));   // c0 + 2 (X.hi)
        }
        uint32_t o[DEQ_NRES][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const uint32_t xx = __builtin_amdgcn_perm(k.q[2 * m + 1], k.q[2 * m], selx + 0x00020002u * b);
                const h2 ql = __builtin_bit_cast(h2, and_or(xx, LO)), qh = __builtin_bit_cast(h2, and_or(xx, HI));
                o[b][m] = __builtin_bit_cast(uint32_t, (ql - zlo[b]) * slo[b]);
                o[NB + b][m] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(qh, sixteenth, -zhi[b]) * shi[b]);
            }
#pragma unroll
        for (int i = 0; i < DEQ_NRES; ++i) res[i] = (u32x4){o[i][0], o[i][1], o[i][2], o[i][3]};
        return;
    } else {
#endif
    uint32_t o[8];
    if constexpr (PARTS == 4) {
        constexpr bool MIXED = LAYOUT == MXQ_LAYOUT_MIXED || LAYOUT == MXQ_LAYOUT_MIXEDC;
        if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
            mxq_deq4x8(k.c[0], c.s4, c.z4, o);
            mxq_deq4x8(k.c[1], c.s4, c.z4, o + 4);
        } else if constexpr (MIXED && H == 3) {
            if constexpr (!NOQ4) {
                mxq_deq4x8(k.c[2], c.s4, c.z4, o);
                mxq_deq4x8(k.c[3], c.s4, c.z4, o + 4);
            }
        } else {
            mxq_deq2x16(k.c[0], mxq_scale(k.qq[0][0], k.qq[0][1], (k.scw >> (4 * H)) & 15u), __uint_as_float(k.z[0]), o);
        }
        res[0] = (u32x4){o[0], o[1], o[2], o[3]};
        res[1] = (u32x4){o[4], o[5], o[6], o[7]};
    } else if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            mxq_deq4x8(k.c[2 * q], c.s4, c.z4, o);
            mxq_deq4x8(k.c[2 * q + 1], c.s4, c.z4, o + 4);
            res[2 * q] = (u32x4){o[0], o[1], o[2], o[3]};
            res[2 * q + 1] = (u32x4){o[4], o[5], o[6], o[7]};
        }
    } else {
        const int g0 = H * 2;
        mxq_deq2x16(k.c[0], mxq_scale(k.qq[0][0], k.qq[0][1], (k.scw >> (4 * g0)) & 15u), __uint_as_float(k.z[0]), o);
        res[0] = (u32x4){o[0], o[1], o[2], o[3]};
        res[1] = (u32x4){o[4], o[5], o[6], o[7]};
        if (LAYOUT == MXQ_LAYOUT_W2G16 || H == 0) {
            mxq_deq2x16(k.c[1], mxq_scale(k.qq[1][0], k.qq[1][1], (k.scw >> (4 * g0 + 4)) & 15u), __uint_as_float(k.z[1]), o);
        } else if constexpr (!NOQ4) {
            mxq_deq4x8(k.c[2], c.s4, c.z4, o);
            mxq_deq4x8(k.c[3], c.s4, c.z4, o + 4);
        }
        res[2] = (u32x4){o[0], o[1], o[2], o[3]};
        res[3] = (u32x4){o[4], o[5], o[6], o[7]};
    }
#ifdef MXQ_G8_AWQ
    }
#endif
}
// ... into W16[t & 1]: the thread's column half H = slots 4 H .. 4 H + 3
template <int H, int LAYOUT = MXQ_LAYOUT_MIXED>
__device__ __forceinline__ void store_pk(const Deq& c, int t, const u32x4 (&res)[DEQ_NRES]) {
    char* wt = c.smem + OFF_W + (t & 1) * W_STAGE;
#ifdef MXQ_G8_AWQ
    if constexpr (LAYOUT == LAYOUT_AWQ) {     // c.row = the thread's first channel; its k octet is 16-byte slot ko of each row
#pragma unroll
        for (int i = 0; i < DEQ_NRES; ++i) *(u32x4*)(wt + swz(c.row + (DEQ_NRES == 2 ? 2 * i : i), c.ko)) = res[i];
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < DEQ_NRES; ++i) *(u32x4*)(wt + swz(c.row, H * DEQ_NRES + i)) = res[i];
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
#ifdef MXQ_G8_AWQ
    if constexpr (LAYOUT == LAYOUT_AWQ) {
        // thread -> (k octet ko = low 3 lane bits: the 8 lanes of one 16-byte-store group write 8 different slots of rows with
        // equal n & 7, i.e. 8 different bank groups; pair group hq; channel octet co)
        constexpr int TPB = N_DEQ / 2;                         // threads per (co, ko) block: 2 (4 channels each) or 4 (2 each)
        const int dt = c.d * 64 + lane;
        c.ko = dt & 7;
        const int hq = (dt >> 3) & (TPB - 1);                  // = the word byte X the thread's channels sit in
        c.co = dt / (8 * TPB);
        c.sh = hq * 0x00010001;
        const int c0 = TPB == 2 ? 4 * hq : (hq & 1) * 4 + (hq >> 1);   // the thread's first channel inside the octet
        c.h = 0;
        c.row = 8 * c.co + c0;
        c.r = 0;
        c.rsrc = make_rsrc(qweight, 0u);       // (the MXQ layouts' fields: defined, so that copies of a Deq stay in registers)
        c.s4 = c.z4 = 0.f;
        c.rm = float4{0.f, 0.f, 0.f, 0.f};
        const uint32_t row_bytes = (uint32_t)N / 2u;
        const int OC8 = N >> 3;
        int col = (n0 >> 3) + c.co;
        col = col < OC8 ? col : OC8 - 1;       // (a tile past the last channel re-reads the last octet; never stored)
        c.voff_blk = (uint32_t)(8 * c.ko) * row_bytes + (uint32_t)col * 4u;
        {   // DMA lane L of dequant wave d fetches chunk (row 8 ko' + i', column quad q') into slot position (i' 32 + q' 8 + ko') * 16:
            // for a fixed row-in-octet and quad, the 8 k octets sit in 8 different 16-byte bank groups
            const int kd = lane & 7, qd = (lane >> 3) & 3, id = 2 * c.d + (lane >> 5);
            int o0 = (n0 >> 3) + 4 * qd;
            o0 = o0 < OC8 - 4 ? o0 : OC8 - 4;  // (a quad past the last channel re-reads the last one: in bounds, never stored)
            c.raw_voff = (uint32_t)(8 * kd + id) * row_bytes + (uint32_t)o0 * 4u;
            c.raw_rd = (uint32_t)((c.co >> 2) * 128 + c.ko * 16 + (c.co & 3) * 4);
        }
        c.zoff = (uint32_t)col * 4u;
        c.soff = (uint32_t)col * 16u + (uint32_t)c0 * 2u;
        c.k0 = (uint32_t)kt0;
        return;
    }
#endif
    const uint32_t blk_stride = (uint32_t)NT_tile * BLK_B;   // bytes between consecutive row-blocks
    c.rsrc = make_rsrc(qweight, (uint32_t)(N >> 4) * blk_stride);
    c.k0 = (uint32_t)kt0 * BLK_B;
    const int dt = c.d * 64 + lane;   // 0 .. 64 N_DEQ - 1
    c.row = dt % BN;
    c.h = __builtin_amdgcn_readfirstlane(dt / BN);   // wave-uniform part: dequant waves 0,1 -> 0; 2,3 -> 1; (4,5 -> 2; 6,7 -> 3)
    c.r = c.row & 15;
    c.voff_blk = (uint32_t)((n0 >> 4) + (c.row >> 4)) * blk_stride;
}
__device__ __forceinline__ void deq_none(Deq& c, const Deq& like) {   // a descriptor whose every load is out of range
    c = like;
    c.voff_blk = 0x80000000u;
    c.k0 = 0;
#ifdef MXQ_G8_AWQ
    c.zoff = c.soff = c.raw_voff = 0x80000000u;
#endif
}

// One segment on the dequant waves, R chunks at a time ("group").  A BURST waits for the group's R register sets
// (loaded during the previous group's steps), converts all of them into result registers, and at once issues the loads
// of the next group -- of this segment, or, after its last group, group 0 of `nxt` (the next tile of the persistent
// loop, or nothing).  The following R K-steps only write one chunk's results into the W16 double buffer and meet the
// barrier.  Why bursts instead of ~90 VALU ops in every step: next to MFMAs a VALU op of this wave issues at ~8-10
// cycles, and the same ops interleaved into the MFMA waves' issue slots slow those down; in a burst step the MFMA
// waves finish their step and wait at the barrier while the rest of the burst runs at full rate, and the R-1 steps
// after it carry no VALU work at all (measured +3 % over the same work spread evenly, R = 3).
// pre: the sets already hold / are loading chunks 0 .. R-1 (issued by the previous tile's last burst).
template <int ABL, int LAYOUT, int H, int R>
__device__ __forceinline__ void deq_segment_h(Deq& c, const Deq& nxt, const float4* __restrict__ rowmeta, int N, int n0,
                                              bool pre, typename PkOf<LAYOUT>::type (&S)[R] G8_AWQU_PARAM) {
    const int NT = c.NT;
    u32x4 res[R][DEQ_NRES];
    auto load_group = [&](int base) {          // chunks base .. base+R-1 of this segment, or group 0 of nxt past its end
        if constexpr (ABL & ABL_NO_DEQ) return;
        const bool over = base >= NT;
        Deq d = c;
        d.voff_blk = over ? nxt.voff_blk : c.voff_blk;
        d.k0 = over ? nxt.k0 : c.k0;
#ifdef MXQ_G8_AWQ
        d.zoff = over ? nxt.zoff : c.zoff;
        d.soff = over ? nxt.soff : c.soff;
#endif
        const int b0 = over ? 0 : base;
#pragma unroll
        for (int i = 0; i < R; ++i) load_pk<LAYOUT, H, (ABL & ABL_NO_Q4) != 0>(d, b0 + i, S[i] G8_AWQU_ARG);
    };
    auto burst = [&](int base) {               // convert the loaded group, then fetch the one after it
        if constexpr (!(ABL & ABL_NO_DEQ)) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if constexpr ((ABL & ABL_THIRD_DEQ) != 0 && LAYOUT != LAYOUT_AWQ) {
                    if (i > 0 && base > 0) {   // (the first burst converts all three: real weights stay in res)
                        asm volatile("" ::"v"(S[i].c[0]), "v"(S[i].c[1]));
                        continue;
                    }
                }
                widen_pk<LAYOUT>(S[i]);
                convert_pk<LAYOUT, H, (ABL & ABL_NO_Q4) != 0>(c, S[i], res[i]);
            }
        }
        load_group(base + R);
    };
    auto put = [&](int q, const u32x4 (&r4)[DEQ_NRES]) {   // chunk q -> W16[q & 1], then the step's barrier
        if constexpr (!(ABL & ABL_NO_DEQ)) store_pk<H, LAYOUT>(c, q, r4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (!pre) load_group(0);
    if constexpr (LAYOUT != LAYOUT_AWQ) {      // (the AWQ layout has no per-row metadata)
        int gn = n0 + c.row;
        gn = gn < N ? gn : N - 1;
        c.rm = rowmeta[gn];
    }
    __builtin_amdgcn_s_barrier();              // prologue barrier 1 (the MFMA waves' x tile 0 has landed)
    if constexpr (LAYOUT != LAYOUT_AWQ) {
        c.s4 = mxq_scale(c.rm.z, c.rm.w, (uint32_t)c.rm.y);
        c.z4 = c.rm.x;
    }
    burst(0);
    put(0, res[0]);                            // prologue barrier 2: W16(0) written
    // step t = q - 1 writes chunk q; the segment's last step (t = NT - 1) writes nothing
    for (int base = 0; base < NT; base += R) {
#pragma unroll
        for (int i = 1; i < R; ++i)
            if (base + i < NT) put(base + i, res[i]);
        if (base + R < NT) {
            burst(base + R);
            put(base + R, res[0]);
        }
    }
    __builtin_amdgcn_s_barrier();              // step NT - 1
}

#ifdef MXQ_AWQ_RAW
// deq_segment_h's twin for the AWQ layout with LDS-staged code words -- the same bursts, the same barriers (prologue 1 and 2,
// one per K-step).  A group's words are fetched by DMA right AFTER the barrier that follows the previous burst (every wave has
// finished reading the ring by then), land under the group's remaining steps, and are waited for (vmcnt(0), each wave its
// own) in front of the barrier that precedes the burst that reads them.  After the segment's last group: group 0 of `nxt`.
template <int ABL, int R>
__device__ __forceinline__ void awq_segment_raw(Deq& c, const Deq& nxt, bool pre, PkA (&S)[R], const AwqU& au) {
    const int NT = c.NT;
    u32x4 res[R][DEQ_NRES];
    char* raw = c.smem + OFF_RAW;
    auto issue_group = [&](int base) {
        const bool over = base >= NT;
        const uint32_t rv = over ? nxt.raw_voff : c.raw_voff, zo = over ? nxt.zoff : c.zoff, sof = over ? nxt.soff : c.soff;
        const int k0 = (int)(over ? nxt.k0 : c.k0) + (over ? 0 : base);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int kt = __builtin_amdgcn_readfirstlane(k0 + i);
            kt = kt < au.last_kt ? kt : au.last_kt;           // (chunks past the segment's end: in bounds, never converted into use)
            const uint32_t kbase = (uint32_t)kt * BK;
            bufdma16(au.rs_q, rv, kbase * au.row_bytes, raw + i * RAW_SLOT + c.d * 1024);
            const uint32_t g = __umulhi(kbase + 8u * (uint32_t)c.ko, au.gmul);
            S[i].z = __builtin_amdgcn_raw_buffer_load_b32(au.rs_z, zo + g * au.row_bytes, 0, 0);
            S[i].s[0] = __builtin_amdgcn_raw_buffer_load_b32(au.rs_s, sof + g * (4u * au.row_bytes), 0, 0);
            S[i].s[1] = __builtin_amdgcn_raw_buffer_load_b32(au.rs_s, sof + g * (4u * au.row_bytes) + 4u, 0, 0);
        }
    };
    auto burst = [&]() {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const char* w = raw + i * RAW_SLOT + c.raw_rd;
#pragma unroll
            for (int j = 0; j < 8; ++j) S[i].q[j] = *(const uint32_t*)(w + j * 512);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) convert_pk<LAYOUT_AWQ, 0, false>(c, S[i], res[i]);
    };
    auto put = [&](int q, const u32x4 (&r4)[DEQ_NRES], bool drain) {
        store_pk<0, LAYOUT_AWQ>(c, q, r4);
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (!pre) issue_group(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // prologue barrier 1: every wave's part of group 0 is in the ring
    burst();
    put(0, res[0], false);                     // prologue barrier 2: W16(0) written; the ring has been read
    for (int base = 0; base < NT; base += R) {
        issue_group(base + R);
#pragma unroll
        for (int i = 1; i < R; ++i)
            if (base + i < NT) put(base + i, res[i], i == R - 1);
        if (base + R < NT) {
            burst();
            put(base + R, res[0], false);
        }
    }
    __builtin_amdgcn_s_barrier();              // step NT - 1
}
// ... and for the small-tile builds (one chunk per K-step, no bursts; 8 dequant waves of which the first four carry the DMAs):
// chunk q lives in ring slot q % 3 and is fetched TWO steps before it is converted -- its DMA goes out right after the barrier
// that follows the conversion of chunk q - 3 + 2 ... i.e. at the top of iteration q - 2 (slot last read by iteration q - 3),
// and every wave waits for its own part of chunk q + 1 (vmcnt: this iteration's four operations may stay in flight) in front
// of the barrier of iteration q.  The group's scale / zero words come one step ahead, straight into registers.
template <int ABL>
__device__ __forceinline__ void awq_segment_raw1(Deq& c, const Deq& nxt, bool pre, PkA (&S)[1], const AwqU& au, int& raw_slot) {
    const int NT = c.NT;
    u32x4 res[DEQ_NRES];
    char* raw = c.smem + OFF_RAW;
    PkA nx = S[0];                             // (pre: the previous segment's last iteration loaded this segment's chunk 0 into S[0])
    auto kbase_of = [&](int q, bool& over) {   // first k row of chunk q of this segment / of chunk q - NT of nxt; clamped to the tensor
        over = q >= NT;
        int kt = __builtin_amdgcn_readfirstlane((int)(over ? nxt.k0 : c.k0) + (over ? q - NT : q));
        kt = kt < au.last_kt ? kt : au.last_kt;
        return (uint32_t)kt * BK;
    };
    auto issue_zs = [&](int q, PkA& k) {
        bool over;
        const uint32_t kbase = kbase_of(q, over);
        const uint32_t zo = over ? nxt.zoff : c.zoff, sof = over ? nxt.soff : c.soff;
        const uint32_t g = __umulhi(kbase + 8u * (uint32_t)c.ko, au.gmul);
        k.z = __builtin_amdgcn_raw_buffer_load_b32(au.rs_z, zo + g * au.row_bytes, 0, 0);
        if constexpr (DEQ_NRES == 2) {
            k.s[0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(au.rs_s, sof + g * (4u * au.row_bytes), 0, 0);
            k.s[1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(au.rs_s, sof + g * (4u * au.row_bytes) + 4u, 0, 0);
        } else {
            k.s[0] = __builtin_amdgcn_raw_buffer_load_b32(au.rs_s, sof + g * (4u * au.row_bytes), 0, 0);
            k.s[1] = __builtin_amdgcn_raw_buffer_load_b32(au.rs_s, sof + g * (4u * au.row_bytes) + 4u, 0, 0);
        }
    };
    auto issue_dma = [&](int q, int slot) {    // (every wave issues one operation: the waves beyond the fourth fetch nothing)
        bool over;
        const uint32_t kbase = kbase_of(q, over);
        const uint32_t rv = over ? nxt.raw_voff : c.raw_voff;
        if (c.d < 4) bufdma16(au.rs_q, rv, kbase * au.row_bytes, raw + slot * RAW_SLOT + c.d * 1024);
    };
    // slot of the chunk being converted; it runs on from segment to segment (the previous segment's last two iterations have
    // fetched this segment's chunks 0 and 1 into the slots that follow its own)
    int slot = raw_slot;
    if (!pre) {
        issue_zs(0, nx);
        issue_dma(0, slot);
        issue_dma(1, slot == 2 ? 0 : slot + 1);
    }
    // in flight at most: z/s(0), DMA(0), DMA(1) in this order (pre: z/s(0), DMA(1)) -- chunk 0's words have landed once at most
    // one operation is outstanding
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // prologue barrier 1
    for (int q = 0; q < NT; ++q) {
        PkA cur = nx;
        issue_zs(q + 1, nx);                   // 3 operations
        issue_dma(q + 2, slot == 0 ? 2 : slot - 1);   // slot (q + 2) % 3
        const char* w = raw + slot * RAW_SLOT + c.raw_rd;
#pragma unroll
        for (int j = 0; j < 8; ++j) cur.q[j] = *(const uint32_t*)(w + j * 512);
        convert_pk<LAYOUT_AWQ, 0, false>(c, cur, res);
        store_pk<0, LAYOUT_AWQ>(c, q, res);
        // this iteration's 4 operations may fly on; everything older -- chunk q + 1's words among it -- has landed
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // q = 0: prologue barrier 2; q >= 1: the barrier of step q - 1
        slot = slot == 2 ? 0 : slot + 1;
    }
    S[0] = nx;                                 // scale / zero words of the next segment's chunk 0
    raw_slot = slot;
    __builtin_amdgcn_s_barrier();              // step NT - 1
}
#endif

// the column half a dequant wave works on is wave-uniform but not a constant: dispatch once, outside the loops
#ifndef MXQ_DEQ_R
#define MXQ_DEQ_R (MXQ_G8_BM == 256 ? 3 : 1)
#endif
constexpr int DEQ_R = MXQ_DEQ_R;    // measured: R = 2 -5 %, 3 and 4 +2 % over per-step dequant; R = 4 spills at the 168-VGPR cap
template <int ABL, int LAYOUT>
__device__ __forceinline__ void deq_segment(Deq& c, const Deq& nxt, int wave, int lane, const float4* __restrict__ rowmeta,
                                            int N, int n0, bool pre, typename PkOf<LAYOUT>::type (&S)[DEQ_R] G8_AWQU_PARAM) {
    if constexpr (LAYOUT == LAYOUT_AWQ) {      // (which channels a thread converts is a per-lane selector, not a code path)
#ifdef MXQ_AWQ_RAW
        if constexpr (DEQ_R > 1) awq_segment_raw<ABL, DEQ_R>(c, nxt, pre, S, au);
        else awq_segment_raw1<ABL>(c, nxt, pre, S, au, raw_slot);
#else
        deq_segment_h<ABL, LAYOUT, 0, DEQ_R>(c, nxt, rowmeta, N, n0, pre, S G8_AWQU_ARG);
#endif
    } else if constexpr (DEQ_PARTS == 4) {
        if (c.h == 0) deq_segment_h<ABL, LAYOUT, 0, DEQ_R>(c, nxt, rowmeta, N, n0, pre, S G8_AWQU_ARG);
        else if (c.h == 1) deq_segment_h<ABL, LAYOUT, 1, DEQ_R>(c, nxt, rowmeta, N, n0, pre, S G8_AWQU_ARG);
        else if (c.h == 2) deq_segment_h<ABL, LAYOUT, 2, DEQ_R>(c, nxt, rowmeta, N, n0, pre, S G8_AWQU_ARG);
        else deq_segment_h<ABL, LAYOUT, DEQ_PARTS - 1, DEQ_R>(c, nxt, rowmeta, N, n0, pre, S G8_AWQU_ARG);
    } else {
        if (c.h == 0) deq_segment_h<ABL, LAYOUT, 0, DEQ_R>(c, nxt, rowmeta, N, n0, pre, S G8_AWQU_ARG);
        else deq_segment_h<ABL, LAYOUT, 1, DEQ_R>(c, nxt, rowmeta, N, n0, pre, S G8_AWQU_ARG);
    }
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

// grid = max(dp_grid, 8 * units) persistent workgroups: the first dp_grid deal the first dp_tiles tiles round-robin
// (tile = block + k * dp_grid: blocks b and b + 8 share an XCD and dp_grid is a multiple of 8 or the tile count itself, so a
// workgroup's tiles keep its XCD's label) and overlap one tile's output with the next one's first DMAs; the first
// 8 * units then run, as stream-K units, their shares of the `tail` tiles beyond them.
template <int ABL, int LAYOUT>
__global__ __launch_bounds__(THREADS) void G8_KERNEL(const uint16_t* __restrict__ x,
                                                               const uint32_t* __restrict__ qweight,
                                                               const float4* __restrict__ rowmeta,
                                                               uint16_t* __restrict__ y, int M, int N, int K,
                                                               int tiles_m, int tiles_n, int dp_tiles, int dp_grid,
                                                               int tail, int units, float* __restrict__ ws,
                                                               int* __restrict__ cnt G8_AWQ_PARAM) {
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
        // lose a 4-cycle issue slot each time, the chain would lose up to 16).  Not in the 128-token build, whose eight
        // dequant waves have slack and whose one MFMA wave per SIMD has none: 44.5 -> 42.8 us at 1024 x 4096^2 without it
        if constexpr (!(ABL & (EXP_NO_PRIO | EXP_MMA_PRIO)) && BM == 256) __builtin_amdgcn_s_setprio(3);
    }

    constexpr int DIST_MIN = (ABL & EXP_SK_DIST3) ? 3 : SK_DIST_MIN;
    if (wave < N_MMA) sk_stamp<ABL>(cnt, wave, (int)(threadIdx.x & 63), 0);
    sk.owner = 0;
    sk.dist = 0;
    sk.pend_j = -1;
    sk.pend_n = 0;
    // ---- slices mode (tail = -S; launcher: launch8_slices): workgroup b runs slice b % S of tile b / S -- K-steps
    // [NT s / S, NT (s + 1) / S) -- and parks its fp32 partial tile in slab b of the workspace; a combine launch sums the
    // slabs in slice order.  No counters, no waiting: for launches so small that every tile is cut 4-8 ways, where the
    // stream-K fix-up's chain (park -> count -> poll -> read) costs more than a second launch does.
    const bool slices = tail < 0;
    const bool has_dp = !slices && bid < dp_grid;
    const bool has_sk = slices || (tail > 0 && bid < 8 * units);
    // ---- stream-K unit u of XCD e (= this workgroup): K-steps [b0, b1) of that XCD's tail tiles laid end to end
    int b0 = 0, b1 = 0, base = 0;
    sk.park = nullptr;
    if (slices) {
        const int SL = -tail, sl = bid % SL;
        base = bid / SL;
        b0 = sk_bound(sl, NT, SL);
        b1 = sk_bound(sl + 1, NT, SL);
        sk.park = ws + (int64_t)bid * (BM * BN);
    } else if (has_sk) {
        sk.e = bid & 7;
        sk.u = bid >> 3;
        base = dp_tiles + sk.e;
        sk.S = ((tail + 7 - sk.e) >> 3) * NT;   // tail tile t belongs to XCD t & 7: the first tail % 8 XCDs hold one more
        b0 = sk_bound(sk.u, sk.S, units);
        b1 = sk_bound(sk.u + 1, sk.S, units);
    }
    // a unit's pieces in DESCENDING K order: the piece that starts a tile first (parked early), whole tiles, the piece that
    // ends a tile last (its owner finishes it).  Every wave walks the same list, so the barrier counts of the roles match.
    const int j_hi = b1 > b0 ? (b1 - 1) / NT : -1, j_lo = b1 > b0 ? b0 / NT : 0;

    // piece j of this unit (descending order): K-steps [pos, end) of tail tile base + 8 j
    auto piece = [&](int j, int& pos, int& end) {
        pos = b0 > j * NT ? b0 : j * NT;
        end = b1 < (j + 1) * NT ? b1 : (j + 1) * NT;
    };

    if (wave < N_MMA) {
        int ln;
        MXQ_LANE_ID(ln);
        XDma cur, nxt;
        bool pre = false;
        // the prologue DMAs of piece j, issued behind the previous segment's last barrier (they fly under its epilogue)
        auto issue_piece = [&](int j) {
            int pos, end, tm, tn;
            piece(j, pos, end);
            tile_of_block(base + j * 8, tiles_m, tiles_n, tm, tn);
            xdma_setup(nxt, x, M, K, tm * BM, SK_KOFF(pos - j * NT), wave, ln);
            mma_prologue_issue<ABL>(nxt, smem, wave, end - pos);
        };
        if (has_dp) {
            // ---- persistent data-parallel part: whole tiles bid, bid + dp_grid, ...
            int tm, tn;
            tile_of_block(bid, tiles_m, tiles_n, tm, tn);
            xdma_setup(cur, x, M, K, tm * BM, 0, wave, ln);
            mma_prologue_issue<ABL>(cur, smem, wave, NT);
            for (int tile = bid; tile < dp_tiles; tile += dp_grid) {
                MXQ_LANE_ID(ln);   // recomputed per tile and opaque: nothing lane-derived is hoisted (and spilled) across the loop
                const int m0 = tm * BM, n0 = tn * BN;
                const bool more = tile + dp_grid < dp_tiles;
                if (more) tile_of_block(tile + dp_grid, tiles_m, tiles_n, tm, tn);
                mma_segment<ABL, LAYOUT == LAYOUT_DENSE16>(smem, wave, ln, NT, cur, true, y, M, N, m0, n0, NT, sk, [&] {
                    if (more) {
                        xdma_setup(nxt, x, M, K, tm * BM, 0, wave, ln);
                        mma_prologue_issue<ABL>(nxt, smem, wave, NT);
                    } else if (j_hi >= j_lo) {
                        issue_piece(j_hi);         // the unit's first piece follows the last whole tile
                    }
                });
                cur = nxt;
            }
            pre = j_hi >= j_lo;
        }
        for (int j = j_hi; j >= j_lo; --j) {
            int pos, end, tm, tn;
            piece(j, pos, end);
            sk.j = j;
            sk.first = pos == b0;
            sk.owner = !slices && end - pos != NT && end == (j + 1) * NT;
            int uf_;
            sk.dist = !slices && end - pos != NT && sk_contributors(sk, j, NT, uf_) >= DIST_MIN;
            tile_of_block(base + j * 8, tiles_m, tiles_n, tm, tn);
            MXQ_LANE_ID(ln);
            if (!pre) xdma_setup(cur, x, M, K, tm * BM, SK_KOFF(pos - j * NT), wave, ln);
            mma_segment<ABL, LAYOUT == LAYOUT_DENSE16>(smem, wave, ln, end - pos, cur, pre, y, M, N, tm * BM, tn * BN, NT, sk, [&] {
                if (j > j_lo) issue_piece(j - 1);
            });
            cur = nxt;
            pre = true;
        }
        if (sk.pend_j >= 0) {   // the unit ended on a parked piece: its stores must have retired before the count moves
            MXQ_LANE_ID(ln);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sk_bump_pending<ABL>(sk, wave, ln);
            sk_stamp<ABL>(cnt, wave, ln, 3);
        }
        // tiles with many contributors are reduced by all of them, after every piece of this unit is parked and counted
        // (ascending: the tile this unit ENDS first -- its other contributors parked long ago)
        for (int j = j_lo; j <= j_hi && !slices; ++j) {
            int pos, end, uf_, tm, tn;
            piece(j, pos, end);
            if (end - pos == NT) continue;
            const int C = sk_contributors(sk, j, NT, uf_);
            if (C < DIST_MIN) continue;
            tile_of_block(base + j * 8, tiles_m, tiles_n, tm, tn);
            MXQ_LANE_ID(ln);
            sk_reduce_distributed<ABL>(sk, j, C, uf_, NT, wave, ln, smem, y, M, N, tm * BM, tn * BN);
        }
        if constexpr (MXQ_SKSTAMPS(ABL)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MXQ_LANE_ID(ln);
            sk_stamp<ABL>(cnt, wave, ln, 6);
        }
    } else if constexpr (LAYOUT == LAYOUT_DENSE16) {
        if (has_dp) {
            int tm, tn;
            tile_of_block(bid, tiles_m, tiles_n, tm, tn);
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
        }
        for (int j = j_hi; j >= j_lo; --j) {
            const int pos = b0 > j * NT ? b0 : j * NT, end = b1 < (j + 1) * NT ? b1 : (j + 1) * NT;
            int tm, tn;
            tile_of_block(base + j * 8, tiles_m, tiles_n, tm, tn);
            int ln;
            MXQ_LANE_ID(ln);
            WDma w;
            wdma_setup(w, (const uint16_t*)qweight, N, K, tn * BN, pos - j * NT, end - pos, wave, ln);
            wdma_segment(w, smem, false, nothing);
        }
    } else {
        typename PkOf<LAYOUT>::type S[DEQ_R] = {};
        int ln;
        MXQ_LANE_ID(ln);
        Deq cur, nxt;
#ifdef MXQ_G8_AWQ
        const AwqU au = {make_rsrc(qweight, (uint32_t)K * (uint32_t)(N >> 1)), make_rsrc(awq.scales, (uint32_t)awq.groups * (uint32_t)N * 2u),
                         make_rsrc(awq.zeros, (uint32_t)awq.groups * (uint32_t)(N >> 1)), awq.gmul, (uint32_t)N >> 1, K / BK - 1};
        int raw_slot = 0;
#endif
        bool pre = false;                  // a segment's last burst loads the next segment's first group
        auto piece_deq = [&](int j, Deq& d, int& n0) {
            int pos, end, tm, tn;
            piece(j, pos, end);
            tile_of_block(base + j * 8, tiles_m, tiles_n, tm, tn);
            n0 = tn * BN;
            deq_setup<LAYOUT>(d, smem, wave, ln, qweight, N, K, n0, SK_KOFF(pos - j * NT), end - pos);
        };
        int n0_sk = 0;
        if (has_dp) {
            int tm, tn;
            tile_of_block(bid, tiles_m, tiles_n, tm, tn);
            deq_setup<LAYOUT>(cur, smem, wave, ln, qweight, N, K, tn * BN, 0, NT);
            for (int tile = bid; tile < dp_tiles; tile += dp_grid) {
                MXQ_LANE_ID(ln);
                const int n0 = tn * BN;
                const bool more = tile + dp_grid < dp_tiles;
                if (more) {
                    tile_of_block(tile + dp_grid, tiles_m, tiles_n, tm, tn);
                    deq_setup<LAYOUT>(nxt, smem, wave, ln, qweight, N, K, tn * BN, 0, NT);
                } else if (j_hi >= j_lo) {
                    piece_deq(j_hi, nxt, n0_sk);   // the unit's first piece follows the last whole tile
                } else {
                    deq_none(nxt, cur);
                }
                deq_segment<ABL, LAYOUT>(cur, nxt, wave, ln, rowmeta, N, n0, pre, S G8_AWQU_ARG);
                pre = true;
                cur = nxt;
            }
        }
        for (int j = j_hi; j >= j_lo; --j) {
            MXQ_LANE_ID(ln);
            if (!pre) piece_deq(j, cur, n0_sk);
            const int n0 = n0_sk;
            if (j > j_lo) piece_deq(j - 1, nxt, n0_sk);
            else deq_none(nxt, cur);
            deq_segment<ABL, LAYOUT>(cur, nxt, wave, ln, rowmeta, N, n0, pre, S G8_AWQU_ARG);
            pre = true;
            cur = nxt;
        }
    }
}

// slices mode, second launch: y = the S slabs of every tile summed in slice order (from +0: fixed).  One wave per (tile,
// producing wave, token block): the slabs' four fragments of up to four slices in flight together, then the fp16 block.
__global__ __launch_bounds__(256) void G8_SYM(, _combine_kernel)(const float* __restrict__ slab, uint16_t* __restrict__ y, int M,
                                                                int N, int tiles_m, int tiles_n, int S) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= tiles_m * tiles_n * N_MMA * NJ) return;
    const int tile = task / (N_MMA * NJ), rem = task % (N_MMA * NJ);
    const int ws = rem / NJ, jj = rem % NJ;
    int tm, tn;
    tile_of_block(tile, tiles_m, tiles_n, tm, tn);
    const float* src = slab + (int64_t)tile * S * (BM * BN) + ws * (WSLOT / 4) + lane * 4;
    f32x4 c4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // U slices' loads go out together (clamped, never branched around), then they are added in slice order: 8 slices -- the
    // 32-tile launches -- are ONE round trip of 32 loads per lane, not two of 16
    auto sum_slices = [&](auto Uc) __attribute__((always_inline)) {
        constexpr int U = decltype(Uc)::value;
        for (int s0 = 0; s0 < S; s0 += U) {
            f32x4 v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sc = s0 + u < S ? s0 + u : S - 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[u][i] = *(const f32x4*)(src + (int64_t)sc * (BM * BN) + (i * NJ + jj) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (s0 + u < S) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) c4[i] = c4[i] + v[u][i];
                }
        }
    };
    if (S <= 4) sum_slices(std::integral_constant<int, 4>{});
    else sum_slices(std::integral_constant<int, 8>{});
    store_block_xpose(c4, y, M, N, tm * BM, tn * BN, ws / WGN, ws % WGN, jj, lane & 15, lane >> 4);
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

constexpr size_t CNT_BYTES = 64 * 1024;   // K-step counters at the head of the workspace (>= 8*units*N_MMA ints)

template <int ABL, int LAYOUT = MXQ_LAYOUT_MIXED>
static int launch8(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   void* workspace, size_t ws_bytes, bool force, hipStream_t stream G8_AWQ_PARAM) {
    // the DMA descriptors address one tile's rows with 32-bit offsets: 256 rows of x, 8 row-blocks of packed weights
    // 32-bit offsets: 256 rows of x per DMA descriptor; the whole packed weight behind one (offsets < 2^31)
    if ((int64_t)BM * K * 2 >= ((int64_t)1 << 32) || (int64_t)(N / 16) * (K / BK) * MXQ_BLK_BYTES >= ((int64_t)1 << 31))
        return -1;   // MXQ_E_SHAPE
    hipError_t e = mxq_set_dyn_lds_once<&G8_KERNEL<ABL, LAYOUT>>(SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    const int NT = K / BK;
    const int cus = cu_count() / 8 * 8;
    int units = cus / 8;
    int dp_tiles = tiles, tail = 0;
    // a stream-K launch's workgroups wait for one another: its grid (<= cus) must fit the device all at once -- asked of the
    // occupancy calculator for THIS kernel's registers and LDS, once per device; if it does not (or the query fails) the
    // launch keeps whole tiles, which never wait
    const bool coresident = mxq_resident_workgroups<&G8_KERNEL<ABL, LAYOUT>>(THREADS, SMEM_BYTES) >= cus;
    if (workspace && coresident && tiles % cus != 0 && units * 8 * N_MMA <= SK_DONE_OFF && (SK_DONE_OFF + units * 8) * sizeof(int) <= CNT_BYTES &&
        ws_bytes >= CNT_BYTES + (size_t)cus * 2 * BM * BN * sizeof(float)) {
        const int t8 = (tiles % cus) / 8;   // tail tiles per XCD (the first tail % 8 XCDs hold one more)
        // Splitting the tail saves the idle share of one tile time, (1 - tail/CUs) * NT K-steps of ~1 us, and costs the
        // pieces' extra prologues, the parked partials and -- most of it -- K-steps that run ~1.3 x slower than whole
        // tiles' (units of one tile walk different K ranges: nothing they read is shared through L2).  Round 4's
        // protocol (pieces in descending K order, the owner reduces in registers: header): worth it from ~20 idle K-steps
        // per CU -- Llama's gate/up at 2048 tokens (tail 176 / 256, NT = 64): 183.6 -> 180.5 us; at 1024 tokens 109 -> 101
        // (profiles/r04_streamk.txt).  Round 2-3's (every piece parked, last arriver reduces) needed 24.
        const bool pays = (int64_t)(cus - tiles % cus) * NT >= (int64_t)20 * cus;
        if ((force || pays) && (int64_t)t8 * NT >= (int64_t)units * 4) {
            tail = tiles % cus;
            dp_tiles = tiles - tail;
        } else if (pays && tiles > cus && (int64_t)t8 * NT < (int64_t)units * 4) {
            // A tail of a few tiles (258 tiles on 256 CUs: Llama's gate/up at 640-768 tokens; 516 at 1536) used to run
            // as a whole extra round of 2-4 workgroups.  Split it over FEWER units per XCD instead: >= 8 K-steps per
            // unit and at most 8 contributors per tile (the finisher reads them all back).
            const int tmax = (tiles % cus + 7) / 8;
            int u = tmax * NT / 8;
            if (u > 8 * tmax) u = 8 * tmax;
            if (u >= 2) {
                units = u < units ? u : units;
                tail = tiles % cus;
                dp_tiles = tiles - tail;
            }
        }
    }
    const int dp_grid = dp_tiles < cus ? dp_tiles : cus;   // persistent: at most one data-parallel workgroup per CU
    // workgroup 8u + e runs its whole tiles and then, as unit u of XCD e, its share of the tail: never more than one
    // workgroup per CU, all of them resident (an owner may wait for lower-numbered units: header)
    const int grid = tail && 8 * units > dp_grid ? 8 * units : dp_grid;
    G8_KERNEL<ABL, LAYOUT><<<grid, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n,
        dp_tiles, dp_grid, tail, units, (float*)((char*)workspace + CNT_BYTES), (int*)workspace G8_AWQ_ARG);
    return (int)hipGetLastError();
}

// slices mode: every tile's K range cut into S equal slices (S <= 0: one workgroup per CU), partial tiles through the
// workspace beyond its counter head (untouched), a combine launch.  S = 1 (or no room): the ordinary whole-tile launch.
template <int LAYOUT>
static int launch8_slices(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                          void* workspace, size_t ws_bytes, int S, hipStream_t stream G8_AWQ_PARAM) {
    const int NT = K / BK, tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    if (S <= 0) S = cu_count() / tiles;               // (2-3 x as many slices, i.e. 2-3 workgroups per CU: no faster, profiles/r05_smalltile_slices.txt)
    if (S > NT / 4) S = NT / 4;                       // at least 4 K-steps per slice
    const size_t room = workspace && ws_bytes > CNT_BYTES ? (ws_bytes - CNT_BYTES) / ((size_t)tiles * BM * BN * sizeof(float)) : 0;
    if ((size_t)S > room) S = (int)room;
    if (S <= 1) return launch8<0, LAYOUT>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream G8_AWQ_ARG);
    if (M <= 0 || N <= 0 || K < BK || K % BK != 0 || N % (LAYOUT == LAYOUT_AWQ ? 8 : 16) != 0) return -1;
    if ((int64_t)BM * K * 2 >= ((int64_t)1 << 32) || (int64_t)(N / 16) * (K / BK) * MXQ_BLK_BYTES >= ((int64_t)1 << 31))
        return -1;
    hipError_t e = mxq_set_dyn_lds_once<&G8_KERNEL<0, LAYOUT>>(SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    float* slab = (float*)((char*)workspace + CNT_BYTES);
    G8_KERNEL<0, LAYOUT><<<tiles * S, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n, 0, 0,
        -S, 0, slab, nullptr G8_AWQ_ARG);
    G8_SYM(, _combine_kernel)<<<(tiles * N_MMA * NJ + 3) / 4, 256, 0, stream>>>(slab, (uint16_t*)y, M, N, tiles_m, tiles_n, S);
    return (int)hipGetLastError();
}

}   // namespace

#ifdef MXQ_G8_AWQ
// gemm_forward_cuda's operands (gemm_cuda.h:3-4): x fp16 [M, IC], kernel int32 [IC, OC / 8], scales fp16 [IC / G, OC], zeros int32
// [IC / G, OC / 8] -> y fp16 [M, OC].  slices = 0: whole tiles + stream-K tail (workspace: counter head + partial slots, nullable);
// slices > 0 (or -1: one workgroup per CU): every tile's K range cut that many ways + combine launch; -2: stream-K with the
// tail always split (launches of few tiles).
size_t G8_SYM(, _workspace_bytes)() { return CNT_BYTES + (size_t)(cu_count() / 8 * 8) * 2 * BM * BN * sizeof(float); }
int G8_SYM(launch_, _f16)(const void* x, const void* kernel, const void* scales, const void* zeros, void* y, int M, int IC, int OC,
                         int G, void* workspace, size_t ws_bytes, int slices, hipStream_t stream) {
    if (M <= 0 || IC < BK || IC % BK != 0 || OC <= 0 || OC % 8 != 0 || G < 8 || G % 8 != 0 || IC % G != 0 || G >= 4096 || IC >= (1 << 20))
        return -1;   // MXQ_E_SHAPE
    if ((int64_t)IC * OC / 2 >= ((int64_t)1 << 31) || (int64_t)(IC / G) * OC * 2 >= ((int64_t)1 << 31)) return -1;
    AwqOps awq;
    awq.scales = (const uint16_t*)scales;
    awq.zeros = (const uint32_t*)zeros;
    awq.gmul = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)G) + 1u;
    awq.groups = IC / G;
    if (slices == -2)    // stream-K, the tail always split
        return launch8<0, LAYOUT_AWQ>(x, kernel, nullptr, y, M, OC, IC, workspace, ws_bytes, true, stream, awq);
    if (slices != 0)
        return launch8_slices<LAYOUT_AWQ>(x, kernel, nullptr, y, M, OC, IC, workspace, ws_bytes, slices, stream, awq);
    return launch8<0, LAYOUT_AWQ>(x, kernel, nullptr, y, M, OC, IC, workspace, ws_bytes, false, stream, awq);
}
#else
int G8_SYM(launch_, _slices_f16)(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                 int layout, void* workspace, size_t ws_bytes, int S, hipStream_t stream) {
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launch8_slices<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, S, stream);
        case MXQ_LAYOUT_W2G16: return launch8_slices<MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, S, stream);
        case MXQ_LAYOUT_W4ROW: return launch8_slices<MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, S, stream);
        case MXQ_LAYOUT_MIXEDC: return launch8_slices<MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, S, stream);
    }
    return -1;
}

size_t G8_SYM(, _workspace_bytes)() { return CNT_BYTES + (size_t)(cu_count() / 8 * 8) * 2 * BM * BN * sizeof(float); }

int G8_SYM(launch_, _f16)(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                         void* workspace, size_t ws_bytes, int force, hipStream_t stream) {
    return launch8<0>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, force != 0, stream);
}

int G8_SYM(launch_, _layout_f16)(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                int layout, void* workspace, size_t ws_bytes, hipStream_t stream) {
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launch8<0, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, false, stream);
        case MXQ_LAYOUT_W2G16: return launch8<0, MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, false, stream);
        case MXQ_LAYOUT_W4ROW: return launch8<0, MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, false, stream);
        case MXQ_LAYOUT_MIXEDC: return launch8<0, MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, false, stream);
    }
    return -1;
}

#if MXQ_G8_BM == 256   // (the dense mode's four DMA waves belong to the 256-token build)
// hoisted-dequant mode: w16 = dense fp16 [N, K] weight (the dequant kernel's output); same tiles, no stream-K tail
int G8_SYM(launch_, _dense_f16)(const void* x, const void* w16, void* y, int M, int N, int K, hipStream_t stream) {
    return launch8<0, LAYOUT_DENSE16>(x, w16, nullptr, y, M, N, K, nullptr, 0, false, stream);
}
#endif

#endif   // !MXQ_G8_AWQ

#if defined(MXQ_PROFILING) && !defined(MXQ_G8_AWQ)
// Built only into libmxq_hip_prof.so (make prof; tools/): parts of the kernel removed to time the rest.
// WRONG RESULTS by construction -- never part of libmxq_hip.so or of include/mxq_hip.h.
// 1 = no x DMAs, 2 = no MFMA, 4 = no dequant, 256 = no output stores (sums)
extern "C" int G8_SYM(prof_, _ablate_f16)(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N,
                                         int K, int abl, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    switch (abl) {
        case 0: return launch8<0>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 1: return launch8<1>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 2: return launch8<2>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 4: return launch8<4>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 5: return launch8<5>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 6: return launch8<6>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 256: return launch8<256>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 260: return launch8<260>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 1024: return launch8<1024>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);   // correct results
        case 2048: return launch8<2048>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);   // correct results
        case 2050: return launch8<2050>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 16: return launch8<16>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 32: return launch8<32>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 48: return launch8<48>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 512: return launch8<512>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 8192: return launch8<8192>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 8704: return launch8<8704>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        // 64 / 128 / 192: one / two / three in-stream filler VALU ops behind every MFMA (+ 4: without the dequant)
        case 64: return launch8<64>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 68: return launch8<68>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 132: return launch8<132>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 196: return launch8<196>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
    }
    return -1;   // MXQ_E_SHAPE: not an ablation this build carries
}

#if MXQ_G8_BM == 256
// An UNRELATED kernel that holds `lds_bytes` of LDS per workgroup for `usec` microseconds (100-MHz s_memrealtime; every wave
// leaves when the time is up): tests/test_gpu_parity.py runs it on a second stream beside a stream-K launch, whose workgroups
// (144 KB of LDS each) then cannot all be resident at once -- the launch must still come out right, status 0.
__global__ void mxq_prof_occupy_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ int occ_lds[];
    occ_lds[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int acc = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        acc += occ_lds[(threadIdx.x + acc) & 63];
        __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0x7fffffff) *sink = acc;
}
extern "C" int mxq_prof_occupy(int grid, int lds_bytes, unsigned long long usec, void* sink, void* stream_) {
    if (grid <= 0 || lds_bytes < 256 || lds_bytes > 160 * 1024 || usec > 50000) return -1;       // bounded: never a hang
    hipError_t e = hipFuncSetAttribute((const void*)mxq_prof_occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    mxq_prof_occupy_kernel<<<grid, 64, lds_bytes, (hipStream_t)stream_>>>(usec * 100ull, (int*)sink);
    return (int)hipGetLastError();
}
// the hoisted mode's MFMA kernel alone on an already dequantised fp16 weight (the dense yardstick of tools/ab_gemm.py)
extern "C" int G8_SYM(prof_, _dense_f16)(const void* x, const void* w16, void* y, int M, int N, int K, void* stream_) {
    return launch8<0, LAYOUT_DENSE16>(x, w16, nullptr, y, M, N, K, nullptr, 0, false, (hipStream_t)stream_);
}
#endif

// Diagnostic build with cycle stamps (cdna guide section 7, "In-kernel stamps"): dbg receives, per workgroup and wave,
// {work, wait, barrier, steps} cycle sums over the steady-state K-steps (u64 x 4 x 12 waves x grid).  Never timed.
template <int ABL>
static int launch8_stamps(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                          void* dbg, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)G8_KERNEL<ABL, MXQ_LAYOUT_MIXED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    G8_KERNEL<ABL, MXQ_LAYOUT_MIXED><<<tiles_m * tiles_n, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n,
        tiles_m * tiles_n, tiles_m * tiles_n, 0, 32, (float*)dbg, nullptr);
    return (int)hipGetLastError();
}
extern "C" int G8_SYM(prof_, _stamps_f16)(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N,
                                         int K, int abl, void* dbg, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    switch (abl) {
        case 0: return launch8_stamps<4096>(x, qweight, rowmeta, y, M, N, K, dbg, stream);
        case 2048: return launch8_stamps<4096 + 2048>(x, qweight, rowmeta, y, M, N, K, dbg, stream);
        case 1024: return launch8_stamps<4096 + 1024>(x, qweight, rowmeta, y, M, N, K, dbg, stream);
        case 4: return launch8_stamps<4096 + 4>(x, qweight, rowmeta, y, M, N, K, dbg, stream);
        case 2: return launch8_stamps<4096 + 2>(x, qweight, rowmeta, y, M, N, K, dbg, stream);
    }
    return -1;
}
// fault injection for the expiry test (EXP_SK_WITHHOLD above): the tail always split, unit 0's counts withheld
extern "C" int G8_SYM(prof_, _skwithhold_f16)(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N,
                                             int K, void* workspace, size_t ws_bytes, void* stream_) {
    return launch8<EXP_SK_WITHHOLD>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, true, (hipStream_t)stream_);
}
// stream-K fix-up phases (tools/sk_stamps.py): the product dispatch (force = 0) or the tail always split, with the owner
// protocol as shipped (dist3 = 0) or the all-contributors reduction from 3 contributors on; correct results
extern "C" int G8_SYM(prof_, _skstamps_f16)(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N,
                                           int K, int dist3, void* workspace, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (dist3) return launch8<EXP_SKSTAMPS | EXP_SK_DIST3>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, true, stream);
    return launch8<EXP_SKSTAMPS>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, true, stream);
}
#endif   // MXQ_PROFILING
