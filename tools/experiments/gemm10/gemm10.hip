// W2/4 x A16 dequant-GEMM for prefill, generation 10: EIGHT waves, no dedicated dequant waves -- every wave converts its
// share of the weight tile BETWEEN ITS OWN MFMAs.
//
// Counterpart of the reference's (never built) AWQ tensor-core GEMM
// mxq_quant/cuda_kernel/csrc/quantization/gemm_cuda_gen.cu:28-218 (dequant into shared memory + mma, :121-205) and of the
// implicit nn.Linear on the fake-quant weight (mxq_quant/main.py:85); arithmetic contract x16 . fp16(scale * (q - zero))^T
// of lib/quantizer.py:19-20 + mxqgpt.py:448, fp32 accumulation.  Same tile, same products, same summation order as
// gemm8.hip: the two kernels agree bit for bit.
//
// Why (profiles/r04_inwave_filler_probe.txt): gemm8 gives the dequant to one extra wave per SIMD, and a third wave's
// vector-ALU work is ADDITIVE to the SIMD's matrix time (r03 probe): ~90 ops per SIMD and K-step cost 11.3 us of 65.9 at
// 2048 x 4096^2.  The same ops issued by the MFMA waves themselves, one or two per MFMA gap, cost 4-6 us: an MFMA holds
// the SIMD's vector issue for 8 of its 16 cycles and ops that follow it in program order fit into the other 8.
//
// Tile 256 tokens x 128 channels x K-step 64 (= one MXQ chunk); 8 waves = 4 (tokens) x 2 (channels), a wave owns a
// 64 x 64 sub-tile (4 x 4 v_mfma_f32_16x16x32_f16, D^T = W . x^T: a lane owns 4 consecutive channels of a token).  Per
// K-step every wave
//   * reads its fragments and issues 32 MFMAs (as gemm8's MFMA waves);
//   * issues 4 x 1 KiB LDS-DMA pieces of the x tile two steps ahead (3-slot ring) and ONE piece of the packed weight:
//     row block `wave` of the tile, raw (576 / 480 / 512 B = 36 / 30 / 32 lanes x 16 B), four steps ahead into a
//     4-slot ring, issued BEHIND the step's x pieces so that the in-order vmcnt never waits for HBM on x's behalf;
//   * converts one (row, 16-column quarter) unit of the NEXT chunk per lane -- wave w: quarter w >> 1, rows
//     64 (w & 1) + lane; mixed layouts: quarters 0-2 are 2-bit groups (4-entry fp16 LUT + v_perm_b32, ~40 ops), quarter
//     3 the 4-bit arm (v_cvt_f32_ubyte, sub, mul, cvt: ~62 ops); wave-uniform roles, never divergent -- in four pieces
//     pinned between the MFMAs by sched_group_barrier, and writes its 32 bytes into the fp16 W16 double buffer;
//   * meets ONE raw s_barrier.  All vector-memory traffic of the loop is LDS-DMA: one counted vmcnt(6) per step.
// LDS: x ring 96 KB + W16 32 KB + packed ring 32 KB = 160 KB.  8 waves = 2 per SIMD: no 168-VGPR cap.
//
// Persistent workgroups and the hybrid stream-K tail are gemm8's (same protocol, same workspace): grid = min(tiles, CUs)
// workgroups dealing whole tiles round-robin + 8 * units stream-K workgroups for the tiles beyond the last full round.
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int NW = 8, THREADS = NW * 64;
constexpr int A_STAGE = BM * BK * 2, A_SLOTS = 3;
constexpr int W_STAGE = BN * BK * 2;
// packed ring: 8 row blocks of one chunk per slot, 1 KB apart (a DMA instruction writes 64 x 16 B: the lanes beyond
// the block's 36 / 30 / 32 fetch nothing and deposit zeros behind it -- no branch around the instruction)
constexpr int P_SLOTS = 4, P_WAVE = 1024, P_SLOT = 8 * P_WAVE;
constexpr int OFF_A = 0;
constexpr int OFF_W = OFF_A + A_SLOTS * A_STAGE;
constexpr int OFF_P = OFF_W + 2 * W_STAGE;
constexpr int SMEM_BYTES = OFF_P + P_SLOTS * P_SLOT;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");

// profiling-only switches (libmxq_hip_prof.so): 4 = no conversion arithmetic (timing only, WRONG results)
[[maybe_unused]] constexpr int ABL_NO_CONV = 4;
// 8: the conversion's ops replaced by as many INDEPENDENT v_add_f32 (2 / 2.2 per micro-step) and its result by cheap
// pseudo-weights made of the code word's bits (same magnitude as real weights: the clock the chip holds depends on the data);
// 16: no packed-weight DMA inside the loop; 32: no LDS reads of the packed words, no W16 writes by the conversion
[[maybe_unused]] constexpr int ABL_FILL = 8, ABL_NO_PDMA = 16, ABL_NO_CVLDS = 32;

template <int LAYOUT>
struct Lay {
    static constexpr bool MIXED = LAYOUT == MXQ_LAYOUT_MIXED || LAYOUT == MXQ_LAYOUT_MIXEDC;
    static constexpr bool COMPACT = LAYOUT == MXQ_LAYOUT_MIXEDC;
    static constexpr int BLK_B = LAYOUT == MXQ_LAYOUT_W4ROW ? 512 : COMPACT ? MXQC_BLK_BYTES : MXQ_BLK_BYTES;
    static constexpr int QQ0 = COMPACT ? MXQC_OFF_QQ : MXQ_OFF_QQ;
    // does quarter q of a chunk hold 4-bit codes?
    static __device__ __forceinline__ bool is4(int q) { return LAYOUT == MXQ_LAYOUT_W4ROW || (MIXED && q == 3); }
};

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ void bufdma16(rsrc_t rsrc, uint32_t voff, uint32_t soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

// XCD-aware tile order (speed only; gemm8.hip): tiles are dealt to the 8 XCDs as compact 2-D blocks
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
// stream-K bookkeeping (protocol: gemm8.hip header; slots and counters have gemm8's layout, so one workspace serves both)
// ------------------------------------------------------------------------------------------------
struct SkSeg {
    float* ws;        // partial slots: [unit = 8u+e][2][BM*BN] fp32
    int* cnt;         // K-step counters: [tail tile = 8j+e][NW waves]
    int u, e, units;  // this unit, its XCD, units per XCD
    int S;            // K-steps in one XCD's tail
    int j;            // tile index inside the XCD's tail
    int first;        // 1: the segment starts at the unit's range start (slot 0), else slot 1
};
__device__ __forceinline__ void st_agent(float* slot, int f, int lane, f32x4 v) {
    const rsrc_t r = make_rsrc(slot, 16384u);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (uint32_t)(f * 64 + lane) * 16u, 0u, 16);
}
__device__ __forceinline__ f32x4 ld_agent(const float* slot, int f, int lane) {
    const rsrc_t r = make_rsrc(slot, 16384u);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)(f * 64 + lane) * 16u, 0u, 16));
}
__device__ __forceinline__ int sk_bound(int u, int S, int units) { return (int)((uint32_t)u * (uint32_t)S / (uint32_t)units); }

// ------------------------------------------------------------------------------------------------
// operands
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

template <int I0, int I1>
__device__ __forceinline__ void mfma_rows(f32x4 (&acc)[4][4], const Frag4& wf, const Frag4& xf) {
#pragma unroll
    for (int i = I0; i < I1; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
}

// the x tile of K-step t: wave w fills rows 32w .. 32w+31 of slot t % 3 with 4 DMAs of 8 full 128-B rows (gemm8.hip)
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
__device__ __forceinline__ void xdma_setup(XDma& xd, const uint16_t* __restrict__ x, int M, int K, int m0, int kt0,
                                           int wave, int lane) {
    const int rows = M - m0 < BM ? M - m0 : BM;
    xd.rsrc = make_rsrc(x + (int64_t)m0 * K, (uint32_t)rows * (uint32_t)K * 2u);
    xd.k0 = (uint32_t)kt0 * (BK * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        xd.voff[i] = (uint32_t)row * (uint32_t)K * 2u + ((((uint32_t)lane & 7u) ^ ((uint32_t)row & 7u)) << 4);
    }
}

// the packed blocks of K-step t: wave w copies row block w of the tile, raw, into slot t % 3 (BLK_B / 16 lanes x 16 B).
// One descriptor for the WHOLE packed weight: a row block beyond the weight's last one lies beyond the buffer and
// arrives as zeros (scale 0 -> weights 0); a chunk beyond the segment's end delivers another chunk's bytes, which
// nobody converts.
struct PDma {
    rsrc_t rsrc;
    uint32_t voff;       // byte offset of (row block, lane's 16 bytes); 0x80000000 = nothing to load
    uint32_t k0;         // byte offset of the segment's first chunk inside a row block's run
};
template <int LAYOUT>
__device__ __forceinline__ void pdma_setup(PDma& p, const uint32_t* __restrict__ qweight, int N, int K, int n0, int kt0,
                                           int wave, int lane) {
    constexpr int BLK_B = Lay<LAYOUT>::BLK_B;
    const uint32_t blk_stride = (uint32_t)(K / BK) * BLK_B;
    p.rsrc = make_rsrc(qweight, (uint32_t)(N >> 4) * blk_stride);
    p.voff = lane < BLK_B / 16 ? (uint32_t)((n0 >> 4) + wave) * blk_stride + (uint32_t)lane * 16u : 0x80000000u;
    p.k0 = (uint32_t)kt0 * BLK_B;
}
template <int LAYOUT>
__device__ __forceinline__ void issue_p(const PDma& p, char* smem, int wave, int t) {
    bufdma16(p.rsrc, p.voff, p.k0 + (uint32_t)t * Lay<LAYOUT>::BLK_B, smem + OFF_P + (t % P_SLOTS) * P_SLOT + wave * P_WAVE);
}

#define MXQ_FENCE() __builtin_amdgcn_sched_barrier(0)

// ------------------------------------------------------------------------------------------------
// conversion: one (row, quarter) unit = 16 weights per lane and K-step, in pieces that sit between the MFMAs
// ------------------------------------------------------------------------------------------------
struct Cv {
    // role (per wave / lane, fixed for the kernel)
    int q, row, r;       // quarter of the chunk (wave-uniform), W-tile row 0..127, row inside its block
    float s4, z4;        // 4-bit role: the row's scale / zero-point (rowmeta)
    // per-step state
    uint32_t c0, c1;     // code words
    uint32_t zb, scw;    // 2-bit role: zero-point bits, the row's scale codes
    float qs, qz, s, z;
    float t0, t1, a0, a1, a2;            // values in flight between micro-steps
    uint32_t m, m1, p01, p23, lut_lo, lut_hi, lo, hi, lo1, hi1;
    uint32_t o[8];
};

// piece 0: the unit's packed words out of the raw ring (slot of chunk tc)
template <int LAYOUT, bool R4>
__device__ __forceinline__ void cv_read(Cv& c, const char* smem, int tc) {
    typedef Lay<LAYOUT> L;
    const uint32_t* blk = (const uint32_t*)(smem + OFF_P + (tc % P_SLOTS) * P_SLOT + (c.row >> 4) * P_WAVE);
    if constexpr (R4) {
        if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
            c.c0 = blk[mxq_w4_c4(c.q, 0, c.r)];
            c.c1 = blk[mxq_w4_c4(c.q, 1, c.r)];
        } else {
            c.c0 = blk[mxq_c4(0, c.r)];
            c.c1 = blk[mxq_c4(1, c.r)];
        }
    } else {
        const int g = c.q;
        if constexpr (LAYOUT == MXQ_LAYOUT_W2G16) {
            c.c0 = blk[mxq_w2_c2(g, c.r)];
            c.zb = blk[mxq_w2_z2(g, c.r)];
        } else {
            c.c0 = blk[mxq_c2(g, c.r)];
            if constexpr (L::COMPACT) c.zb = ((const uint16_t*)blk)[mxqc_z2_u16(g, c.r)];
            else c.zb = blk[mxq_z2(g, c.r)];
        }
        c.scw = ((const uint16_t*)blk)[L::COMPACT ? mxqc_sc_u16(c.r) : mxq_sc_u16(c.r)];
        c.qs = __uint_as_float(blk[L::QQ0 + g * 2]);
        c.qz = __uint_as_float(blk[L::QQ0 + g * 2 + 1]);
    }
}

// The arithmetic of mxq_deq2x16 / mxq_deq4x8 (mxq_dequant.h: same ops on the same values, bit-identical results) cut into
// MICRO-STEPS of ~2 vector-ALU ops: step K sits right behind the K-th MFMA of sections C..I of a K-step, kept there by
// the scheduling fence that follows it (sched_group_barrier pipelines do not hold for these chains).  The two ops of a
// step are INDEPENDENT of each other and depend on earlier steps only -- an in-order wave stalls on a dependent pair
// (measured: the same op count as independent fillers 67.2 us, as dependent pairs 73.6 us at 2048 x 4096^2).
// 2-bit unit: 20 steps (K = 0..19), 40 ops: scale, 4-entry LUT, then the four 4-element selections two at a time.
template <int LAYOUT, int K>
__device__ __forceinline__ void cv_micro2(Cv& c) {
    constexpr uint32_t M2 = 0x03030303u;
    if constexpr (K == 0) {
        if constexpr (Lay<LAYOUT>::COMPACT) c.z = (float)__builtin_bit_cast(_Float16, (uint16_t)c.zb);
        else c.z = __uint_as_float(c.zb);
        c.m = (c.scw >> (4 * c.q)) & 15u;
    } else if constexpr (K == 1) {
        c.t0 = (float)c.m;
        c.a0 = 0.0f - c.z;
    } else if constexpr (K == 2) {
        c.t0 = c.t0 - c.qz;
        c.a1 = 1.0f - c.z;
    } else if constexpr (K == 3) {
        c.s = c.qs * c.t0;                             // mxq_scale
        c.a2 = 2.0f - c.z;
    } else if constexpr (K == 4) {
        c.a0 = c.s * c.a0;
        c.a1 = c.s * c.a1;
    } else if constexpr (K == 5) {
        c.a2 = c.s * c.a2;
        c.t1 = 3.0f - c.z;
    } else if constexpr (K == 6) {
        c.p01 = mxq_pack_f16(c.a0, c.a1);
        c.t1 = c.s * c.t1;
    } else if constexpr (K == 7) {
        c.m = c.c0 & M2;                               // codes of elements 0..3
        c.m1 = (c.c0 >> 2) & M2;                       // 4..7
        c.p23 = mxq_pack_f16(c.a2, c.t1);
    } else if constexpr (K == 8) {
        c.lut_lo = __builtin_amdgcn_perm(c.p23, c.p01, 0x06040200u);
        c.lut_hi = __builtin_amdgcn_perm(c.p23, c.p01, 0x07050301u);
    } else if constexpr (K == 9 || K == 14) {          // two selections (J, J+1) side by side, five steps
        c.lo = __builtin_amdgcn_perm(0u, c.lut_lo, c.m);
        c.lo1 = __builtin_amdgcn_perm(0u, c.lut_lo, c.m1);
    } else if constexpr (K == 10 || K == 15) {
        c.hi = __builtin_amdgcn_perm(0u, c.lut_hi, c.m);
        c.hi1 = __builtin_amdgcn_perm(0u, c.lut_hi, c.m1);
    } else if constexpr (K == 11 || K == 16) {
        constexpr int J = K == 11 ? 0 : 2;
        c.o[2 * J] = __builtin_amdgcn_perm(c.hi, c.lo, 0x05010400u);
        c.o[2 * J + 2] = __builtin_amdgcn_perm(c.hi1, c.lo1, 0x05010400u);
    } else if constexpr (K == 12 || K == 17) {
        constexpr int J = K == 12 ? 0 : 2;
        c.o[2 * J + 1] = __builtin_amdgcn_perm(c.hi, c.lo, 0x07030602u);
        c.o[2 * J + 3] = __builtin_amdgcn_perm(c.hi1, c.lo1, 0x07030602u);
    } else if constexpr (K == 13) {
        c.m = (c.c0 >> 4) & M2;                        // 8..11
        c.m1 = (c.c0 >> 6) & M2;                       // 12..15
    }
    // K == 18, 19: nothing left
}
// 4-bit unit (two code words): 28 steps, 62 ops; two elements side by side through cvt, sub, mul.
template <int K>
__device__ __forceinline__ void cv_micro4(Cv& c) {
    constexpr uint32_t M4 = 0x0F0F0F0Fu;
    constexpr int W = K / 14, k = K % 14, OB = W * 4;
    const uint32_t d = W ? c.c1 : c.c0;
    const float s = c.s4, z = c.z4;
    if constexpr (k == 0) {
        c.m = d & M4;
        c.m1 = (d >> 4) & M4;
        if constexpr (W == 1) c.o[3] = mxq_pack_f16(c.a2, c.t1);       // the first word's last pair
    }
    else if constexpr (k == 1) { c.t0 = mxq_ubyte0(c.m); c.a0 = mxq_ubyte1(c.m); }
    else if constexpr (k == 2) { c.t0 = c.t0 - z; c.a0 = c.a0 - z; }
    else if constexpr (k == 3) { c.t0 = s * c.t0; c.a0 = s * c.a0; }
    else if constexpr (k == 4) { c.a1 = mxq_ubyte2(c.m); c.a2 = mxq_ubyte3(c.m); }
    else if constexpr (k == 5) { c.o[OB] = mxq_pack_f16(c.t0, c.a0); c.a1 = c.a1 - z; c.a2 = c.a2 - z; }
    else if constexpr (k == 6) { c.a1 = s * c.a1; c.a2 = s * c.a2; }
    else if constexpr (k == 7) { c.t0 = mxq_ubyte0(c.m1); c.a0 = mxq_ubyte1(c.m1); }
    else if constexpr (k == 8) { c.o[OB + 1] = mxq_pack_f16(c.a1, c.a2); c.t0 = c.t0 - z; c.a0 = c.a0 - z; }
    else if constexpr (k == 9) { c.t0 = s * c.t0; c.a0 = s * c.a0; }
    else if constexpr (k == 10) { c.a1 = mxq_ubyte2(c.m1); c.t1 = mxq_ubyte3(c.m1); }
    else if constexpr (k == 11) { c.o[OB + 2] = mxq_pack_f16(c.t0, c.a0); c.a1 = c.a1 - z; c.t1 = c.t1 - z; }
    else if constexpr (k == 12) { c.a2 = s * c.a1; c.t1 = s * c.t1; }
    else if constexpr (W == 1) { c.o[7] = mxq_pack_f16(c.a2, c.t1); }   // k == 13 of the second word (the first word's: its k == 0)
}
constexpr int CV_SLOTS = 28;      // MFMAs of sections C, E, G, I
template <int ABL, int LAYOUT, bool R4, int K>
__device__ __forceinline__ void cv_micro(Cv& c) {
    if constexpr ((ABL & ABL_FILL) != 0) {           // timing only: the op COUNT of the real step, no dependent chains
        if constexpr (R4 || K < 20) {
            if constexpr (K % 2 == 0) { c.t0 += 1.5f; c.t1 += 2.5f; }
            else { c.s += 1.5f; c.z += 2.5f; }
            if constexpr (R4 && K % 5 == 4) c.qs += 0.5f;
            if constexpr (K < 8) c.o[K] = (__builtin_amdgcn_alignbit(c.c0, c.c0, 3 * K + 1) & 0x83FF83FFu) | 0x24002400u;
        }
    } else if constexpr (R4) cv_micro4<K>(c);
    else if constexpr (K < 20) cv_micro2<LAYOUT, K>(c);
}
template <int ABL, int LAYOUT, bool R4, int K0, int K1>
__device__ __forceinline__ void cv_micro_range(Cv& c) {
    if constexpr (K0 < K1) {
        cv_micro<ABL, LAYOUT, R4, K0>(c);
        cv_micro_range<ABL, LAYOUT, R4, K0 + 1, K1>(c);
    }
}

// ... and the unit's 32 bytes into W16[tc & 1]: 16-byte slots 2q, 2q+1 of its row
__device__ __forceinline__ void cv_write(const Cv& c, char* smem, int tc) {
    char* wt = smem + OFF_W + (tc & 1) * W_STAGE;
    *(u32x4*)(wt + swz(c.row, 2 * c.q)) = (u32x4){c.o[0], c.o[1], c.o[2], c.o[3]};
    *(u32x4*)(wt + swz(c.row, 2 * c.q + 1)) = (u32x4){c.o[4], c.o[5], c.o[6], c.o[7]};
}

template <int ABL, int LAYOUT, bool R4>
__device__ __forceinline__ void cv_all(Cv& c, char* smem, int tc) {   // a whole unit at once (prologue)
    cv_read<LAYOUT, R4>(c, smem, tc);
    if constexpr (!(ABL & ABL_NO_CONV)) cv_micro_range<ABL, LAYOUT, R4, 0, CV_SLOTS>(c);
    cv_write(c, smem, tc);
}

// MFMAs (I, 0..3) of a fragment row, each followed by its conversion micro-step K0 + j and a scheduling fence
template <int ABL, int LAYOUT, bool R4, bool ARITH, int I, int K0>
__device__ __forceinline__ void mfma_row_cv(f32x4 (&acc)[4][4], const Frag4& wf, const Frag4& xf, Cv& cv) {
#define MXQ_SLOT(J)                                                                                     \
    acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[I], xf[J], acc[I][J], 0, 0, 0);              \
    if constexpr (ARITH && K0 >= 0) cv_micro<ABL, LAYOUT, R4, (K0 >= 0 ? K0 + J : 0)>(cv);                   \
    __builtin_amdgcn_sched_barrier(0);
    MXQ_SLOT(0) MXQ_SLOT(1) MXQ_SLOT(2) MXQ_SLOT(3)
#undef MXQ_SLOT
}

// ------------------------------------------------------------------------------------------------
// One K-step t >= 1: MFMAs of (t-1, kk=1) and (t, kk=0), fragment reads of step t, -- ISSUE -- the DMAs of x(t+2) and
// P(t+4) (x slot (t+2) % 3 and packed slot t % 4 were last read in step t-1), and -- CONV -- the conversion of chunk
// t+1 out of packed slot (t+1) % 4 (landed before the previous barrier) into W16[(t+1) & 1] (last read in step t-1).
// ------------------------------------------------------------------------------------------------
template <int ABL, int LAYOUT, bool R4, bool ISSUE, bool CONV>
__device__ __forceinline__ void mma_step(char* smem, int t, int wave, int lane, int wm, int wn, int fr, int fq,
                                         const XDma& xd, const PDma& pd, Cv& cv, f32x4 (&acc)[4][4], Frag4& wf0,
                                         Frag4& xf0, Frag4& wf1, Frag4& xf1) {
    constexpr bool ARITH = CONV && !(ABL & ABL_NO_CONV);
    constexpr bool CVLDS = CONV && !(ABL & ABL_NO_CVLDS);
    if constexpr (CVLDS) cv_read<LAYOUT, R4>(cv, smem, t + 1);              // A: the unit's packed words, 4 MFMAs
    MXQ_FENCE();
    mfma_row_cv<ABL, LAYOUT, R4, false, 0, -1>(acc, wf1, xf1, cv);
    load_frags(smem, t, 0, wm, wn, fr, fq, wf0, xf0);                      // B
    MXQ_FENCE();
    mfma_row_cv<ABL, LAYOUT, R4, ARITH, 1, 0>(acc, wf1, xf1, cv);                // C: micro-steps 0..3
    if constexpr (ISSUE) {                                                 // D
        issue_x<0, 2>(xd, smem, wave, t + 2);
    }
    MXQ_FENCE();
    mfma_row_cv<ABL, LAYOUT, R4, ARITH, 2, 4>(acc, wf1, xf1, cv);                // E: 4..11
    mfma_row_cv<ABL, LAYOUT, R4, ARITH, 3, 8>(acc, wf1, xf1, cv);
    load_frags(smem, t, 1, wm, wn, fr, fq, wf1, xf1);                      // F
    MXQ_FENCE();
    mfma_row_cv<ABL, LAYOUT, R4, ARITH, 0, 12>(acc, wf0, xf0, cv);               // G: 12..19
    mfma_row_cv<ABL, LAYOUT, R4, ARITH, 1, 16>(acc, wf0, xf0, cv);
    if constexpr (ISSUE) {                                                 // H
        issue_x<2, 4>(xd, smem, wave, t + 2);
        if constexpr (!(ABL & ABL_NO_PDMA)) issue_p<LAYOUT>(pd, smem, wave, t + 4);   // AFTER the step's x pieces: see the wait
    }
    MXQ_FENCE();
    if constexpr (CVLDS && !R4) cv_write(cv, smem, t + 1);                  // a 2-bit unit is complete after step 19
    mfma_row_cv<ABL, LAYOUT, R4, ARITH, 2, 20>(acc, wf0, xf0, cv);               // I: 20..27
    mfma_row_cv<ABL, LAYOUT, R4, ARITH, 3, 24>(acc, wf0, xf0, cv);
    if constexpr (CVLDS && R4) cv_write(cv, smem, t + 1);
    // vmcnt retires in order.  The packed blocks stream from HBM (every weight byte is read once), x comes out of L2 (32
    // workgroups share it): a packed-block DMA that precedes an x piece in the queue makes the wait for that piece a wait
    // for HBM (+4..7 us per launch when P was issued between the x pieces).  So P(t+4) is the LAST DMA of step t and the
    // wait leaves six in flight: P(t+3) from the previous step, this step's four x pieces and P(t+4).  x(t+1) has landed;
    // a P is covered by the wait of the step after the next one and converted in the step after that.
    if constexpr (ISSUE && (ABL & ABL_NO_PDMA)) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (ISSUE) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// the tile WITHOUT an LDS round trip (gemm8.hip: store_tile_xpose)
__device__ __forceinline__ void store_tile_xpose(const f32x4 (&acc)[4][4], uint16_t* __restrict__ y, int M, int N,
                                                 int m0, int n0, int wm, int wn, int fr, int fq) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    const int n = n0 + wn * 64 + fq * 16;
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
        const int m = m0 + wm * 64 + j * 16 + fr;
        if (m < M && n < N) {
            uint16_t* dst = y + (int64_t)m * N + n;
            __builtin_nontemporal_store((u32x4){c[0][0], c[0][1], c[1][0], c[1][1]}, (u32x4*)dst);
            __builtin_nontemporal_store((u32x4){c[2][0], c[2][1], c[3][0], c[3][1]}, (u32x4*)(dst + 8));
        }
    }
}

// a segment's prologue DMAs: x(0), P(0), P(1), x(1), P(2), P(3) -- in THIS order (the first wait leaves the last six in
// flight).  Needs both rings idle: every read of the previous segment's slots lies before that segment's last barrier.
template <int LAYOUT>
__device__ __forceinline__ void prologue_issue(const XDma& xd, const PDma& pd, char* smem, int wave, int NT) {
    issue_x<0, 4>(xd, smem, wave, 0);
    issue_p<LAYOUT>(pd, smem, wave, 0);
    if (NT > 1) {
        issue_p<LAYOUT>(pd, smem, wave, 1);
        issue_x<0, 4>(xd, smem, wave, 1);
    }
    if (NT > 2) issue_p<LAYOUT>(pd, smem, wave, 2);
    if (NT > 3) issue_p<LAYOUT>(pd, smem, wave, 3);
}

// One segment = NT K-steps of one tile.  pre: its prologue DMAs are already in flight (issued by the caller behind the
// previous segment's last barrier; the previous tile's output stores may sit in between, so the first wait is a full one).
// After the last barrier, BEFORE the final 16 MFMAs and the output, `next()` runs: the persistent loop issues the next
// tile's prologue DMAs there, which then fly under this tile's epilogue.
template <int ABL, int LAYOUT, bool R4, class Next>
__device__ __forceinline__ void mma_segment(char* smem, int wave, int lane, int NT, const XDma& xd, const PDma& pd,
                                            Cv& cv, bool pre, uint16_t* __restrict__ y, int M, int N, int m0, int n0,
                                            int NT_tile, const SkSeg& sk, Next&& next) {
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag4 wf0, xf0, wf1, xf1;

    if (!pre) prologue_issue<LAYOUT>(xd, pd, smem, wave, NT);
    if (!pre && NT > 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // x(0), P(0), P(1) landed; x(1), P(2), P(3) in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // prologue barrier 1: x(0), P(0), P(1) of every wave landed
    cv_all<ABL, LAYOUT, R4>(cv, smem, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // prologue barrier 2: W16(0) written

    // step 0: no previous half; chunk 1 is converted whole next to the first 16 MFMAs
    load_frags(smem, 0, 0, wm, wn, fr, fq, wf0, xf0);
    load_frags(smem, 0, 1, wm, wn, fr, fq, wf1, xf1);
    if (NT > 2) {
        issue_x<0, 4>(xd, smem, wave, 2);
        issue_p<LAYOUT>(pd, smem, wave, 4);
    }
    MXQ_FENCE();
    mfma_rows<0, 4>(acc, wf0, xf0);
    if (NT > 1) cv_all<ABL, LAYOUT, R4>(cv, smem, 1);
    if (NT > 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // x(1), P(2) landed; P(3), x(2), P(4) in flight
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int t = 1;
    for (; t + 2 < NT; ++t)
        mma_step<ABL, LAYOUT, R4, true, true>(smem, t, wave, lane, wm, wn, fr, fq, xd, pd, cv, acc, wf0, xf0, wf1, xf1);
    if (t + 1 < NT) {
        mma_step<ABL, LAYOUT, R4, false, true>(smem, t, wave, lane, wm, wn, fr, fq, xd, pd, cv, acc, wf0, xf0, wf1, xf1);
        ++t;
    }
    if (t < NT) mma_step<ABL, LAYOUT, R4, false, false>(smem, t, wave, lane, wm, wn, fr, fq, xd, pd, cv, acc, wf0, xf0, wf1, xf1);
    next();                                  // both rings are idle from here on
    mfma_rows<0, 4>(acc, wf1, xf1);          // (NT-1, kk=1)

    if (NT != NT_tile) {
        // partial segment: park the accumulators in this unit's slot; counted in after the unit's last segment
        float* mine = sk.ws + ((int64_t)((sk.u * 8 + sk.e) * 2 + (sk.first ? 0 : 1)) * (BM * BN)) + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) st_agent(mine, i * 4 + j, lane, acc[i][j]);
        return;
    }
    store_tile_xpose(acc, y, M, N, m0, n0, wm, wn, fr, fq);
}

// The wave that completed a tile's K-step count: sum every contributor's slot in unit order and write y (gemm8.hip).
__device__ __forceinline__ void sk_finish(const SkSeg& sk, int j, int NT_tile, int wave, int lane,
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
        f32x4 p[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) p[i][jj] = ld_agent(src, i * 4 + jj, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i][jj] = any ? acc[i][jj] + p[i][jj] : p[i][jj];
        __builtin_amdgcn_sched_barrier(0);
        any = true;
    }
    if (lane == 0)   // ready for the next launch
        __hip_atomic_store(sk.cnt + (j * 8 + sk.e) * NW + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    store_tile_xpose(acc, y, M, N, m0, n0, wave >> 1, wave & 1, lane & 15, lane >> 4);
}

#define MXQ_LANE_ID(ln) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln))

template <int LAYOUT>
__device__ __forceinline__ void cv_setup(Cv& cv, const float4* __restrict__ rowmeta, int N, int n0, int wave, int lane) {
    cv.q = wave >> 1;
    cv.row = (wave & 1) * 64 + lane;
    cv.r = cv.row & 15;
    if (Lay<LAYOUT>::is4(cv.q)) {
        int gn = n0 + cv.row;
        gn = gn < N ? gn : N - 1;
        const float4 rm = rowmeta[gn];
        cv.s4 = mxq_scale(rm.z, rm.w, (uint32_t)rm.y);
        cv.z4 = rm.x;
    }
}

template <int ABL, int LAYOUT, bool R4>
__device__ __forceinline__ void run(char* smem, int wave, const uint16_t* __restrict__ x,
                                    const uint32_t* __restrict__ qweight, const float4* __restrict__ rowmeta,
                                    uint16_t* __restrict__ y, int M, int N, int K, int tiles_m, int tiles_n, int dp_tiles,
                                    int dp_grid, int tail, int units, float* __restrict__ ws, int* __restrict__ cnt) {
    const int NT = K / BK;
    const int bid = blockIdx.x;
    SkSeg sk;
    sk.ws = ws;
    sk.cnt = cnt;
    sk.units = units;
    sk.S = 0;
    sk.u = sk.e = sk.j = sk.first = 0;
    auto nothing = [] {};
    Cv cv = {};

    if (bid < dp_grid) {
        // ---- persistent data-parallel workgroup: whole tiles bid, bid + dp_grid, ...
        int tm, tn;
        tile_of_block(bid, tiles_m, tiles_n, tm, tn);
        int ln;
        MXQ_LANE_ID(ln);
        XDma xcur, xnxt;
        PDma pcur, pnxt;
        xdma_setup(xcur, x, M, K, tm * BM, 0, wave, ln);
        pdma_setup<LAYOUT>(pcur, qweight, N, K, tn * BN, 0, wave, ln);
        prologue_issue<LAYOUT>(xcur, pcur, smem, wave, NT);
        for (int tile = bid; tile < dp_tiles; tile += dp_grid) {
            MXQ_LANE_ID(ln);   // recomputed per tile and opaque: nothing lane-derived is hoisted (and spilled) across the loop
            const int m0 = tm * BM, n0 = tn * BN;
            cv_setup<LAYOUT>(cv, rowmeta, N, n0, wave, ln);
            const bool more = tile + dp_grid < dp_tiles;
            if (more) tile_of_block(tile + dp_grid, tiles_m, tiles_n, tm, tn);
            mma_segment<ABL, LAYOUT, R4>(smem, wave, ln, NT, xcur, pcur, cv, true, y, M, N, m0, n0, NT, sk, [&] {
                if (more) {
                    xdma_setup(xnxt, x, M, K, tm * BM, 0, wave, ln);
                    pdma_setup<LAYOUT>(pnxt, qweight, N, K, tn * BN, 0, wave, ln);
                    prologue_issue<LAYOUT>(xnxt, pnxt, smem, wave, NT);
                }
            });
            xcur = xnxt;
            pcur = pnxt;
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
    int pj0 = -1, pn0 = 0, pj1 = -1, pn1 = 0;
    for (int pos = b0; pos < b1;) {
        sk.j = pos / NT;
        const int end = b1 < (sk.j + 1) * NT ? b1 : (sk.j + 1) * NT;
        sk.first = pos == b0;
        int tm, tn;
        tile_of_block(base + sk.j * 8, tiles_m, tiles_n, tm, tn);
        int ln;
        MXQ_LANE_ID(ln);
        XDma xd;
        PDma pd;
        xdma_setup(xd, x, M, K, tm * BM, pos - sk.j * NT, wave, ln);
        pdma_setup<LAYOUT>(pd, qweight, N, K, tn * BN, pos - sk.j * NT, wave, ln);
        cv_setup<LAYOUT>(cv, rowmeta, N, tn * BN, wave, ln);
        mma_segment<ABL, LAYOUT, R4>(smem, wave, ln, end - pos, xd, pd, cv, false, y, M, N, tm * BM, tn * BN, NT, sk, nothing);
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
            if (pj0 >= 0) old0 = __hip_atomic_fetch_add(cnt + (pj0 * 8 + sk.e) * NW + wave, pn0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pj1 >= 0) old1 = __hip_atomic_fetch_add(cnt + (pj1 * 8 + sk.e) * NW + wave, pn1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
}

template <int ABL, int LAYOUT>
__global__ __launch_bounds__(THREADS) void mxq_gemm10_f16_kernel(const uint16_t* __restrict__ x,
                                                               const uint32_t* __restrict__ qweight,
                                                               const float4* __restrict__ rowmeta,
                                                               uint16_t* __restrict__ y, int M, int N, int K,
                                                               int tiles_m, int tiles_n, int dp_tiles, int dp_grid,
                                                               int tail, int units, float* __restrict__ ws,
                                                               int* __restrict__ cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the unit a wave converts is 2-bit or 4-bit for the whole kernel: dispatch once, outside every loop
    if (Lay<LAYOUT>::is4(wave >> 1))
        run<ABL, LAYOUT, true>(smem, wave, x, qweight, rowmeta, y, M, N, K, tiles_m, tiles_n, dp_tiles, dp_grid, tail, units, ws, cnt);
    else if constexpr (LAYOUT != MXQ_LAYOUT_W4ROW)
        run<ABL, LAYOUT, false>(smem, wave, x, qweight, rowmeta, y, M, N, K, tiles_m, tiles_n, dp_tiles, dp_grid, tail, units, ws, cnt);
}

int cu_count10() {
    static int cus = 0;
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

constexpr size_t CNT_BYTES = 64 * 1024;   // K-step counters at the head of the workspace (gemm8's layout)

template <int ABL, int LAYOUT>
static int launch10(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   void* workspace, size_t ws_bytes, bool force, hipStream_t stream) {
    // 32-bit offsets: 256 rows of x per DMA descriptor; the whole packed weight behind one (offsets < 2^31)
    if ((int64_t)BM * K * 2 >= ((int64_t)1 << 32) || (int64_t)(N / 16) * (K / BK) * MXQ_BLK_BYTES >= ((int64_t)1 << 31))
        return -1;   // MXQ_E_SHAPE
    hipError_t e = hipFuncSetAttribute((const void*)mxq_gemm10_f16_kernel<ABL, LAYOUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return (int)e;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    const int NT = K / BK;
    const int cus = cu_count10() / 8 * 8;
    int units = cus / 8;
    int dp_tiles = tiles, tail = 0;
    if (workspace && tiles % cus != 0 && units * 8 * NW * sizeof(int) <= CNT_BYTES &&
        ws_bytes >= CNT_BYTES + (size_t)cus * 2 * BM * BN * sizeof(float)) {
        const int t8 = (tiles % cus) / 8;   // tail tiles per XCD (the first tail % 8 XCDs hold one more)
        // (gemm8.hip: when splitting the tail pays)
        const bool pays = (int64_t)(cus - tiles % cus) * NT >= (int64_t)24 * cus;
        if ((force || pays) && (int64_t)t8 * NT >= (int64_t)units * 4) {
            tail = tiles % cus;
            dp_tiles = tiles - tail;
        } else if (pays && tiles > cus && (int64_t)t8 * NT < (int64_t)units * 4) {
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
    const int dp_grid = dp_tiles < cus ? dp_tiles : cus;
    const int grid = dp_grid + (tail ? 8 * units : 0);
    mxq_gemm10_f16_kernel<ABL, LAYOUT><<<grid, THREADS, SMEM_BYTES, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K, tiles_m, tiles_n,
        dp_tiles, dp_grid, tail, units, (float*)((char*)workspace + CNT_BYTES), (int*)workspace);
    return (int)hipGetLastError();
}

}   // namespace

int mxq_launch_gemm10_layout_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                int layout, void* workspace, size_t ws_bytes, int force, hipStream_t stream) {
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launch10<0, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, force != 0, stream);
        case MXQ_LAYOUT_W2G16: return launch10<0, MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, force != 0, stream);
        case MXQ_LAYOUT_W4ROW: return launch10<0, MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, force != 0, stream);
        case MXQ_LAYOUT_MIXEDC: return launch10<0, MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, force != 0, stream);
    }
    return -1;
}

#ifdef MXQ_PROFILING
// Built only into libmxq_hip_prof.so: 4 = the conversion arithmetic removed (WRONG results, timing only)
extern "C" int mxq_prof_gemm10_ablate_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N,
                                         int K, int abl, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    switch (abl) {
        case 0: return launch10<0, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 4: return launch10<4, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 8: return launch10<8, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 24: return launch10<24, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 56: return launch10<56, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
        case 16: return launch10<16, MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, nullptr, 0, false, stream);
    }
    return -1;
}
#endif
