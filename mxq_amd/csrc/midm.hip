// W2/4 x A16 dequant-GEMM for the mid-size token regime (48 < M <= ~1024): split-K over workgroups, every wave both
// dequantises and multiplies, every operand arrives by LDS-DMA.
//
// Counterpart of the reference launcher's split-K regime (mxq_quant/cuda_kernel/csrc/quantization/gemm_cuda_gen.cu:429-475:
// `split_k_iters` K-slices whose [split_k, M, OC] partial outputs are summed afterwards; batch rows on gridDim.z in
// gemv_mxq_cuda.cu:261-262); arithmetic contract x16 . fp16(scale * (q - zero))^T of lib/quantizer.py:19-20 +
// mxqgpt.py:448, fp32 accumulation.
//
// Why its own kernel: with 64..1024 tokens a 4096-wide Linear has only 16..128 tiles of the prefill kernel's 256 x 128
// shape, a tile walks its 64 K-steps serially at ~1 us each (gemm8: the dequant chain of ONE wave per SIMD is the
// critical path of a K-step), and the stream-K tail's last-arriver reduction reads 0.5 MB per tile from one workgroup.
// Round 2 measured 28-34 us at every M in 64..512 on 4096^2 -- slower than hipBLASLt on the 16-bit weight (16-30 us).
// Here:
//   * tile BM (64 | 128) tokens x 128 channels, K cut into S slices so that tiles x S ~ the CU count; a workgroup
//     runs its slice's chunks two at a time ("double-step": 128 input channels per barrier);
//   * 8 waves, no roles: in a double-step every wave dequantises (row, column half H) of one of the two chunks for
//     64 of the 128 weight rows -- the 32-weight task of the prefill kernel's dequant waves, on all 8 waves at once
//     (H is wave-uniform; SIMD partners w and w + 4 take opposite halves, so every SIMD carries 62 + 91 VALU ops) --
//     into a double-buffered fp16 tile in LDS, and multiplies a (BM/2 tokens) x (32 channels) sub-tile of the
//     PREVIOUS double-step: 2 x TB accumulators, TB = BM/32 token blocks;
//   * BOTH operands arrive by LDS-DMA with full-width instructions: the activations in 1-KB pieces (8 full 128-B
//     rows, XOR-swizzled on the source side like the prefill kernel's x tile) into a ring of 2-3 slots, the PACKED
//     weight blocks raw (576 B = 36 lanes x 16 B per block, two per wave and double-step) into a ring of 3-5 slots,
//     from which a thread picks its 7-9 dwords with ds_read_b32.  What this is about (tools/midm_stamps.py, in-kernel
//     cycle stamps of the first builds, which loaded the packed words straight into registers as gemm8's dequant
//     waves do): a vector-memory instruction costs the CU's address pipe ~16 cycles whatever its width, and 8 waves x
//     (9 dword loads + 4 x pieces) per double-step kept the waves 1200-2100 cycles in ISSUE alone, of ~3000 per
//     double-step.  gemm8 has 4 such waves per 8 MFMA waves and twice the MFMA work per step; here every wave loads;
//   * one raw s_barrier per double-step; DMA completion by counted vmcnt (all vector-memory traffic of the loop is
//     DMA, so the counts are plain instruction counts);
//   * S == 1: fp16 output straight from the accumulators (one v_permlane16_swap per dword pairs two lanes' 8-byte
//     cells into 16-byte stores).  S > 1: fp32 partial tiles ("slabs") in fragment order, 1 KB contiguous per store
//     instruction, write-through stores; the kernel boundary publishes them, and a combine kernel spread over ALL CUs sums
//     the S slabs of every tile in slice order (deterministic: no atomics, no arrival order) and writes fp16 y.
// Rows beyond M / N and chunks beyond a slice's end read as zeros through the buffer descriptors' range checks.
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int BN = 128, THREADS = 512, WAVES = 8;
constexpr int LDS_MAX = 160 * 1024;
// timing-only ablations (tools/build_variant.sh ... -DMIDM_ABL=n; WRONG results; never in a shipped build):
// 1 no conversion arithmetic, 2 no MFMAs (fragment reads kept), 4 no fp16 weight-tile stores, 8 no x DMA,
// 16 no packed DMA, 32 no fragment reads and no MFMAs
#ifndef MIDM_ABL
#define MIDM_ABL 0
#endif

template <int BM, bool COMPACT>
struct Geo {
    static constexpr int TB = BM / 32;                 // token blocks of 16 per wave: wave tile = BM/2 tokens x 32 channels
    static constexpr int XL = BM / 32;                 // 1-KB activation pieces per wave and double-step
    static constexpr int XPC = BM / 64;                // ... per chunk
    static constexpr int XS_BYTES = 2 * BM * 128;      // activation slot [chunk 2][BM rows][128 B]
    static constexpr int WS_BYTES = 2 * BN * 128;      // fp16 weight stage [chunk 2][128 rows][128 B]
    static constexpr int BLK_B = COMPACT ? MXQC_BLK_BYTES : MXQ_BLK_BYTES;
    static constexpr int RS_BYTES = 16 * BLK_B;        // raw slot: the double-step's 16 packed blocks [chunk 2][row block 8]
    // x ring: DX + 1 slots (the DMA runs DX double-steps ahead of the reads); raw ring: DP slots (DP ahead of the
    // conversion).  128-token tiles leave room for 2 + 3, 64-token tiles for 3 + 5.
    static constexpr int DX = BM == 128 ? 1 : 2;
    static constexpr int XSLOTS = DX + 1;
    static constexpr int OFF_X = 0, OFF_W = XSLOTS * XS_BYTES, OFF_R = OFF_W + 2 * WS_BYTES;
    static constexpr int DP = (LDS_MAX - OFF_R) / RS_BYTES < 5 ? (LDS_MAX - OFF_R) / RS_BYTES : 5;
    static constexpr int SMEM = OFF_R + DP * RS_BYTES;
    static constexpr int SLAB = BM * BN;               // floats per partial tile
    static_assert(DP >= 2 && SMEM <= LDS_MAX, "LDS budget");
};

__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
// LDS-DMA, 16 B per lane: LDS destination = wave-uniform base + 16 * lane; source = descriptor base + voff + soff,
// out-of-range sources deliver zeros
__device__ __forceinline__ void dma16(rsrc_t rsrc, uint32_t voff, uint32_t soff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

// One thread's task: (row, column half H) of one chunk = 32 weights.  H = 0: 2-bit groups 0, 1; H = 1: group 2 and the
// two 4-bit code words.  blk: the row block's packed block in the raw ring (mxq_format.h field offsets).
// -> 16-byte slots 4 H .. 4 H + 3 of the row in the fp16 tile
template <bool COMPACT, int H>
__device__ __forceinline__ void convert_blk(const uint32_t* blk, int r, float s4, float z4, u32x4 (&res)[4]) {
    constexpr int QQ0 = COMPACT ? MXQC_OFF_QQ : MXQ_OFF_QQ;
    constexpr int g0 = H * 2;
    auto zf = [&](int g) -> float {
        if constexpr (COMPACT) return (float)__builtin_bit_cast(_Float16, ((const uint16_t*)blk)[mxqc_z2_u16(g, r)]);
        else return __uint_as_float(blk[mxq_z2(g, r)]);
    };
    const uint32_t scw = ((const uint16_t*)blk)[COMPACT ? mxqc_sc_u16(r) : mxq_sc_u16(r)];
    uint32_t o[8];
    mxq_deq2x16(blk[mxq_c2(g0, r)],
                mxq_scale(__uint_as_float(blk[QQ0 + g0 * 2]), __uint_as_float(blk[QQ0 + g0 * 2 + 1]), (scw >> (4 * g0)) & 15u),
                zf(g0), o);
    res[0] = (u32x4){o[0], o[1], o[2], o[3]};
    res[1] = (u32x4){o[4], o[5], o[6], o[7]};
    if constexpr (H == 0) {
        mxq_deq2x16(blk[mxq_c2(1, r)], mxq_scale(__uint_as_float(blk[QQ0 + 2]), __uint_as_float(blk[QQ0 + 3]), (scw >> 4) & 15u),
                    zf(1), o);
    } else {
        mxq_deq4x8(blk[mxq_c4(0, r)], s4, z4, o);
        mxq_deq4x8(blk[mxq_c4(1, r)], s4, z4, o + 4);
    }
    res[2] = (u32x4){o[0], o[1], o[2], o[3]};
    res[3] = (u32x4){o[4], o[5], o[6], o[7]};
}

// bijective XCD-contiguous order (blocks b and b + 8 share an XCD: give each XCD a contiguous run of work items, so
// that the workgroups of one XCD read the same K-slice of x; placement is speed only)
__device__ __forceinline__ int xcd_order(int bid, int total) {
    const int q = total >> 3, r = total & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int N>
struct IC { static constexpr int value = N; };

#ifdef MXQ_PROFILING
// tools/midm_stamps.py: [workgroup][4] = start, prologue published, K loop done, output stored (100 MHz wall clock)
__device__ unsigned long long* g_midm_stamps = nullptr;
#define MIDM_STAMP(i)                                                                            \
    if (g_midm_stamps != nullptr && threadIdx.x == 0) g_midm_stamps[blockIdx.x * 4 + (i)] = wall_clock64();
// per-(workgroup, wave) shader-cycle sums over the double-steps: issue, convert + publish, multiply, vmcnt wait, barrier, steps
__device__ unsigned long long* g_midm_cycles = nullptr;
__device__ __forceinline__ unsigned long long midm_clk() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MIDM_CLK(v) const unsigned long long v = g_midm_cycles != nullptr ? midm_clk() : 0ull;
#define MIDM_ACC(k, a, b) cyc[k] += (b) - (a);
#else
#define MIDM_STAMP(i)
#define MIDM_CLK(v)
#define MIDM_ACC(k, a, b)
#endif

template <int BM, bool COMPACT, int H>
__device__ __forceinline__ void midm_run(char* smem, const uint16_t* __restrict__ x, const uint32_t* __restrict__ qweight,
                                         const float4* __restrict__ rowmeta, uint16_t* __restrict__ y,
                                         float* __restrict__ part, int M, int N, int K, int tiles_m, int tiles_n, int S,
                                         int cps) {
    typedef Geo<BM, COMPACT> G;
    constexpr int TB = G::TB, XL = G::XL, DX = G::DX, DP = G::DP, BLK_B = G::BLK_B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NC = K >> 6;
    const int tiles = tiles_m * tiles_n;
    const int lin = xcd_order(blockIdx.x, tiles * S);
    const int s = lin / tiles, tile = lin - s * tiles;
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int c0 = s * cps;
    const int c1 = c0 + cps < NC ? c0 + cps : NC;
    const int nd = (c1 - c0 + 1) >> 1;                    // double-steps of this slice (>= 1: launcher)
    MIDM_STAMP(0)

    // ---- activations: XL pieces of 1 KB (8 full 128-B rows) per wave and double-step; the 16-byte slots of a row are
    // XOR-swizzled on the SOURCE side (the LDS image is lane-linear)
    const int rows = M - m0 < BM ? M - m0 : BM;
    const uint32_t xbytes = (uint32_t)rows * (uint32_t)K * 2u;
    uint32_t xvoff[XL];
    int xlds[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) {
        const int d = (i % G::XPC) * WAVES + wave;        // 8-row group inside the chunk
        const int row = d * 8 + (lane >> 3);
        xvoff[i] = (uint32_t)row * (uint32_t)K * 2u + ((((uint32_t)lane & 7u) ^ ((uint32_t)row & 7u)) << 4);
        xlds[i] = (i / G::XPC) * (BM * 128) + d * 1024;
    }
    auto load_x = [&](int t) __attribute__((always_inline)) {
        char* b = smem + G::OFF_X + (t % G::XSLOTS) * G::XS_BYTES;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            // a chunk beyond the slice's end: an EMPTY descriptor (every address out of range: zeros land, no traffic)
            const int chunk = c0 + 2 * t + i / G::XPC;
            if (!(MIDM_ABL & 8))
            dma16(make_rsrc(x + (int64_t)m0 * K, chunk < c1 ? xbytes : 0u), xvoff[i], (uint32_t)chunk * 128u, b + xlds[i]);
        }
    };

    // ---- packed weights, raw: wave w copies row block w's two blocks of the double-step (BLK_B / 16 lanes x 16 B each; a row
    // block's chunks are contiguous, its row blocks NC blocks apart).  A row block beyond N lies beyond the buffer:
    // zeros.  A chunk beyond the slice's end delivers another chunk's (finite) weights, against activations that are zero.
    const uint32_t blk_stride = (uint32_t)NC * BLK_B;
    const uint32_t wbytes = (uint32_t)(N >> 4) * blk_stride;
    const uint32_t pvoff = (uint32_t)((n0 >> 4) + wave) * blk_stride + (uint32_t)lane * 16u;
    auto load_p = [&](int t) __attribute__((always_inline)) {
        char* b = smem + G::OFF_R + (t % DP) * G::RS_BYTES + wave * BLK_B;
        const rsrc_t wr = make_rsrc(qweight, t < nd ? wbytes : 0u);      // past the slice: empty descriptor, as for x
        if (lane < BLK_B / 16 && !(MIDM_ABL & 16)) {     // 36 lanes (exact metadata) / 30 (compact)
            dma16(wr, pvoff, (uint32_t)(c0 + 2 * t) * BLK_B, b);
            dma16(wr, pvoff, (uint32_t)(c0 + 2 * t + 1) * BLK_B, b + 8 * BLK_B);
        }
    };
    constexpr int NPI = 2;                                // DMA instructions of one load_p

    // ---- dequant: wave -> (chunk parity, row half, column half H); thread -> one of its 64 rows
    const int par = wave >> 2, rh = (wave >> 1) & 1;
    const int row = rh * 64 + lane, r = row & 15;
    int gn = n0 + row;
    gn = gn < N ? gn : N - 1;
    const float4 rm = rowmeta[gn];
    const float s4 = mxq_scale(rm.z, rm.w, (uint32_t)rm.y), z4 = rm.x;
    auto convert = [&](int t, u32x4 (&res)[4]) __attribute__((always_inline)) {
        const uint32_t* blk = (const uint32_t*)(smem + G::OFF_R + (t % DP) * G::RS_BYTES + (par * 8 + (row >> 4)) * BLK_B);
        if (MIDM_ABL & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(res[i]));
        } else {
            convert_blk<COMPACT, H>(blk, r, s4, z4, res);
        }
    };
    auto store_w = [&](int buf, const u32x4 (&res)[4]) __attribute__((always_inline)) {
        char* b = smem + G::OFF_W + buf * G::WS_BYTES + par * (BN * 128);
        if (MIDM_ABL & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(res[i]));
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(b + swz(row, H * 4 + i)) = res[i];
    };

    // ---- MFMA: wave (wm, wn) owns tokens [wm BM/2, +BM/2) x channels [32 wn, +32); D^T = W . x^T
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[2][TB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int buf, int xslot) __attribute__((always_inline)) {
        if (MIDM_ABL & 32) return;
        const char* xb = smem + G::OFF_X + xslot * G::XS_BYTES;
        const char* wb = smem + G::OFF_W + buf * G::WS_BYTES;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                half8 wf[2], xf[TB];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    wf[i] = *(const half8*)(wb + cc * (BN * 128) + swz(wn * 32 + i * 16 + fr, kk * 4 + fq));
#pragma unroll
                for (int j = 0; j < TB; ++j)
                    xf[j] = *(const half8*)(xb + cc * (BM * 128) + swz(wm * (BM / 2) + j * 16 + fr, kk * 4 + fq));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TB; ++j) {
                        if (MIDM_ABL & 2) asm volatile("" ::"v"(wf[i]), "v"(xf[j]));
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
                    }
            }
    };

    // ---- prologue.  Request order: P(0), P(1), X(0) -- what the first double-step needs -- then the rest of the run-ahead
    load_p(0);
    load_p(1);
    load_x(0);
    if constexpr (DX > 1) load_x(1);
#pragma unroll
    for (int i = 2; i < DP; ++i) load_p(i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DX - 1) * XL + (DP - 2) * NPI) : "memory");
    __builtin_amdgcn_s_barrier();                          // P(0), P(1), X(0) are in LDS, for every wave
    {
        u32x4 res[4];
        convert(0, res);
        store_w(0, res);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // W(0) published
    MIDM_STAMP(1)

    // double-step t: request X(t + DX) and P(t + DP); dequantise P(t + 1) out of the raw ring and publish W(t + 1)
    // into the other weight stage; the MFMAs of step t; retire the DMAs that step t + 1 depends on; one barrier.
    // Hazards (every wave passes barrier B(t) at the end of step t):
    //   x slot (t + DX) % (DX + 1)  was last read by the MFMAs of step t - 1, before B(t - 1);
    //   raw slot (t + DP) % DP      held P(t), converted in step t - 1, before B(t - 1);
    //   weight stage (t + 1) & 1    was last read in step t - 1;
    //   step t + 1 reads X(t + 1) and converts P(t + 2): both must be retired by their issuers before B(t) (a staged
    //   buffer is read one barrier AFTER the wait that retires it).  In issue order the queue ends ... X(t + 1),
    //   [P(t + DP - 1) if DX = 2], X(t + DX) if DX = 2, P(t + DP): with DX = 1 only this step's P(t + DP) may stay in
    //   flight (NPI instructions), with DX = 2 also P(t - 1 + DP) and X(t + 2) (2 NPI + XL).  P(t + 2) is older.
#ifdef MXQ_PROFILING
    unsigned long long cyc[6] = {0, 0, 0, 0, 0, 0};
#endif
    for (int t = 0; t < nd; ++t) {
        MIDM_CLK(k0)
        load_x(t + DX);
        load_p(t + DP);
        __builtin_amdgcn_sched_barrier(0);
        MIDM_CLK(k1)
        const bool more = t + 1 < nd;
        if (more) {
            u32x4 res[4];
            convert(t + 1, res);
            store_w((t + 1) & 1, res);
        }
        __builtin_amdgcn_sched_barrier(0);
        MIDM_CLK(k2)
        mma(t & 1, t % G::XSLOTS);
        MIDM_CLK(k3)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DX == 1 ? NPI : 2 * NPI + XL) : "memory");
        MIDM_CLK(k4)
        __builtin_amdgcn_s_barrier();
        MIDM_CLK(k5)
        MIDM_ACC(0, k0, k1)
        MIDM_ACC(1, k1, k2)
        MIDM_ACC(2, k2, k3)
        MIDM_ACC(3, k3, k4)
        MIDM_ACC(4, k4, k5)
#ifdef MXQ_PROFILING
        cyc[5] += 1;
#endif
    }
    MIDM_STAMP(2)
#ifdef MXQ_PROFILING
    if (g_midm_cycles != nullptr && lane == 0) {
        unsigned long long* d = g_midm_cycles + ((int64_t)blockIdx.x * WAVES + wave) * 6;
        for (int k = 0; k < 6; ++k) d[k] = cyc[k];
    }
#endif

    // ---- epilogue
    if (S > 1) {
        // fp32 slab of (tile, slice): [wave][fragment i * TB + j][lane] x 16 bytes -- 1 KB contiguous per instruction
        float* slab = part + (int64_t)(tile * S + s) * G::SLAB + wave * (2 * TB * 256) + lane * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                // write-through (sc1): nothing dirty is left for the kernel boundary to flush before the combine
                // launch can start (2-4 % per launch pair against plain stores, same process)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), make_rsrc(slab - lane * 4, 2 * TB * 1024u),
                                                       (uint32_t)((i * TB + j) * 64 + lane) * 16u, 0u, 16);
            }
#ifdef MXQ_PROFILING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MIDM_STAMP(3)
#endif
        return;
    }
    // lane (fr, fq) holds token fr, channels 16 i + 4 fq .. + 3 of block i.  One v_permlane16_swap per dword pairs lane
    // rows (fq, fq ^ 1): even rows end with 8 consecutive channels of block 0, odd rows of block 1 -> 16-byte stores
    const int n = n0 + wn * 32 + ((fq & 1) << 4) + ((fq >> 1) << 3);
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        uint32_t a[2], b[2];
        a[0] = mxq_pack_f16(acc[0][j][0], acc[0][j][1]);
        a[1] = mxq_pack_f16(acc[0][j][2], acc[0][j][3]);
        b[0] = mxq_pack_f16(acc[1][j][0], acc[1][j][1]);
        b[1] = mxq_pack_f16(acc[1][j][2], acc[1][j][3]);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const u32x2 sw = __builtin_amdgcn_permlane16_swap(a[d], b[d], false, false);
            a[d] = sw[0];
            b[d] = sw[1];
        }
        const int m = m0 + wm * (BM / 2) + j * 16 + fr;
        if (m < M && n < N) *(u32x4*)(y + (int64_t)m * N + n) = (u32x4){a[0], a[1], b[0], b[1]};
    }
}

template <int BM, bool COMPACT>
__global__ __launch_bounds__(THREADS) void mxq_midm_f16_kernel(const uint16_t* __restrict__ x,
                                                              const uint32_t* __restrict__ qweight,
                                                              const float4* __restrict__ rowmeta,
                                                              uint16_t* __restrict__ y, float* __restrict__ part, int M,
                                                              int N, int K, int tiles_m, int tiles_n, int S, int cps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the column half a wave dequantises: wave-uniform, not a constant -- dispatched once, outside the loop
    if ((((wave & 1) ^ (wave >> 2)) & 1) == 0)
        midm_run<BM, COMPACT, 0>(smem, x, qweight, rowmeta, y, part, M, N, K, tiles_m, tiles_n, S, cps);
    else
        midm_run<BM, COMPACT, 1>(smem, x, qweight, rowmeta, y, part, M, N, K, tiles_m, tiles_n, S, cps);
}

// y = sum over the S slices' slabs, in slice order (fixed: the result does not depend on timing).  One wave per
// (tile, producing wave, token block j): both channel blocks of the lanes' fragments, so that the pairing of the fp16
// epilogue above applies.  Spread over the whole chip: tiles x 8 x TB wave tasks.
template <int BM>
__global__ __launch_bounds__(256) void mxq_midm_combine_kernel(const float* __restrict__ part, uint16_t* __restrict__ y,
                                                              int M, int N, int tiles_m, int tiles, int S) {
    typedef Geo<BM, false> G;
    constexpr int TB = G::TB;
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= tiles * WAVES * TB) return;
    const int tile = task / (WAVES * TB), rem = task - tile * (WAVES * TB);
    const int wave = rem / TB, j = rem - wave * TB;
    const float* src = part + (int64_t)tile * S * G::SLAB + wave * (2 * TB * 256) + lane * 4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    // U slices' loads go out together (U = 2, 4 or 8 by slice count; clamped, never branched around), then they are added
    // in slice order: a loop that loads, adds, loads ... is one dependent memory round trip per slice (4.9 us for 8
    // slices -- the launch is nothing but that latency; loading 8 where there are 2 costs 1 us the other way)
    auto sum_slices = [&](auto Uc) __attribute__((always_inline)) {
        constexpr int U = decltype(Uc)::value;
        for (int s0 = 0; s0 < S; s0 += U) {
            f32x4 v0[U], v1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sc = s0 + u < S ? s0 + u : S - 1;
                v0[u] = *(const f32x4*)(src + (int64_t)sc * G::SLAB + j * 256);
                v1[u] = *(const f32x4*)(src + (int64_t)sc * G::SLAB + (TB + j) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);   // (else the scheduler recycles the first loads' registers and waits for them)
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (s0 + u < S) {
                    a0 += v0[u];
                    a1 += v1[u];
                }
        }
    };
    if (S <= 2) sum_slices(IC<2>{});
    else if (S <= 4) sum_slices(IC<4>{});
    else sum_slices(IC<8>{});
    uint32_t a[2] = {mxq_pack_f16(a0[0], a0[1]), mxq_pack_f16(a0[2], a0[3])};
    uint32_t b[2] = {mxq_pack_f16(a1[0], a1[1]), mxq_pack_f16(a1[2], a1[3])};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const u32x2 sw = __builtin_amdgcn_permlane16_swap(a[d], b[d], false, false);
        a[d] = sw[0];
        b[d] = sw[1];
    }
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const int wm = wave & 1, wn = wave >> 1, fr = lane & 15, fq = lane >> 4;
    const int m = tm * BM + wm * (BM / 2) + j * 16 + fr;
    const int n = tn * BN + wn * 32 + ((fq & 1) << 4) + ((fq >> 1) << 3);
    if (m < M && n < N) *(u32x4*)(y + (int64_t)m * N + n) = (u32x4){a[0], a[1], b[0], b[1]};
}

int cu_count_() {
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

constexpr size_t WS_HEAD = 64 * 1024;   // the stream-K counters of gemm8 live at the head of the shared workspace: untouched here

template <int BM, bool COMPACT>
int launch_t(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, void* workspace,
             size_t ws_bytes, int splits, hipStream_t stream) {
    typedef Geo<BM, COMPACT> G;
    const int NC = K / 64;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, tiles = tiles_m * tiles_n;
    // slices: as many as keep tiles x S within one workgroup per CU, at least 2 double-steps (4 chunks) each
    int S = 1;
    if (workspace && ws_bytes > WS_HEAD) {
        S = splits > 0 ? splits : cu_count_() / tiles;
        if (S > NC / 4) S = NC / 4;
        const size_t room = (ws_bytes - WS_HEAD) / ((size_t)tiles * G::SLAB * sizeof(float));
        if ((size_t)S > room) S = (int)room;
        if (S < 1) S = 1;
    }
    int cps = (NC + S - 1) / S;
    cps += cps & 1;                                   // whole double-steps
    S = (NC + cps - 1) / cps;                         // no empty slice
    hipError_t e = mxq_set_dyn_lds_once<&mxq_midm_f16_kernel<BM, COMPACT>>(G::SMEM);
    if (e != hipSuccess) return (int)e;
    float* part = S > 1 ? (float*)((char*)workspace + WS_HEAD) : nullptr;
    mxq_midm_f16_kernel<BM, COMPACT><<<tiles * S, THREADS, G::SMEM, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, part, M, N, K, tiles_m,
        tiles_n, S, cps);
    if (S > 1) {
        const int tasks = tiles * WAVES * G::TB;
        mxq_midm_combine_kernel<BM><<<(tasks + 3) / 4, 256, 0, stream>>>(part, (uint16_t*)y, M, N, tiles_m, tiles, S);
    }
    return (int)hipGetLastError();
}

}   // namespace

#ifdef MXQ_PROFILING
extern "C" int mxq_prof_midm_set_cycles(void* cycles) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_midm_cycles), &cycles, sizeof(cycles));
}
extern "C" int mxq_prof_midm_set(void* stamps, int order) {
    (void)order;    // (the SIMD-partner stagger of the second build measured no gain and is gone)
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_midm_stamps), &stamps, sizeof(stamps));
}
// tools/ only (libmxq_hip_prof.so): explicit tile height (64 | 128) and K-slice count (correct results)
extern "C" int mxq_prof_midm_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                 int bm, int splits, void* workspace, size_t ws_bytes, void* stream) {
    return mxq_launch_midm_f16(x, qweight, rowmeta, y, M, N, K, MXQ_LAYOUT_MIXED, workspace, ws_bytes, bm, splits,
                               (hipStream_t)stream);
}
#endif

// 1 <= M; layout MXQ_LAYOUT_MIXED or MXQ_LAYOUT_MIXEDC.  workspace (nullable): the shared GEMM workspace
// (mxq_gemm_workspace_bytes); without one the K range is not split.  bm: 0 = by M, else 64 | 128; splits: 0 = by
// the CU count, else the number of K slices asked for (clamped to what K and the workspace allow).
int mxq_launch_midm_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, int layout,
                        void* workspace, size_t ws_bytes, int bm, int splits, hipStream_t stream) {
    if (layout != MXQ_LAYOUT_MIXED && layout != MXQ_LAYOUT_MIXEDC) return -1;
    if (bm != 0 && bm != 64 && bm != 128) return -1;
    // 32-bit buffer offsets: one tile's rows of x, the whole packed weight
    if ((int64_t)128 * K * 2 >= ((int64_t)1 << 31) || (int64_t)(N / 16) * (K / 64) * MXQ_BLK_BYTES >= ((int64_t)1 << 31))
        return -1;
    if (bm == 0) bm = M <= 64 ? 64 : 128;
    const bool c = layout == MXQ_LAYOUT_MIXEDC;
    if (bm == 64)
        return c ? launch_t<64, true>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, splits, stream)
                 : launch_t<64, false>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, splits, stream);
    return c ? launch_t<128, true>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, splits, stream)
             : launch_t<128, false>(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, splits, stream);
}
