// Decode-path GEMV (M <= 4 tokens) on the packed formats:
//   y[m, n] = sum_k x[m, k] * (scale * (q - zero))[n, k], fp32 accumulate, fp16 out.
//
// Native counterpart of the reference's fused unpack+dot kernel gemv_mxq_kernel_g16_v0
// (mxq_quant/cuda_kernel/csrc/quantization/gemv_mxq_cuda.cu:39-208) -- same idea (never materialise fp16 weights in
// memory), different everything else.  What round 2's measurements say bounds this kernel, and what the structure
// does about each (tools/gemv_stamps.py, tools/ab_gemv.py, tools/probes/stream_probe.hip):
//   * Not the HBM peak: a loads-only kernel in this access pattern streams 5.2-5.4 TB/s, weights served from the
//     Infinity Cache instead of HBM made the round-1 kernel only ~15 % faster, and halving its VALU work only ~7 %.
//   * Round trips: a wave that loads a tile, waits, computes, loads the next ... pays one loaded HBM latency
//     (2.5-4 us with every workgroup's requests queued at once) per tile, and the workgroup's prologue (activation
//     staging + barrier) waited for the first tile as well, because vector-memory loads retire in order (vmcnt).
//     So: each wave keeps TWO tiles in flight while it computes a third (three register sets, rotated by unrolling
//     -- no copies, a copy would wait for its load); the activation loads are issued BEFORE the first tiles, so the
//     staging waits for them alone; and nothing on the hot path branches around a load -- a branch makes the
//     compiler's wait counts conservative, i.e. a drain.  Instead: tiles are read through a buffer descriptor that
//     spans exactly the row block's K range (a tile index past the end returns zeros and costs no memory traffic),
//     and threads with nothing to stage write to a dummy LDS slot.
//   * Fewer, longer waves: 2 / 4 / 8 waves per workgroup by row-block count (every workgroup resident, ~2-3 k waves
//     on the chip, 2-8 tiles each) instead of 16 waves with one tile.
//   * VALU: the integer CODES go into v_dot2_f32_f16 and the group's scale / zero-point are applied to the group's
//     16-term sum,
//         sum_k x_k s (q_k - z)  =  s * (sum_k q_k x_k)  -  s z * (sum_k x_k),
//     with the per-group activation sums computed once per workgroup next to the staged activations.  A code becomes
//     an fp16 operand without a conversion: OR-ed into the top mantissa bits of 1.0 it reads 1 + q/4 (2-bit) or
//     1 + q/16 (4-bit), i.e. one shift and one v_and_or_b32 per TWO weights (the byte-spread code words put
//     elements k and k + 2 sixteen bits apart; the staged activations are permuted to (x0, x2, x1, x3) per four to
//     match).  ~120 VALU ops per 64 weights instead of ~230 for building the exact fp16 weights first (LUT +
//     4 v_perm per 4 weights).  The result is the fp32 sum over the UNROUNDED weights s (q - z): it differs from
//     the sum over the reference's fp16-rounded weights by the fp16 rounding of each weight (2^-11 relative,
//     independent per weight), ~1e-5 of the output scale at K = 4096 -- far inside the path's 1e-3 tolerance.
//   * Round 4: a workgroup takes RB CONSECUTIVE row blocks (1 / 2 / 4 by row-block count) and stages the activations ONCE
//     for all of them: the widest launches (q|k|v: 768 row blocks, gate|up: 1376) were 768 / 1376 small workgroups, each
//     waiting 2-4 us for its own copy of x, and the launch ended with the workgroups the dispatcher started last
//     (profiles/r03_gemv_dma_experiment.txt: median finish 12.2 us, max 16.1).  The RB row blocks' packed blocks are
//     contiguous, so the waves walk ONE flattened tile list (same wave count on the chip, same tiles per wave, two tiles
//     in flight + one computing across row-block boundaries); a wave whose next tile belongs to the next row block
//     leaves its partial sums in LDS, and one reduction at the end writes all RB x 16 outputs.
// lane -> (row r = lane & 15, chunk slot cs = lane >> 4); a wave consumes 4 consecutive 16x64 blocks (2304 contiguous
// bytes) per tile, straight to VGPRs; waves split the tile list.
#include <hip/hip_runtime.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_gemv_common.h"
#include "mxq_kernels.h"

namespace {

// cache policy of the WEIGHT loads (raw_buffer_load aux bits: 2 = nt).  The weights of a decode GEMV are read once,
// by one CU each; x, rowmeta and everything re-read stay on the default policy.  A/B: profiles/r03_gemv_nt_ab.txt
#ifndef MXQ_GEMV_WAUX
#define MXQ_GEMV_WAUX 0
#endif

#ifdef MXQ_PROFILING
// phase stamps of every workgroup (tools/gemv_stamps.py): [workgroup][4] = start, activations staged, weight loop
// done, result stored; 100 MHz wall clock (comparable across CUs)
__device__ unsigned long long* g_gemv_stamps = nullptr;
#define GEMV_STAMP(i)                                                                            \
    if (g_gemv_stamps != nullptr && threadIdx.x == 0) g_gemv_stamps[blockIdx.x * 4 + (i)] = wall_clock64();
#else
#define GEMV_STAMP(i)
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// PRO: prologue fused into the activation staging (decode, M = 1 only):
//   0 = none; 1 = RMSNorm: x <- fp16(x * norm_w), and the row's scalar rsqrt(mean(x^2) + eps) -- linear in the
//   output -- multiplies the 16 results at the end (so the staging does not wait for the reduction);
//   2 = SwiGLU gate: the input row is [2K] = (gate, up) and x <- fp16(silu(gate)) * up.
// residual (nullable): y <- residual + W.x (the decoder layer's skip connection).
// LAYOUT: MXQ_LAYOUT_MIXED / MIXEDC (3 two-bit groups + the 4-bit quarter per chunk; exact / compact metadata),
// MXQ_LAYOUT_W2G16 (4 two-bit groups) or MXQ_LAYOUT_W4ROW (4 four-bit quarters, scale / zero per row from rowmeta).
//   3 = the input row is ALREADY staged (round 5): fp16 values in the code-dot order + their per-group fp32 sums in `aux`
//   (float [K / 16]) -- what the EPI = 1 launch below writes; the staging is a plain copy into LDS.
// EPI 1 (round 5; RB == 2, M == 1): the weight is gate | up stacked ([2 * I, K]); the workgroup takes the PAIR of row blocks
// (i, i + I / 16) -- the same 16 rows of gate and of up -- and its final reduction applies SwiGLU itself: act = fp16(silu(gate))
// * up, the 16 values ONE scale group of the down projection's input, written in the code-dot order with their fp32 sum
// (y = the staged row [I], aux = the sums [I / 16]).  Bit for bit the values the PRO 2 staging computes from the fp16
// gate | up row, computed once by the producer instead of by every consumer workgroup (22 KB instead of 44 KB per
// consumer, no exp).
template <int MB, int GEMV_THREADS, int PRO, int LAYOUT = MXQ_LAYOUT_MIXED, int RB = 1, int EPI = 0>
__global__ __launch_bounds__(GEMV_THREADS) void mxq_gemv_f16_kernel(const uint16_t* __restrict__ x,
                                                                     const uint32_t* __restrict__ qweight,
                                                                     const float4* __restrict__ rowmeta,
                                                                     uint16_t* __restrict__ y, int M, int N, int K,
                                                                     const uint16_t* __restrict__ norm_w, float eps,
                                                                     const uint16_t* __restrict__ residual,
                                                                     float* __restrict__ aux) {
    static_assert(EPI == 0 || (RB == 2 && MB == 1), "the SwiGLU epilogue pairs two row blocks of one token");
    constexpr int W = GEMV_THREADS / 64;
    constexpr bool MIXED = LAYOUT == MXQ_LAYOUT_MIXED || LAYOUT == MXQ_LAYOUT_MIXEDC;   // exact / compact metadata
    constexpr bool COMPACT = LAYOUT == MXQ_LAYOUT_MIXEDC;
    constexpr int NG2 = MIXED ? 3 : LAYOUT == MXQ_LAYOUT_W2G16 ? 4 : 0;   // two-bit groups per chunk
    constexpr int NW4 = MIXED ? 2 : LAYOUT == MXQ_LAYOUT_W4ROW ? 8 : 0;   // four-bit code words per chunk
    constexpr int BLK_DW = LAYOUT == MXQ_LAYOUT_W4ROW ? 128 : COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    // LDS: x [MB][K] fp16 (code-dot order) | xsum [MB][K/16] f32 | red [RB][W][MB][16][3] f32 | wsum [W] f32 | dummy slots
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, cs = lane >> 4;
    const int rb0 = EPI ? blockIdx.x : blockIdx.x * RB;   // this workgroup's first row block
    const int PS = EPI ? N / 32 : 1;                      // row-block distance between the workgroup's row blocks
    const int NC = K / 64, NC4 = (NC + 3) / 4, NG = K / 16;
    const int nrb = EPI ? 2 : min(RB, N / 16 - rb0);      // ... and how many it has (the last workgroup may have fewer)
    const int NT = nrb * NC4;                           // its tile list (RB > 1 only when K % 256 == 0: tiles never straddle)
    GEMV_STAMP(0)

    float* xsum = (float*)(smem + (size_t)MB * K * 2);
    float* red = xsum + MB * NG;
    float* wsum = red + RB * W * MB * 16 * 3;
    char* dummy = (char*)(wsum + W);                    // [64 lanes][32 B] + [64] f32: where idle threads "stage"

    // ---- activation loads first (L2-resident, small), 2 staging steps hoisted; branch-free (clamped addresses)
    struct Act {
        h8 a0, a1;     // 16 activations (one scale group)
        h8 b0, b1;     // PRO 1: their RMSNorm weights; PRO 2: the up projection (a = gate)
        float sum;     // PRO 3: the group's fp32 sum, as staged by the producer
    };
    auto load_act = [&](int i) {
        Act t;
        const int ic = min(i, MB * NG - 1);
        const int m = ic / NG, g = ic % NG;
        const uint16_t* xr = x + (int64_t)min(m, M - 1) * K + g * 16;      // PRO != 0: one token, m == 0
        t.a0 = *(const h8*)xr;
        t.a1 = *(const h8*)(xr + 8);
        if constexpr (PRO == 1) {
            t.b0 = *(const h8*)(norm_w + g * 16);
            t.b1 = *(const h8*)(norm_w + g * 16 + 8);
        }
        if constexpr (PRO == 2) {
            t.b0 = *(const h8*)(xr + K);
            t.b1 = *(const h8*)(xr + K + 8);
        }
        if constexpr (PRO == 3) t.sum = aux[g];
        return t;                                       // (nothing here may USE a loaded value: that would be a wait)
    };
    const Act act0 = load_act(tid), act1 = load_act(tid + GEMV_THREADS);

    // ---- the first two weight tiles right behind them.  A tile = 4 consecutive blocks (one per chunk slot).
    struct Tile {
        uint32_t c2w[NG2 ? NG2 : 1], z2w[NG2 ? NG2 : 1], c4w[NW4 ? NW4 : 1], scw;
        uint2 qq[NG2 ? NG2 : 1];
    };
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(qweight + (int64_t)rb0 * NC * BLK_DW), 0, EPI ? (PS + 1) * NC * BLK_DW * 4 : nrb * NC * BLK_DW * 4, 0x00020000);
    const int lane_off = (cs * BLK_DW) * 4;             // byte offset of the lane's block inside a tile
    auto load_tile = [&](int c4) {                      // tile index in the list; beyond its end (or a chunk >= NC): zeros, no traffic
        Tile t = {};
        int so = c4 * (4 * BLK_DW * 4);                 // wave-uniform
        if constexpr (EPI) {   // the second row block lies PS row blocks on; a tile past the list: an offset beyond the buffer
            so = c4 < NC4 ? so : c4 < 2 * NC4 ? (PS * NC * BLK_DW + (c4 - NC4) * 4 * BLK_DW) * 4 : (int)0xC0000000u;
        }
        auto dw = [&](int idx) { return __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off + idx * 4, so, MXQ_GEMV_WAUX); };
        if constexpr (LAYOUT == MXQ_LAYOUT_W4ROW) {
#pragma unroll
            for (int i = 0; i < 8; ++i) t.c4w[i] = dw(mxq_w4_c4(i >> 1, i & 1, r));
        } else {
#pragma unroll
            for (int g = 0; g < NG2; ++g) {
                t.c2w[g] = dw(MIXED ? mxq_c2(g, r) : mxq_w2_c2(g, r));
                if constexpr (COMPACT)                  // fp16 zero-point (widened where it is used)
                    t.z2w[g] = __builtin_amdgcn_raw_buffer_load_b16(rs, lane_off + mxqc_z2_u16(g, r) * 2, so, MXQ_GEMV_WAUX);
                else
                    t.z2w[g] = dw(MIXED ? mxq_z2(g, r) : mxq_w2_z2(g, r));
                const int q = COMPACT ? mxqc_qq(g) : mxq_qq(g);     // SC / QQ: same offsets in v1 and W2G16
                t.qq[g] = make_uint2(dw(q), dw(q + 1));
            }
            if constexpr (MIXED) {
                t.c4w[0] = dw(mxq_c4(0, r));
                t.c4w[1] = dw(mxq_c4(1, r));
            }
            t.scw = __builtin_amdgcn_raw_buffer_load_b16(rs, lane_off + (COMPACT ? mxqc_sc_u16(r) : mxq_sc_u16(r)) * 2, so, MXQ_GEMV_WAUX);
        }
        return t;
    };
    // (the sched_barriers pin the issue order the wait counts are computed from: activations, row metadata, all of
    // tile 0, all of tile 1, and only then the staging arithmetic -- left alone, the scheduler sinks a tile-0 load
    // behind tile 1, and the loop's first wait then covers most of tile 1 in EVERY iteration)
    // the 4-bit arm's per-row parameters are applied in the final reduction: thread (row block, token, row) loads its own
    const int f_rbl = tid / (MB * 16), f_m = (tid >> 4) % MB, f_r = tid & 15;
    const float4 rm = rowmeta[min(rb0 + min(f_rbl, nrb - 1) * PS, N / 16 - 1) * 16 + f_r];
    __builtin_amdgcn_sched_barrier(0);
    Tile T0 = load_tile(wave);
    __builtin_amdgcn_sched_barrier(0);
    Tile T1 = load_tile(wave + W), T2;
    __builtin_amdgcn_sched_barrier(0);

    // ---- stage the activations: code-dot order + per-group fp32 sums; idle threads write to their dummy slot
    float ss = 0.f;
    // wlive (wave-uniform): some thread of this wave has a group to stage.  A wave without one skips the staging ARITHMETIC (round 6:
    // ~50-100 vector-ALU ops per call on every wave of the workgroup, in the launch's serial prologue; with K = 4096 and 512
    // threads three quarters of it ran on clamped copies) but still consumes every loaded value -- it writes their raw bits to its
    // dummy slots -- so that no load can be sunk into the conditional part behind the weight tiles' loads (a first version that
    // skipped the whole call had its activation loads moved there by the compiler: a drain of both tiles before the staging,
    // and the layer got slower although this part got faster).  Bit-identical.
    auto stage_act = [&](int i, Act t, bool wlive) {
        const bool live = i < MB * NG;                  // i = m * NG + g: rows are K * 2 = NG * 32 bytes
        float sum = 0.f;
        uint4 o0 = __builtin_bit_cast(uint4, t.a0), o1 = __builtin_bit_cast(uint4, t.a1);
        if (wlive) {
        if (MB > 1 && min(i, MB * NG - 1) / NG >= M) {  // rows beyond M are staged as zeros (select, not branch)
            const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            t.a0 = zero;
            t.a1 = zero;
        }
        if constexpr (PRO == 1) {
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += (float)t.a0[j] * (float)t.a0[j] + (float)t.a1[j] * (float)t.a1[j];
            ss += live ? sq : 0.f;                      // (an idle thread holds a clamped copy of the last group)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                t.a0[j] = t.a0[j] * t.b0[j];
                t.a1[j] = t.a1[j] * t.b1[j];
            }
        }
        if constexpr (PRO == 2) {                      // SwiGLU: x <- fp16(silu(gate)) * up
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g0 = (float)t.a0[j], g1 = (float)t.a1[j];
                t.a0[j] = (_Float16)(g0 / (1.0f + __expf(-g0))) * t.b0[j];
                t.a1[j] = (_Float16)(g1 / (1.0f + __expf(-g1))) * t.b1[j];
            }
        }
        if constexpr (PRO == 3) {                      // staged by the producer: a plain copy
            o0 = __builtin_bit_cast(uint4, t.a0);
            o1 = __builtin_bit_cast(uint4, t.a1);
            sum = t.sum;
        } else {
            o0 = stage8(__builtin_bit_cast(uint4, t.a0), sum);
            o1 = stage8(__builtin_bit_cast(uint4, t.a1), sum);
        }
        } else {                                        // an idle wave: every loaded value still goes somewhere
            if constexpr (PRO == 1 || PRO == 2) {
                const uint4 b0 = __builtin_bit_cast(uint4, t.b0), b1 = __builtin_bit_cast(uint4, t.b1);
                o0.x ^= b0.x ^ b0.y ^ b0.z ^ b0.w;
                o1.x ^= b1.x ^ b1.y ^ b1.z ^ b1.w;
            }
            if constexpr (PRO == 3) sum = t.sum;
        }
        char* dst = live ? smem + (size_t)i * 32 : dummy + lane * 32;
        float* sdst = live ? xsum + i : (float*)(dummy + 64 * 32) + lane;
        *(uint4*)dst = o0;
        *(uint4*)(dst + 16) = o1;
        *sdst = sum;
    };
    stage_act(tid, act0, wave * 64 < MB * NG);
    stage_act(tid + GEMV_THREADS, act1, GEMV_THREADS + wave * 64 < MB * NG);
    for (int i = tid + 2 * GEMV_THREADS; i < MB * NG; i += GEMV_THREADS) stage_act(i, load_act(i), true);   // M > 1 / odd shapes
    if constexpr (PRO == 1) {
        ss = wave_allsum(ss);                           // (vector-ALU butterfly, bit-identical to the shuffles: mxq_gemv_common.h)
        if (lane == 0) wsum[wave] = ss;
    }
    __syncthreads();
    GEMV_STAMP(1)

    // P: sum_g s_g * D'_g,  Q: sum_g s_g (1 + z_g / 4) X_g   (two-bit groups; y = 4 (P - Q))
    // R: sum D'4 over every four-bit quarter of the row, X4: sum of their activation sums (y += 16 s4 (R - (1 + z4 / 16) X4))
    float P[MB], Q[MB], R[MB], X4[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) P[m] = Q[m] = R[m] = X4[m] = 0.f;
    // a wave's partial sums of local row block `rbl`: over its 4 chunk slots, then into LDS -- {4 (P - Q), R, X4} per
    // (token, row); the row's s4 / z4 enter in the final reduction
    auto flush = [&](int rbl) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float a = 4.0f * (P[m] - Q[m]), b = R[m], c = X4[m];
            a = swap16_sum(a); b = swap16_sum(b); c = swap16_sum(c);     // + lane ^ 16, then + lane ^ 32: no LDS round trips
            a = swap32_sum(a); b = swap32_sum(b); c = swap32_sum(c);
            if (cs == 0) {
                float* d = red + (((rbl * W + wave) * MB + m) * 16 + r) * 3;
                d[0] = a; d[1] = b; d[2] = c;
            }
            P[m] = Q[m] = R[m] = X4[m] = 0.f;
        }
    };
    int cur_rbl = wave / NC4;                          // row block of the wave's first tile (wave-uniform, as every tile index)
    auto compute = [&](int ti, const Tile& t) {
        if (ti < NT) {                                  // no memory ops inside: the wait counts stay exact
            const int rbl = ti / NC4;
            if (RB > 1 && rbl != cur_rbl) {
                flush(cur_rbl);
                cur_rbl = rbl;
            }
            const int chunk = (ti - rbl * NC4) * 4 + cs;
            if (chunk < NC) {
            const char* xk = smem + (size_t)chunk * 128;
            const float* xg = xsum + chunk * 4;
#pragma unroll
            for (int g = 0; g < NG2; ++g) {
                const float s = mxq_scale(__uint_as_float(t.qq[g].x), __uint_as_float(t.qq[g].y),
                                          (t.scw >> (4 * g)) & 15u);
                float z;
                if constexpr (COMPACT) z = (float)__builtin_bit_cast(_Float16, (uint16_t)t.z2w[g]);
                else z = __uint_as_float(t.z2w[g]);
                const float sz = s * __builtin_fmaf(z, 0.25f, 1.0f);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const uint4 xa = *(const uint4*)(xk + (size_t)m * K * 2 + g * 32);
                    const uint4 xb = *(const uint4*)(xk + (size_t)m * K * 2 + g * 32 + 16);
                    P[m] = __builtin_fmaf(s, codedot2x16(t.c2w[g], xa, xb, 0.f), P[m]);
                    Q[m] = __builtin_fmaf(sz, xg[m * NG + g], Q[m]);
                }
            }
#pragma unroll
            for (int q = 0; q < NW4 / 2; ++q) {      // 16 four-bit weights per pair of code words
                constexpr int G0 = MIXED ? 3 : 0;    // the mixed layout's quarter is the chunk's last
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const uint4 xa = *(const uint4*)(xk + (size_t)m * K * 2 + (G0 + q) * 32);
                    const uint4 xb = *(const uint4*)(xk + (size_t)m * K * 2 + (G0 + q) * 32 + 16);
                    R[m] = codedot4x8(t.c4w[2 * q], xa, R[m]);
                    R[m] = codedot4x8(t.c4w[2 * q + 1], xb, R[m]);
                    X4[m] += xg[m * NG + G0 + q];
                }
            }
            }
        }
    };
    // two tiles in flight while a third is computed; the register sets rotate by unrolling (no copies)
    for (int c4 = wave; c4 < NT; c4 += 3 * W) {
        T2 = load_tile(c4 + 2 * W);
        compute(c4, T0);
        T0 = load_tile(c4 + 3 * W);
        compute(c4 + W, T1);
        T1 = load_tile(c4 + 4 * W);
        compute(c4 + 2 * W, T2);
    }
    GEMV_STAMP(2)
    // (a wave without a tile of some row block -- more waves than tiles, odd shapes -- must still leave zeros there)
    if (wave < NT) flush(cur_rbl);
    for (int rbl = 0; rbl < nrb; ++rbl) {
        // wave w holds tiles w, w + W, ...: it touched row block rbl iff one of them lies in [rbl NC4, (rbl + 1) NC4)
        const int first = wave >= rbl * NC4 ? wave : wave + ((rbl * NC4 - wave + W - 1) / W) * W;
        if (first >= (rbl + 1) * NC4 || first >= NT) {
            if (cs == 0) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    float* d = red + (((rbl * W + wave) * MB + m) * 16 + r) * 3;
                    d[0] = 0.f; d[1] = 0.f; d[2] = 0.f;
                }
            }
        }
    }
    __syncthreads();
    if (tid < nrb * MB * 16) {
        const float s4 = mxq_scale(rm.z, rm.w, (uint32_t)rm.y), z4 = rm.x;
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const float* d = red + (((f_rbl * W + w) * MB + f_m) * 16 + f_r) * 3;
            a += d[0]; b += d[1]; c += d[2];
        }
        float v = 0.f;
        if constexpr (NG2 > 0) v = a;
        if constexpr (NW4 > 0) v += 16.0f * s4 * (b - __builtin_fmaf(z4, 0.0625f, 1.0f) * c);
        if constexpr (PRO == 1) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) tot += wsum[w];
            v *= rsqrtf(tot / (float)K + eps);
        }
        if constexpr (EPI == 1) {
            // threads 0..15 hold the gate rows, 16..31 the up rows of the pair (one wave): SwiGLU exactly as the PRO 2
            // staging computes it from the fp16 row, then the code-dot order (x0, x2, x1, x3, x4, x6, x5, x7 per eight) and the
            // group's fp32 sum in element order
            const _Float16 h = (_Float16)v;
            const uint32_t hb = __builtin_bit_cast(uint16_t, h);
            const uint32_t ub = __shfl(hb, (lane + 16) & 31, 64);
            const float g0 = (float)h;
            const _Float16 act = (_Float16)(g0 / (1.0f + __expf(-g0))) * __builtin_bit_cast(_Float16, (uint16_t)ub);
            const uint32_t ab = __builtin_bit_cast(uint16_t, act);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += (float)__builtin_bit_cast(_Float16, (uint16_t)__shfl(ab, r, 64));
            if (tid < 16) {
                const int e = f_r & 7, pos = (f_r & 8) | ((e == 1 || e == 2) ? (e ^ 3) : (e == 5 || e == 6) ? (e ^ 3) : e);
                y[rb0 * 16 + pos] = (uint16_t)ab;
                if (tid == 0) aux[rb0] = sum;
            }
        } else if (f_m < M) {
            const int n = (rb0 + f_rbl) * 16 + f_r;
            _Float16 h = (_Float16)v;
            if (residual) h = __builtin_bit_cast(_Float16, residual[(int64_t)f_m * N + n]) + h;
            y[(int64_t)f_m * N + n] = __builtin_bit_cast(uint16_t, h);
        }
    }
    GEMV_STAMP(3)
}

template <int MB, int THREADS, int PRO, int LAYOUT = MXQ_LAYOUT_MIXED, int RB = 1, int EPI = 0>
int launch_t(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
             const void* norm_w, float eps, const void* residual, hipStream_t stream, void* aux = nullptr) {
    constexpr int W = THREADS / 64;
    static_assert(RB * MB * 16 <= THREADS, "one thread per output of the final reduction");
    const size_t smem = (size_t)MB * K * 2 + (size_t)MB * (K / 16) * 4 + (size_t)RB * W * MB * 16 * 3 * 4 + (size_t)W * 4 +
                        64 * 32 + 64 * 4;
    if (smem > 64 * 1024) {
        // (the attribute is a ceiling: the CU's whole LDS, set once per kernel and device)
        hipError_t e = mxq_set_dyn_lds_once<&mxq_gemv_f16_kernel<MB, THREADS, PRO, LAYOUT, RB, EPI>>(160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    const int rbs = N / 16;
    mxq_gemv_f16_kernel<MB, THREADS, PRO, LAYOUT, RB, EPI><<<EPI ? rbs / 2 : (rbs + RB - 1) / RB, THREADS, smem, stream>>>(
        (const uint16_t*)x, (const uint32_t*)qweight, (const float4*)rowmeta, (uint16_t*)y, M, N, K,
        (const uint16_t*)norm_w, eps, (const uint16_t*)residual, (float*)aux);
    return (int)hipGetLastError();
}

// Workgroup shape by row-block count: every workgroup resident at once (~5 waves per SIMD at ~90 VGPRs), some 2-3 thousand
// waves on the chip, each with several tiles to pipeline -- 8 waves per row block up to 384 row blocks (N = 4096; K = 4096:
// 2 tiles each, all in flight from the start; K = 11008: 5-6 each), 4 per row block up to 768 (q|k|v), 2 beyond (gate|up:
// 1376).  Round 4 (profiles/r04_gemv_rowblocks.txt): workgroups of RB consecutive row blocks staging the activations once
// pay only where the workgroup COUNT stays high -- gate|up as 688 workgroups of 2 row blocks x 4 waves: 16.65 vs 17.24 us;
// every shape that drops to <= 1.5 workgroups per CU loses 15-30 % (a CU with two 8-wave workgroups next to CUs with one
// finishes last: the launch is bound per CU, not per chip).  Code: RB in the high bits.
__host__ inline int gemv_shape(int N, int K, int forced) {
    if (forced) return forced;
    const int rbs = N / 16;
    if (rbs <= 384) return 512;
    if (rbs <= 768) return 256;
    return K % 256 == 0 ? (2 << 16) | 256 : 128;
}

#define MXQ_GEMV_DISPATCH(MB, PRO, LAYOUT, TH)                                                                                  \
    ((TH) == 512              ? launch_t<MB, 512, PRO, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream)      \
     : (TH) == 256            ? launch_t<MB, 256, PRO, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream)      \
     : (TH) == 128            ? launch_t<MB, 128, PRO, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream)      \
     : (TH) == ((2 << 16) | 512) ? launch_t<MB, 512, PRO, LAYOUT, 2>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream) \
     : (TH) == ((4 << 16) | 512) ? launch_t<MB, 512, PRO, LAYOUT, 4>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream) \
     : (TH) == ((2 << 16) | 256) ? launch_t<MB, 256, PRO, LAYOUT, 2>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream) \
     : (TH) == ((4 << 16) | 256) ? launch_t<MB, 256, PRO, LAYOUT, 4>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream) \
                              : (int)hipErrorInvalidValue)

template <int MB, int LAYOUT = MXQ_LAYOUT_MIXED>
int launch(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, hipStream_t stream,
           int threads = 0) {
    const void *norm_w = nullptr, *residual = nullptr;
    const float eps = 0.f;
    const int th = gemv_shape(N, K, threads);
    return MXQ_GEMV_DISPATCH(MB, 0, LAYOUT, th);
}

template <int LAYOUT>
int launch_layout(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                  hipStream_t stream) {
    if (M == 1) return launch<1, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    if (M == 2) return launch<2, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    if (M <= 4) return launch<4, LAYOUT>(x, qweight, rowmeta, y, M, N, K, stream);
    return (int)hipErrorInvalidValue;
}

}   // namespace

int mxq_launch_gemv_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                        hipStream_t stream) {
    return launch_layout<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, stream);
}

int mxq_launch_gemv_layout_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                               int layout, hipStream_t stream) {
    switch (layout) {
        case MXQ_LAYOUT_MIXED: return launch_layout<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_W2G16: return launch_layout<MXQ_LAYOUT_W2G16>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_W4ROW: return launch_layout<MXQ_LAYOUT_W4ROW>(x, qweight, rowmeta, y, M, N, K, stream);
        case MXQ_LAYOUT_MIXEDC: return launch_layout<MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, M, N, K, stream);
    }
    return (int)hipErrorInvalidValue;
}

template <int LAYOUT>
static int fused_layout(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K, int prologue,
                        const void* norm_w, float eps, const void* residual, hipStream_t stream, int threads = 0) {
    const int th = gemv_shape(N, K, threads), M = 1;
    switch (prologue) {
        case 0: return MXQ_GEMV_DISPATCH(1, 0, LAYOUT, th);
        case 1: return MXQ_GEMV_DISPATCH(1, 1, LAYOUT, th);
        case 2: return MXQ_GEMV_DISPATCH(1, 2, LAYOUT, th);
    }
    return (int)hipErrorInvalidValue;
}

int mxq_launch_gemv_fused_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K,
                              int prologue, const void* norm_w, float eps, const void* residual, int compact,
                              hipStream_t stream) {
    return compact ? fused_layout<MXQ_LAYOUT_MIXEDC>(x, qweight, rowmeta, y, N, K, prologue, norm_w, eps, residual, stream)
                   : fused_layout<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, N, K, prologue, norm_w, eps, residual, stream);
}

// gate | up of ONE token with the SwiGLU applied in the final reduction (EPI 1 above): x fp16 [K] (RMSNorm prologue), weight
// [2 I, K] = gate stacked on up, act fp16 [I] in the code-dot order, act_sum f32 [I / 16].  K % 256 == 0, (2 I) % 32 == 0.
int mxq_launch_gemv_swiglu_f16(const void* x, const void* qweight, const void* rowmeta, void* act, void* act_sum, int N2, int K,
                               const void* norm_w, float eps, int compact, hipStream_t stream) {
    const int M = 1;
    const void* residual = nullptr;
    if (N2 % 32 != 0 || K % 256 != 0) return -1;
    return compact ? launch_t<1, 256, 1, MXQ_LAYOUT_MIXEDC, 2, 1>(x, qweight, rowmeta, act, M, N2, K, norm_w, eps, residual, stream, act_sum)
                   : launch_t<1, 256, 1, MXQ_LAYOUT_MIXED, 2, 1>(x, qweight, rowmeta, act, M, N2, K, norm_w, eps, residual, stream, act_sum);
}

// ... and the Linear that consumes such a staged row (PRO 3): y fp16 [N] = residual + W . act
template <int LAYOUT>
static int staged_layout(const void* x, const void* x_sum, const void* qweight, const void* rowmeta, void* y, int N, int K,
                         const void* residual, hipStream_t stream) {
    const int th = gemv_shape(N, K, 0), M = 1;
    const void* norm_w = nullptr;
    const float eps = 0.f;
    void* aux = const_cast<void*>(x_sum);
    switch (th) {
        case 512: return launch_t<1, 512, 3, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream, aux);
        case 256: return launch_t<1, 256, 3, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream, aux);
        case 128: return launch_t<1, 128, 3, LAYOUT>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream, aux);
        case (2 << 16) | 256: return launch_t<1, 256, 3, LAYOUT, 2>(x, qweight, rowmeta, y, M, N, K, norm_w, eps, residual, stream, aux);
    }
    return (int)hipErrorInvalidValue;
}
int mxq_launch_gemv_staged_f16(const void* x, const void* x_sum, const void* qweight, const void* rowmeta, void* y, int N, int K,
                               const void* residual, int compact, hipStream_t stream) {
    return compact ? staged_layout<MXQ_LAYOUT_MIXEDC>(x, x_sum, qweight, rowmeta, y, N, K, residual, stream)
                   : staged_layout<MXQ_LAYOUT_MIXED>(x, x_sum, qweight, rowmeta, y, N, K, residual, stream);
}

#ifdef MXQ_PROFILING
// tools/gemv_stamps.py: point the kernels of THIS library at a device buffer of 4 stamps per workgroup (NULL: off),
// and a fused one-token launch (prologue 0 / 1 / 2, explicit workgroup size or 0) through this library's kernels
extern "C" int mxq_prof_gemv_set_stamps(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemv_stamps), &p, sizeof(p));
}
extern "C" int mxq_prof_gemv_fused_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K,
                                       int prologue, const void* norm_w, float eps, const void* residual, int threads,
                                       void* stream) {
    return fused_layout<MXQ_LAYOUT_MIXED>(x, qweight, rowmeta, y, N, K, prologue, norm_w, eps, residual,
                                          (hipStream_t)stream, threads);
}
// A/B entry for tools/ (correct results): explicit workgroup size (128 / 256 / 512 threads), M = 1
extern "C" int mxq_prof_gemv_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                                 int threads, void* stream) {
    if (M != 1) return -1;
    return launch<1>(x, qweight, rowmeta, y, M, N, K, (hipStream_t)stream, threads);
}
#endif
