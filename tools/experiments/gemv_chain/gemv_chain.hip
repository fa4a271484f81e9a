// EXPERIMENT RECORD (round 3) -- not built, not part of the product.  It was csrc/gemv_chain.hip with a C entry
// mxq_gemv_chain_f16 (mxq_chain_abi.h, host_side.py.txt beside this file hold the ABI block, the ctypes / Python side and
// the parity test that were removed with it); probes: decode_chain_probe.py, decode_concat_probe.py, decode_overlap_probe.py.
//
// RESULT.  Correct: the chain o_proj -> gate|up -> down -> next q|k|v as ONE launch gave the separate launches' bits
// (6 parity cases, exact + compact metadata, Llama and small sizes), never hung, never tripped its poll budget.
// And SLOWER: 52-53 us per layer against 43-44 us for the four launches (hipGraph replay, 8 distinct layers); with the
// waits compiled out (timing only: all four ops free-running in one grid) 44 us -- no gain even then.  Why:
//   * the launch boundary that tools/.../decode_concat_probe.py prices at ~5 us (three K = 4096 GEMVs as three launches
//     33.4 us, as ONE gemv.hip launch over their row-concatenation 23.0 us) is not launch overhead: it is the
//     latency-bound part of each GEMV (o_proj: 5.6 us for 9.5 MB) hiding under ANOTHER op's streaming.  Dependent ops
//     cannot do that; what a chain can hide is only the first weight round trip, and it pays for it with the flag's
//     poll latency and an activation read that must bypass L2 (sc1: the consumer may sit on another XCD).
//   * this GEMV saturates HBM by wave count (every wave of an op resident, two tiles in flight each).  A chain needs the
//     NEXT op's workgroups resident as well while the current one runs: o_proj + gate|up alone are 5400 waves against
//     ~4600 slots at 96 VGPRs, so the tail of gate|up starts late and takes a full per-wave streaming time (19.5 us
//     for gate|up inside the chain against 14.9 alone).
// Pitfalls met on the way (all in the file's history of this round): a single counter word serialises hundreds of
// producer adds (148 us; 16 words on 16 lines fix that); a consumer-side reset puts a returning atomic on every
// consumer's critical path (the caller zeroes the words instead); ANY branch around the working waves' loads turns the K
// loop's counted waits into vmcnt(0) (hence the dedicated waiting wave); a buffer descriptor built from a per-lane
// pointer turns every load into a 64-trip waterfall loop; the RMSNorm sum of squares has to be added in gemv.hip's
// thread order to give the same bits.
// What would work instead (not built): a persistent kernel per decoder layer whose loader waves stream the NEXT op's
// whole weight slice into LDS (o_proj 37 KB, down 100 KB, q|k|v 111 KB per CU fit; gate|up 199 KB does not quite) while
// the current op computes, so that after the dependency only arithmetic on LDS-resident weights remains.
// ------------------------------------------------------------------------------------------------
// Decode (one token): a CHAIN of fused GEMVs in ONE launch -- op i + 1 consumes op i's output vector.
//
// Why: a decoder layer is o_proj -> (RMSNorm) gate|up -> (SwiGLU) down -> (RMSNorm) next q|k|v, each a GEMV that
// streams 10-50 MB of packed weights in 5-15 us, and each launch boundary costs ~5 us of that (drain, launch, and the
// first loaded-memory round trip of the next kernel: tools/decode_concat_probe.py -- three launches 33.4 us, the same
// rows as ONE launch 23.0 us).  The weights do not depend on the previous op; only the 8-22 KB activation vector does.
//
// How: the grid is the concatenation of the ops' workgroups.  A workgroup of op i + 1 FIRST issues its weight-tile
// loads, THEN waits until a counter says that every workgroup of op i has published its rows, and only then loads and
// stages the activations.  No grid barrier, no persistent workgroups, no co-residency assumption: workgroups are
// dispatched in index order (per XCD), so whenever a workgroup occupies a slot every lower-indexed workgroup of its
// XCD has been dispatched already, and a workgroup only ever waits for LOWER-indexed ones -- the lowest unfinished
// workgroup of the grid never waits for anything unfinished, so the chain always makes progress; waiting workgroups
// hold their slots only while in-flight weight loads are what they would be waiting for anyway.
// Publication: y is written with agent-scope (write-through) stores, s_waitcnt vmcnt(0), then ONE relaxed agent-scope
// add on the op's counter; consumers poll it with relaxed agent-scope loads (one lane per workgroup, s_sleep between
// polls, bounded: a poll budget that runs out raises an error word instead of hanging the device) and read the
// activations / the residual with agent-scope loads.  The counters are zeroed by the caller before every launch (a
// consumer-side reset would put a returning atomic on every consumer's critical path).
//
// Arithmetic = gemv.hip's kernel with the same waves-per-row-block as its dispatch picks (8 / 4 / 2 by row count):
// same K split, same reduction order, bit-identical results (tests/test_gpu_parity.py::test_gemv_chain_*).
// Reference: the fused unpack + dot of gemv_mxq_cuda.cu:39-208, chained the way a decoder layer chains its Linears
// (LLM-QAT/models/modeling_llama_quant.py:262-291, 323-360).
#include <hip/hip_runtime.h>

#include "mxq_hip.h"

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_gemv_common.h"
#include "mxq_kernels.h"

namespace {

constexpr int CH_WAVES = 8;                        // working waves; one more wave per workgroup only waits for the producer
constexpr int CH_WORK_THREADS = CH_WAVES * 64, CH_THREADS = CH_WORK_THREADS + 64;
constexpr int CH_MAX_OPS = MXQ_CHAIN_MAX_OPS;
// sync workspace (ints): edge e's producer counters = CH_SLOTS words, one per 128-byte line, at (e * CH_SLOTS + s) * CH_LINE;
// the error word behind them.  The CALLER zeroes the counters before every launch (mxq_hip.h).
constexpr int CH_SLOTS = 16, CH_LINE = 32;
#ifndef CHAIN_SLEEP
#define CHAIN_SLEEP 16
#endif
#ifndef CHAIN_LD_AUX
#define CHAIN_LD_AUX 16
#endif
#ifndef CHAIN_PLAIN_ST
#define CHAIN_PLAIN_ST 0
#endif
#ifndef CHAIN_NOWAIT
#define CHAIN_NOWAIT 0
#endif
constexpr int WS_ERR = (CH_MAX_OPS - 1) * CH_SLOTS * CH_LINE;
constexpr int WS_INTS = WS_ERR + CH_LINE;
constexpr int POLL_BUDGET = 1 << 18;                  // x s_sleep 16 (~1 us): a quarter of a second

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ChainArgs {
    mxq_chain_op_t op[CH_MAX_OPS];
    int wg_begin[CH_MAX_OPS + 1];   // first workgroup of op i; wg_begin[n] = grid
    int n;
    int* ws;
};

// 16 B at byte offset `off` of the (wave-uniform) descriptor, coherent with other CUs' write-through stores.  (The
// descriptor must be uniform: built from a per-lane pointer, every load becomes a 64-trip waterfall loop.)
__device__ __forceinline__ h8 ld16_agent(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, CHAIN_LD_AUX));
}

// One workgroup of one op.  PRO as in gemv.hip (0 none, 1 RMSNorm, 2 SwiGLU); WPR = waves per 16-row block (the
// workgroup covers 8 / WPR row blocks); dep: the op consumes the previous op's output of this launch (edge `edge`).
template <int PRO, int WPR, bool COMPACT>
__device__ __forceinline__ void chain_wg(const mxq_chain_op_t& o, int lwg, int* __restrict__ ws, bool dep, int edge, int n_prod,
                                         int n_cons, int done_edge) {
    constexpr int RBW = CH_WAVES / WPR;                 // row blocks per workgroup
    constexpr int NG2 = 3, BLK_DW = COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, cs = lane >> 4;
    const int N = o.N, K = o.K;
    const int NRB = N / 16, NC = K / 64, NC4 = (NC + 3) / 4, NG = K / 16;
    const int rb = lwg * RBW + wave / WPR, wsub = wave % WPR;
    const bool has_rb = rb < NRB;                       // wave-uniform
    const uint16_t* x = (const uint16_t*)o.x;
    const uint16_t* norm_w = (const uint16_t*)o.norm_w;
    const uint16_t* residual = (const uint16_t*)o.residual;
    uint16_t* y = (uint16_t*)o.y;

    float* xsum = (float*)(smem + (size_t)K * 2);
    float* xsq = xsum + NG;                             // PRO 1: the groups' sums of squares (summed in gemv.hip's order below)
    float* red = xsq + NG;
    float* wsum = red + CH_WAVES * 16;
    char* dummy = (char*)(wsum + CH_WAVES);

    struct Tile {
        uint32_t c2w[NG2], z2w[NG2], c4w[2], scw;
        uint2 qq[NG2];
    };
    // a row block beyond N (a workgroup's surplus waves): an empty descriptor -- zeros, no traffic
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const uint32_t*)o.qweight + (int64_t)(has_rb ? rb : 0) * NC * BLK_DW), 0, has_rb ? NC * BLK_DW * 4 : 0, 0x00020000);
    const int lane_off = (cs * BLK_DW) * 4;
    auto load_tile = [&](int c4) {
        Tile t = {};
        const int so = c4 * (4 * BLK_DW * 4);
        auto dw = [&](int idx) { return __builtin_amdgcn_raw_buffer_load_b32(rs, lane_off + idx * 4, so, 0); };
#pragma unroll
        for (int g = 0; g < NG2; ++g) {
            t.c2w[g] = dw(mxq_c2(g, r));
            if constexpr (COMPACT) t.z2w[g] = __builtin_amdgcn_raw_buffer_load_b16(rs, lane_off + mxqc_z2_u16(g, r) * 2, so, 0);
            else t.z2w[g] = dw(mxq_z2(g, r));
            const int q = COMPACT ? mxqc_qq(g) : mxq_qq(g);
            t.qq[g] = make_uint2(dw(q), dw(q + 1));
        }
        t.c4w[0] = dw(mxq_c4(0, r));
        t.c4w[1] = dw(mxq_c4(1, r));
        t.scw = __builtin_amdgcn_raw_buffer_load_b16(rs, lane_off + (COMPACT ? mxqc_sc_u16(r) : mxq_sc_u16(r)) * 2, so, 0);
        return t;
    };
    const float4 rm = ((const float4*)o.rowmeta)[(has_rb ? rb : 0) * 16 + r];
    __builtin_amdgcn_sched_barrier(0);
    // ---- the WAITING WAVE (wave 8): polls the producer's counters and takes part in the barriers, nothing else.  It is a
    // wave of its own because (a) a poll's result returns in order behind the wave's older loads, and the working waves
    // have two weight tiles in flight by now; (b) any branch around the working waves' loads would make the compiler's
    // wait counts conservative (vmcnt(0) in the K loop: the first version of this kernel ran 2.5x slower for it).
    // This path never rejoins the working waves' code.
    if (wave == CH_WAVES) {
        if (dep && !CHAIN_NOWAIT) {
            // the edge's producers count themselves on CH_SLOTS words, each on its own 128-byte line (hundreds of adds on
            // ONE word serialise); lane s polls word s
            const int slot = lane & (CH_SLOTS - 1);
            const int want = (n_prod - slot + CH_SLOTS - 1) / CH_SLOTS;          // producers with index % CH_SLOTS == slot
            const int* word = ws + (edge * CH_SLOTS + slot) * CH_LINE;
            int n = 0;
            while (!__all(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want)) {
                __builtin_amdgcn_s_sleep(CHAIN_SLEEP);
                if (++n > POLL_BUDGET) {       // never hang the device: flag it and carry on with whatever is there
                    if (lane == 0) __hip_atomic_store(ws + WS_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();   // 1: the producer's output is there
        __syncthreads();   // 2: activations staged
        __syncthreads();   // 3: partial sums in LDS
        __syncthreads();   // 4: rows published
        return;
    }
    // ---- working waves: the first two weight tiles BEFORE the wait for the producer (they do not depend on it)
    Tile T0 = load_tile(wsub);
    __builtin_amdgcn_sched_barrier(0);
    Tile T1 = load_tile(wsub + WPR), T2;
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();       // 1

    // ---- activations (agent-scope loads when another workgroup of this launch wrote them)
    struct Act {
        h8 a0, a1, b0, b1;
    };
    // (op 0's input comes from an earlier launch: any load would do there)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (PRO == 2 ? 2 : 1) * K * 2, 0x00020000);
    auto load_act = [&](int i) {
        Act t;
        const int g = min(i, NG - 1);
        t.a0 = ld16_agent(xrs, g * 32);
        t.a1 = ld16_agent(xrs, g * 32 + 16);
        if constexpr (PRO == 1) {
            t.b0 = *(const h8*)(norm_w + g * 16);
            t.b1 = *(const h8*)(norm_w + g * 16 + 8);
        }
        if constexpr (PRO == 2) {
            t.b0 = ld16_agent(xrs, K * 2 + g * 32);
            t.b1 = ld16_agent(xrs, K * 2 + g * 32 + 16);
        }
        return t;
    };
    const Act act0 = load_act(tid), act1 = load_act(tid + CH_WORK_THREADS);
    float ss = 0.f;
    auto stage_act = [&](int i, Act t) {
        const bool live = i < NG;
        if constexpr (PRO == 1) {
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += (float)t.a0[j] * (float)t.a0[j] + (float)t.a1[j] * (float)t.a1[j];
            *(live ? xsq + i : (float*)(dummy + 64 * 32) + 64 + lane) = sq;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                t.a0[j] = t.a0[j] * t.b0[j];
                t.a1[j] = t.a1[j] * t.b1[j];
            }
        }
        if constexpr (PRO == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g0 = (float)t.a0[j], g1 = (float)t.a1[j];
                t.a0[j] = (_Float16)(g0 / (1.0f + __expf(-g0))) * t.b0[j];
                t.a1[j] = (_Float16)(g1 / (1.0f + __expf(-g1))) * t.b1[j];
            }
        }
        float sum = 0.f;
        const uint4 o0 = stage8(__builtin_bit_cast(uint4, t.a0), sum);
        const uint4 o1 = stage8(__builtin_bit_cast(uint4, t.a1), sum);
        char* dst = live ? smem + (size_t)i * 32 : dummy + lane * 32;
        float* sdst = live ? xsum + i : (float*)(dummy + 64 * 32) + lane;
        *(uint4*)dst = o0;
        *(uint4*)(dst + 16) = o1;
        *sdst = sum;
    };
    stage_act(tid, act0);
    stage_act(tid + CH_WORK_THREADS, act1);
    for (int i = tid + 2 * CH_WORK_THREADS; i < NG; i += CH_WORK_THREADS) stage_act(i, load_act(i));
    const float s4 = mxq_scale(rm.z, rm.w, (uint32_t)rm.y), z4 = rm.x;
    __syncthreads();
    if constexpr (PRO == 1) {
        // sum of squares in EXACTLY the order of gemv.hip's kernel at 64 * WPR threads (thread t: groups t, t + T, ...;
        // xor-butterfly over the wave; the waves' sums added in order at the end): the RMSNorm scale is then the same
        // float, and the chain's results the same bits as the separate launches'
        if (wave < WPR) {
            for (int i = tid; i < NG; i += 64 * WPR) ss += xsq[i];
#pragma unroll
            for (int of = 32; of >= 1; of >>= 1) ss += __shfl_xor(ss, of, 64);
            if (lane == 0) wsum[wave] = ss;
        }
    }

    float P = 0.f, Q = 0.f, R = 0.f, X4 = 0.f;
    auto compute = [&](int c4, const Tile& t) {
        const int chunk = c4 * 4 + cs;
        if (chunk < NC) {
            const char* xk = smem + (size_t)chunk * 128;
            const float* xg = xsum + chunk * 4;
#pragma unroll
            for (int g = 0; g < NG2; ++g) {
                const float s = mxq_scale(__uint_as_float(t.qq[g].x), __uint_as_float(t.qq[g].y), (t.scw >> (4 * g)) & 15u);
                float z;
                if constexpr (COMPACT) z = (float)__builtin_bit_cast(_Float16, (uint16_t)t.z2w[g]);
                else z = __uint_as_float(t.z2w[g]);
                const float sz = s * __builtin_fmaf(z, 0.25f, 1.0f);
                const uint4 xa = *(const uint4*)(xk + g * 32);
                const uint4 xb = *(const uint4*)(xk + g * 32 + 16);
                P = __builtin_fmaf(s, codedot2x16(t.c2w[g], xa, xb, 0.f), P);
                Q = __builtin_fmaf(sz, xg[g], Q);
            }
            const uint4 xa = *(const uint4*)(xk + 3 * 32);
            const uint4 xb = *(const uint4*)(xk + 3 * 32 + 16);
            R = codedot4x8(t.c4w[0], xa, R);
            R = codedot4x8(t.c4w[1], xb, R);
            X4 += xg[3];
        }
    };
    for (int c4 = wsub; c4 < NC4; c4 += 3 * WPR) {
        T2 = load_tile(c4 + 2 * WPR);
        compute(c4, T0);
        T0 = load_tile(c4 + 3 * WPR);
        compute(c4 + WPR, T1);
        T1 = load_tile(c4 + 4 * WPR);
        compute(c4 + 2 * WPR, T2);
    }
    float v = 4.0f * (P - Q) + 16.0f * s4 * (R - __builtin_fmaf(z4, 0.0625f, 1.0f) * X4);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (cs == 0) red[wave * 16 + r] = v;
    __syncthreads();
    if (tid < RBW * 16) {
        const int rbl = tid >> 4, rr = tid & 15;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WPR; ++w) a += red[(rbl * WPR + w) * 16 + rr];
        if constexpr (PRO == 1) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WPR; ++w) tot += wsum[w];
            a *= rsqrtf(tot / (float)K + o.eps);
        }
        const int row = (lwg * RBW + rbl) * 16 + rr;
        _Float16 h = (_Float16)a;
        if (row < N) {
            if (residual) {
                const uint16_t rv = __hip_atomic_load(residual + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                h = __builtin_bit_cast(_Float16, rv) + h;
            }
            // agent-scope (write-through) store: a consumer workgroup of this launch may sit on another XCD
            if (CHAIN_PLAIN_ST) y[row] = __builtin_bit_cast(uint16_t, h);
            else __hip_atomic_store(y + row, __builtin_bit_cast(uint16_t, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's y stores have reached the coherence point
    __syncthreads();       // 4
    if (done_edge >= 0) {
        if (tid == 0)
            __hip_atomic_fetch_add(ws + (done_edge * CH_SLOTS + (lwg & (CH_SLOTS - 1))) * CH_LINE, 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
}

__host__ __device__ inline int chain_wpr(int N) {       // gemv.hip's gemv_threads(): waves per row block by row-block count
    const int nrb = N / 16;
    return nrb <= 384 ? 8 : nrb <= 768 ? 4 : 2;
}

template <bool COMPACT>
__global__ __launch_bounds__(CH_THREADS, 5) void mxq_gemv_chain_kernel(const ChainArgs a) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < CH_MAX_OPS; ++k)
        if (k < a.n && (int)blockIdx.x >= a.wg_begin[k]) i = k;
    i = __builtin_amdgcn_readfirstlane(i);
    const mxq_chain_op_t& o = a.op[i];
    const int lwg = blockIdx.x - a.wg_begin[i];
    const bool dep = i > 0;
    const int edge = i - 1, n_prod = dep ? a.wg_begin[i] - a.wg_begin[i - 1] : 0, n_cons = a.wg_begin[i + 1] - a.wg_begin[i];
    const int done = i + 1 < a.n ? i : -1;
    const int wpr = chain_wpr(o.N);
#define CHAIN_CASE(PRO, WPR)                                                                \
    if (o.prologue == PRO && wpr == WPR) {                                                  \
        chain_wg<PRO, WPR, COMPACT>(o, lwg, a.ws, dep, edge, n_prod, n_cons, done);         \
        return;                                                                             \
    }
    CHAIN_CASE(0, 8) CHAIN_CASE(0, 4) CHAIN_CASE(0, 2)
    CHAIN_CASE(1, 8) CHAIN_CASE(1, 4) CHAIN_CASE(1, 2)
    CHAIN_CASE(2, 8) CHAIN_CASE(2, 4) CHAIN_CASE(2, 2)
#undef CHAIN_CASE
}

}   // namespace

size_t mxq_gemv_chain_ws_bytes(void) { return WS_INTS * sizeof(int); }

// ops: HOST array of n <= MXQ_CHAIN_MAX_OPS descriptors (copied into the kernel arguments); ws: device, >= mxq_gemv_chain_ws_bytes(),
// zeroed once by the caller (the kernel leaves it zeroed)
static int mxq_launch_gemv_chain_f16(const mxq_chain_op_t* ops, int n, int compact, void* ws, hipStream_t stream) {
    if (n < 1 || n > CH_MAX_OPS) return -1;
    ChainArgs a = {};
    size_t smem = 0;
    int wg = 0;
    for (int i = 0; i < n; ++i) {
        a.op[i] = ops[i];
        const int K = ops[i].K, rbw = CH_WAVES / chain_wpr(ops[i].N);
        a.wg_begin[i] = wg;
        wg += (ops[i].N / 16 + rbw - 1) / rbw;
        const size_t need = (size_t)K * 2 + (size_t)(K / 16) * 8 + CH_WAVES * 16 * 4 + CH_WAVES * 4 + 64 * 32 + 2 * 64 * 4;
        smem = need > smem ? need : smem;
    }
    for (int i = n; i <= CH_MAX_OPS; ++i) a.wg_begin[i] = wg;
    a.n = n;
    a.ws = (int*)ws;
    if (smem > 64 * 1024) return -1;
    if (compact) mxq_gemv_chain_kernel<true><<<wg, CH_THREADS, smem, stream>>>(a);
    else mxq_gemv_chain_kernel<false><<<wg, CH_THREADS, smem, stream>>>(a);
    return (int)hipGetLastError();
}

int mxq_launch_gemv_chain_f16_v(const void* ops, int n, int compact, void* ws, hipStream_t stream) {
    return mxq_launch_gemv_chain_f16((const mxq_chain_op_t*)ops, n, compact, ws, stream);
}
