// EXPERIMENT (round 3): the dependent GEMVs of a decoder layer -- o_proj (+ residual) -> RMSNorm + gate|up -> SwiGLU + down
// (+ residual) [-> RMSNorm + the next layer's q|k|v] -- as ONE PERSISTENT launch: one workgroup per CU (15 working waves
// + 1 waiting wave) runs all ops; every wave keeps two weight tiles in flight ACROSS the op boundaries (the first two
// tiles of op i + 1 are issued right behind the K loop of op i, before anything waits), and the ops' dependency is a set
// of device counters (16 words on 16 lines per edge, monotonic: target = (launch generation + 1) x publishers per word,
// so nothing is ever reset).  What tools/experiments/gemv_chain showed not to work -- the next op's WORKGROUPS resident
// early -- is replaced by the same waves serving every op.
// Self-contained; built by build.sh into abtmp/lib_decode_engine.so; probe: engine_probe.py.
//
// RESULT (gpurun_out/r3c59; first build): correct (<= 6e-4 of the output scale against the separate launches, identical
// from run to run, error word clear, generation word counting) and SLOWER -- 53.3 us per layer for the four ops against
// 43.6 us as four launches (compact metadata 51.3 / 41.4); three ops 41.6.  Reading: a dependency edge between ops is a
// chain of global round trips whoever implements it -- rows written through and acknowledged (~1.5 us), counter add
// (~1), poll round trip (~1.2), activation read past L2 (~2) -- about as long as the ~5 us a launch boundary costs, and
// two register tiles per wave (69 KB per CU) prefetched across the edge do not cover it; behind the edge every wave is
// back to its latency-bound two-tiles-in-flight stream.  What could beat the launches is a loader that keeps an LDS ring
// of ~100 KB per CU full across the edges (the guide's weight-streaming engine measures 0.87-0.89x for such a layer);
// a {data, tag} granule hand-off would take ~2 us off each edge.  Not pursued further this round.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mxq_dequant.h"
#include "mxq_format.h"
#include "mxq_gemv_common.h"

extern "C" {
typedef struct {
    const void* x;
    const void* qweight;
    const void* rowmeta;
    void* y;
    const void* norm_w;
    const void* residual;
    int N, K, prologue;
    float eps;
} eng_op_t;
}

namespace {

constexpr int WAVES = 15, WORK_THREADS = WAVES * 64, THREADS = WORK_THREADS + 64;   // 15 working waves + the waiting wave = 1024 threads
constexpr int MAX_OPS = 4, NJ_MAX = 8;
constexpr int SLOTS = 16, LINE = 32;                 // counters: [edge][slot] on separate 128-byte lines
constexpr int WS_GEN = MAX_OPS * SLOTS * LINE, WS_ERR = WS_GEN + LINE, WS_INTS = WS_ERR + LINE;
constexpr int POLL_BUDGET = 1 << 18;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct Args {
    eng_op_t op[MAX_OPS];
    int n;
    int* ws;
};

__device__ __forceinline__ h8 ld16_agent(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));
}

template <bool COMPACT>
struct Tile {
    uint32_t c2w[3], z2w[3], c4w[2], scw;
    uint2 qq[3];
};

// the weight of an op behind ONE descriptor; tile t of this CU's list = (row block cu + (t / NC4) * G, chunk quad t % NC4)
struct Wsrc {
    __amdgpu_buffer_rsrc_t rs;
    int NC4, NC, NRB, T;      // T = tiles of this CU = its row blocks x NC4
    uint32_t rb_bytes;        // bytes of one row block
};
template <bool COMPACT>
__device__ __forceinline__ Wsrc wsrc_of(const eng_op_t& o, int cu, int G) {
    constexpr int BLK_DW = COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    Wsrc w;
    w.NC = o.K / 64;
    w.NC4 = (w.NC + 3) / 4;
    w.NRB = o.N / 16;
    w.rb_bytes = (uint32_t)w.NC * BLK_DW * 4;
    const int nj = w.NRB > cu ? (w.NRB - cu + G - 1) / G : 0;
    w.T = nj * w.NC4;
    w.rs = __builtin_amdgcn_make_buffer_rsrc((void*)o.qweight, 0, (uint32_t)w.NRB * w.rb_bytes, 0x00020000);
    return w;
}
template <bool COMPACT>
__device__ __forceinline__ Tile<COMPACT> load_tile(const Wsrc& w, int t, int cu, int G, int lane_off, int r) {
    constexpr int BLK_DW = COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    Tile<COMPACT> tl = {};
    const int j = t / w.NC4, c4 = t - j * w.NC4;
    // a tile index past the CU's list: an offset beyond the buffer -- zeros, no traffic
    const uint32_t so = t < w.T ? (uint32_t)(cu + j * G) * w.rb_bytes + (uint32_t)c4 * (4 * BLK_DW * 4) : 0xC0000000u;
    auto dw = [&](int idx) { return __builtin_amdgcn_raw_buffer_load_b32(w.rs, lane_off + idx * 4, so, 0); };
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        tl.c2w[g] = dw(mxq_c2(g, r));
        if constexpr (COMPACT) tl.z2w[g] = __builtin_amdgcn_raw_buffer_load_b16(w.rs, lane_off + mxqc_z2_u16(g, r) * 2, so, 0);
        else tl.z2w[g] = dw(mxq_z2(g, r));
        const int q = COMPACT ? mxqc_qq(g) : mxq_qq(g);
        tl.qq[g] = make_uint2(dw(q), dw(q + 1));
    }
    tl.c4w[0] = dw(mxq_c4(0, r));
    tl.c4w[1] = dw(mxq_c4(1, r));
    tl.scw = __builtin_amdgcn_raw_buffer_load_b16(w.rs, lane_off + (COMPACT ? mxqc_sc_u16(r) : mxq_sc_u16(r)) * 2, so, 0);
    return tl;
}

// One op on the working waves.  T0 / T1: this op's first two tiles of the wave, already in flight; on return they are the
// NEXT op's (`wn`; an op with T = 0 when there is none).
template <int PRO, bool COMPACT>
__device__ __forceinline__ void phase(const eng_op_t& o, const Wsrc& wc, const Wsrc& wn, Tile<COMPACT>& T0, Tile<COMPACT>& T1,
                                      char* smem, int cu, int G) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, cs = lane >> 4;
    constexpr int BLK_DW = COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    const int lane_off = (cs * BLK_DW) * 4;
    const int K = o.K, NG = K / 16, NC = wc.NC, NC4 = wc.NC4;
    float* xsum = (float*)(smem + (size_t)K * 2);
    float* red = xsum + NG;                             // [3][NJ_MAX][WAVES][16]
    float* wsum = red + 3 * NJ_MAX * WAVES * 16;
    char* dummy = (char*)(wsum + WAVES);

    __syncthreads();       // B1: the producer's output is there (the waiting wave has seen the counters)

    // ---- activations: one 16-element group per thread (K <= 16384)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)o.x, 0, (PRO == 2 ? 2 : 1) * K * 2, 0x00020000);
    {
        const int g = min(tid, NG - 1);
        const bool live = tid < NG;
        h8 a0 = ld16_agent(xrs, g * 32), a1 = ld16_agent(xrs, g * 32 + 16), b0, b1;
        if constexpr (PRO == 1) {
            b0 = *(const h8*)((const uint16_t*)o.norm_w + g * 16);
            b1 = *(const h8*)((const uint16_t*)o.norm_w + g * 16 + 8);
        }
        if constexpr (PRO == 2) {
            b0 = ld16_agent(xrs, K * 2 + g * 32);
            b1 = ld16_agent(xrs, K * 2 + g * 32 + 16);
        }
        // zero the partial-sum arrays while the loads fly
        for (int i = tid; i < 3 * NJ_MAX * WAVES * 16; i += WORK_THREADS) red[i] = 0.f;
        float ss = 0.f;
        if constexpr (PRO == 1) {
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += (float)a0[j] * (float)a0[j] + (float)a1[j] * (float)a1[j];
            ss = live ? sq : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0[j] = a0[j] * b0[j];
                a1[j] = a1[j] * b1[j];
            }
#pragma unroll
            for (int of = 32; of >= 1; of >>= 1) ss += __shfl_xor(ss, of, 64);
            if (lane == 0) wsum[wave] = ss;
        }
        if constexpr (PRO == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g0 = (float)a0[j], g1 = (float)a1[j];
                a0[j] = (_Float16)(g0 / (1.0f + __expf(-g0))) * b0[j];
                a1[j] = (_Float16)(g1 / (1.0f + __expf(-g1))) * b1[j];
            }
        }
        float sum = 0.f;
        const uint4 o0 = stage8(__builtin_bit_cast(uint4, a0), sum);
        const uint4 o1 = stage8(__builtin_bit_cast(uint4, a1), sum);
        char* dst = live ? smem + (size_t)tid * 32 : dummy + lane * 32;
        float* sdst = live ? xsum + tid : (float*)(dummy + 64 * 32) + lane;
        *(uint4*)dst = o0;
        *(uint4*)(dst + 16) = o1;
        *sdst = sum;
    }
    __syncthreads();       // B2: activations staged

    // ---- the K loop over this wave's tiles t = wave, wave + 16, ...  (two in flight while a third is computed)
    float P = 0.f, Q = 0.f, R = 0.f, X4 = 0.f;
    int jcur = wave / NC4;
    auto flush = [&](int j) {          // the wave's partial sums of row block j -> LDS (4 chunk slots folded first)
        float pq = P - Q, rr = R, x4 = X4;
        pq += __shfl_xor(pq, 16, 64); pq += __shfl_xor(pq, 32, 64);
        rr += __shfl_xor(rr, 16, 64); rr += __shfl_xor(rr, 32, 64);
        x4 += __shfl_xor(x4, 16, 64); x4 += __shfl_xor(x4, 32, 64);
        if (cs == 0) {
            float* d = red + (j * WAVES + wave) * 16 + r;
            d[0] = pq;
            d[NJ_MAX * WAVES * 16] = rr;
            d[2 * NJ_MAX * WAVES * 16] = x4;
        }
        P = Q = R = X4 = 0.f;
    };
    auto step = [&](int t, const Tile<COMPACT>& tl) {
        if (t < wc.T) {                                 // (no global-memory ops inside: the wait counts stay exact)
            const int j = t / NC4, c4 = t - j * NC4;
            if (j != jcur) {
                flush(jcur);
                jcur = j;
            }
            const int chunk = c4 * 4 + cs;
            if (chunk < NC) {
                const char* xk = smem + (size_t)chunk * 128;
                const float* xg = xsum + chunk * 4;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float s = mxq_scale(__uint_as_float(tl.qq[g].x), __uint_as_float(tl.qq[g].y), (tl.scw >> (4 * g)) & 15u);
                    float z;
                    if constexpr (COMPACT) z = (float)__builtin_bit_cast(_Float16, (uint16_t)tl.z2w[g]);
                    else z = __uint_as_float(tl.z2w[g]);
                    const float sz = s * __builtin_fmaf(z, 0.25f, 1.0f);
                    const uint4 xa = *(const uint4*)(xk + g * 32);
                    const uint4 xb = *(const uint4*)(xk + g * 32 + 16);
                    P = __builtin_fmaf(s, codedot2x16(tl.c2w[g], xa, xb, 0.f), P);
                    Q = __builtin_fmaf(sz, xg[g], Q);
                }
                const uint4 xa = *(const uint4*)(xk + 3 * 32);
                const uint4 xb = *(const uint4*)(xk + 3 * 32 + 16);
                R = codedot4x8(tl.c4w[0], xa, R);
                R = codedot4x8(tl.c4w[1], xb, R);
                X4 += xg[3];
            }
        }
    };
    Tile<COMPACT> T2;
    for (int t = wave; t < wc.T; t += 3 * WAVES) {
        T2 = load_tile<COMPACT>(wc, t + 2 * WAVES, cu, G, lane_off, r);
        step(t, T0);
        T0 = load_tile<COMPACT>(wc, t + 3 * WAVES, cu, G, lane_off, r);
        step(t + WAVES, T1);
        T1 = load_tile<COMPACT>(wc, t + 4 * WAVES, cu, G, lane_off, r);
        step(t + 2 * WAVES, T2);
    }
    // ---- the NEXT op's first two tiles go out now, before anything of this op's tail
    __builtin_amdgcn_sched_barrier(0);
    T0 = load_tile<COMPACT>(wn, wave, cu, G, lane_off, r);
    __builtin_amdgcn_sched_barrier(0);
    T1 = load_tile<COMPACT>(wn, wave + WAVES, cu, G, lane_off, r);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < wc.T) flush(jcur);
    __syncthreads();       // B3: partial sums in LDS
    __syncthreads();       // B4: the waiting wave has read the partial sums (and written the rows)
}

// The op's epilogue, on the WAITING wave (it has no loads in flight, so its s_waitcnt vmcnt(0) waits for nothing but its own
// stores): per row the 15 waves' partial sums in order, scale / zero-point of the 4-bit arm, RMSNorm scale, residual, the
// write-through store of y, and the op's counter.
__device__ __forceinline__ void epilogue(const eng_op_t& o, const Wsrc& wc, char* smem, int cu, int G, int lane, int edge_out,
                                         int* __restrict__ ws) {
    const int K = o.K, NG = K / 16;
    const float* xsum = (const float*)(smem + (size_t)K * 2);
    const float* red = xsum + NG;
    const float* wsum = red + 3 * NJ_MAX * WAVES * 16;
    const int nj = wc.T / wc.NC4;
    for (int i = lane; i < nj * 16; i += 64) {
        const int j = i >> 4, rr = i & 15;
        const int rb = cu + j * G;
        float pq = 0.f, rs = 0.f, x4 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float* d = red + (j * WAVES + w) * 16 + rr;
            pq += d[0];
            rs += d[NJ_MAX * WAVES * 16];
            x4 += d[2 * NJ_MAX * WAVES * 16];
        }
        const float4 rm = ((const float4*)o.rowmeta)[rb * 16 + rr];
        const float s4 = mxq_scale(rm.z, rm.w, (uint32_t)rm.y), z4 = rm.x;
        float v = 4.0f * pq + 16.0f * s4 * (rs - __builtin_fmaf(z4, 0.0625f, 1.0f) * x4);
        if (o.prologue == 1) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) tot += wsum[w];
            v *= rsqrtf(tot / (float)K + o.eps);
        }
        const int row = rb * 16 + rr;
        _Float16 h = (_Float16)v;
        if (o.residual) {
            const uint16_t rv = __hip_atomic_load((const uint16_t*)o.residual + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h = __builtin_bit_cast(_Float16, rv) + h;
        }
        __hip_atomic_store((uint16_t*)o.y + row, __builtin_bit_cast(uint16_t, h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the rows have reached the coherence point
    if (edge_out >= 0 && lane == 0)
        __hip_atomic_fetch_add(ws + (edge_out * SLOTS + (cu & (SLOTS - 1))) * LINE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// fixed op pattern of a Llama decoder layer: o_proj (prologue 0), gate|up (1), down (2) [, next q|k|v (1)]
template <bool COMPACT>
__global__ __launch_bounds__(THREADS, 4) void decode_engine_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cu = blockIdx.x, G = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* ws = a.ws;
    const int gen = __hip_atomic_load(ws + WS_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // launches completed so far

    eng_op_t none = a.op[0];
    none.N = 0;
    const Wsrc w0 = wsrc_of<COMPACT>(a.op[0], cu, G), w1 = wsrc_of<COMPACT>(a.op[1], cu, G), w2 = wsrc_of<COMPACT>(a.op[2], cu, G);
    const Wsrc w3 = wsrc_of<COMPACT>(a.n > 3 ? a.op[3] : none, cu, G), wend = wsrc_of<COMPACT>(none, cu, G);
    if (wave == WAVES) {
        // ---- the waiting wave: before every op but the first, wait until all G workgroups have published the op before;
        // behind the op's K loop, its epilogue.  Every launch publishes edges 0..2 (the counters run in step with the
        // generation word), also when there is no op 3.
        const int slot = lane & (SLOTS - 1);
        const int want = (gen + 1) * ((G - slot + SLOTS - 1) / SLOTS);
        for (int i = 0; i < a.n; ++i) {
            if (i > 0) {
                const int* word = ws + ((i - 1) * SLOTS + slot) * LINE;
                int n = 0;
                while (!__all(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want >= 0)) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++n > POLL_BUDGET) {
                        if (lane == 0) __hip_atomic_store(ws + WS_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            __syncthreads();   // B1
            __syncthreads();   // B2
            __syncthreads();   // B3
            epilogue(a.op[i], i == 0 ? w0 : i == 1 ? w1 : i == 2 ? w2 : w3, smem, cu, G, lane, i < 3 ? i : -1, ws);
            __syncthreads();   // B4
        }
        if (cu == 0 && lane == 0) __hip_atomic_fetch_add(ws + WS_GEN, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int r = lane & 15, cs = lane >> 4;
    constexpr int BLK_DW = COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    const int lane_off = (cs * BLK_DW) * 4;
    Tile<COMPACT> T0 = load_tile<COMPACT>(w0, wave, cu, G, lane_off, r);
    __builtin_amdgcn_sched_barrier(0);
    Tile<COMPACT> T1 = load_tile<COMPACT>(w0, wave + WAVES, cu, G, lane_off, r);
    __builtin_amdgcn_sched_barrier(0);
    // every launch publishes edges 0..2 (the counters run in step with the generation word), also when there is no op 3
    phase<0, COMPACT>(a.op[0], w0, w1, T0, T1, smem, cu, G);
    phase<1, COMPACT>(a.op[1], w1, w2, T0, T1, smem, cu, G);
    phase<2, COMPACT>(a.op[2], w2, w3, T0, T1, smem, cu, G);
    if (a.n > 3) phase<1, COMPACT>(a.op[3], w3, wend, T0, T1, smem, cu, G);
}

}   // namespace

extern "C" size_t mxq_exp_decode_engine_ws_bytes(void) { return WS_INTS * sizeof(int); }

// ops: host array, n = 3 or 4 with prologues (0, 1, 2 [, 1]); ws: device, zeroed ONCE at allocation
extern "C" int mxq_exp_decode_engine_f16(const eng_op_t* ops, int n, int compact, void* ws, void* stream) {
    if (n != 3 && n != 4) return -1;
    static const int pro[4] = {0, 1, 2, 1};
    Args a = {};
    size_t smem = 0;
    for (int i = 0; i < n; ++i) {
        if (ops[i].prologue != pro[i] || ops[i].K % 64 || ops[i].N % 16 || ops[i].K > 16384) return -1;
        a.op[i] = ops[i];
        const size_t need = (size_t)ops[i].K * 2 + (size_t)(ops[i].K / 16) * 4 + 3 * NJ_MAX * WAVES * 16 * 4 + WAVES * 4 + 64 * 32 + 64 * 4;
        smem = need > smem ? need : smem;
    }
    a.n = n;
    a.ws = (int*)ws;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 16)
        return -2;
    for (int i = 0; i < n; ++i)
        if (ops[i].N / 16 > NJ_MAX * cus) return -1;
    hipError_t e;
    if (compact) e = hipFuncSetAttribute((const void*)decode_engine_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    else e = hipFuncSetAttribute((const void*)decode_engine_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    if (compact) decode_engine_kernel<true><<<cus, THREADS, smem, (hipStream_t)stream>>>(a);
    else decode_engine_kernel<false><<<cus, THREADS, smem, (hipStream_t)stream>>>(a);
    return (int)hipGetLastError();
}
