// Decode-harness glue for BASELINE config 3 (not part of the reference's hot path; the
// reference leaves attention / RoPE to HF transformers): one fused kernel per decoder layer
// and token that applies rotary embedding to q/k, appends k/v to the cache and runs
// single-query attention.  One workgroup per head; position is read from device memory so
// the launch is hipGraph-replayable.
#include <hip/hip_runtime.h>
#include <math.h>

#include "mxq_kernels.h"

#include "mxq_gemv_common.h"   // wave_allsum / wave_allmax / quad_allsum: the shuffle butterflies as vector-ALU operations, bit-identical

namespace {

constexpr int HD = 128;        // head dim (Llama-2-7B)
constexpr int ATT_THREADS = 256;

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(float f) {
    const _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}

// qkv: [3 * heads * HD] fp16 (q | k | v); caches: [heads][max_ctx][HD] fp16; cos/sin: [max_ctx][HD/2] f32;
// out: [heads * HD] fp16.  Rotate-half convention, as mxq_amd/llama_decode.py.
//
// The launch is latency-bound (32 workgroups, a few KB each), so what counts is the number of DEPENDENT memory round
// trips.  Round 1's kernel had four (position -> q/k/v + cos/sin -> K rows -> V rows, ~5.6 us per layer); here
// everything that does not need the position is issued with it: q/k/v of the head and, speculatively, the first 64
// K rows (4 lanes per key) and the first 64 V rows (16 key groups x 4) of the cache -- rows at or beyond the position
// are loaded (inside the allocation: clamped to max_ctx) and ignored.  Only cos/sin of the position (an L2-resident
// table) is a second trip, and contexts beyond 64 keys continue with ordinary loads.  The new key / value take part
// from LDS, not through the cache they are appended to.
// ROW (round 5): cos_t / sin_t are the ONE row of the rotary tables for this token's position ([HD/2] each, gathered once
// per token by rope_row_kernel below) instead of the [max_ctx][HD/2] tables: the 128 values are then loaded in round trip 1
// with everything else, and the dependent second trip (position -> table row) is gone from every layer's launch.
// Returns false (before any side effect) when `max_keys` > 0 and the position needs more keys than that: the caller then
// takes the split path.  (The test sits INSIDE, behind round trip 1: a caller that read the position first and branched
// would put a dependent round trip in front of every load of this function -- measured +2.2 us per layer.)
template <bool ROW>
__device__ __forceinline__ bool attn_one_workgroup(float* sm, int h, const uint16_t* __restrict__ qkv,
                                                   uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                   const int64_t* __restrict__ pos_p, const float* __restrict__ cos_t,
                                                   const float* __restrict__ sin_t, uint16_t* __restrict__ out, int heads,
                                                   int max_ctx, int max_keys = 0) {
    float* q_s = sm;                       // q[HD], knew[HD], vnew[HD], scores[max_ctx], red[8], o2[16*HD]
    float* k_s = sm + HD;
    float* v_s = sm + 2 * HD;
    float* sc = sm + 3 * HD;
    float* red = sc + max_ctx;
    float* o2 = red + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hidden = heads * HD;
    uint16_t* kc = k_cache + (int64_t)h * max_ctx * HD;
    uint16_t* vc = v_cache + (int64_t)h * max_ctx * HD;

    // ---- round trip 1: position, this head's q / k / v, speculative K and V rows
    const int64_t pos64 = *pos_p;
    const int d = tid & (HD - 1), d2 = d & (HD / 2 - 1);
    const uint16_t* qh = qkv + h * HD;
    const uint16_t* kh = qkv + hidden + h * HD;
    const uint16_t q1h = qh[d2], q2h = qh[d2 + HD / 2], k1h = kh[d2], k2h = kh[d2 + HD / 2];
    const uint16_t vh = qkv[2 * hidden + h * HD + d];
    const int part = tid & 3, kj = tid >> 2;                   // scores: 4 lanes per key (32 dims each)
    uint4 kspec[4];
    {
        const uint4* kr = (const uint4*)(kc + (int64_t)min(kj, max_ctx - 1) * HD + part * 32);
#pragma unroll
        for (int v = 0; v < 4; ++v) kspec[v] = kr[v];
    }
    float c_row = 0.f, s_row = 0.f;
    if constexpr (ROW) {
        c_row = cos_t[d2];
        s_row = sin_t[d2];
    }
    const int dg = tid & 15, kg = tid >> 4;                    // P.V: 16 key groups x 16 lanes of 8 dims
    uint4 vspec[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vspec[i] = *(const uint4*)(vc + (int64_t)min(kg + 16 * i, max_ctx - 1) * HD + dg * 8);

    // Precondition (include/mxq_hip.h): 0 <= *pos < max_ctx.  The position lives in device memory (graph replay),
    // so the launcher cannot check it: a position outside the cache must neither be written to the cache nor index
    // the LDS score array.  The head's output is poisoned (NaN) instead, which the caller cannot miss downstream.
    if (pos64 < 0 || pos64 >= max_ctx) {   // uniform over the workgroup: taken before any barrier
        if (tid < HD) out[h * HD + tid] = 0x7E00;
        return true;
    }
    const int pos = (int)pos64;
    if (max_keys > 0 && pos + 1 > max_keys) return false;   // (uniform; nothing written yet)

    // ---- round trip 2 (L2-resident table): cos / sin of the position; RoPE on q and k; append k, v at `pos`
    if (tid < HD) {
        const float c = ROW ? c_row : cos_t[(int64_t)pos * (HD / 2) + d2], s = ROW ? s_row : sin_t[(int64_t)pos * (HD / 2) + d2];
        const float q1 = h2f(q1h), q2 = h2f(q2h), k1 = h2f(k1h), k2 = h2f(k2h);
        const float qr = d < HD / 2 ? q1 * c - q2 * s : q2 * c + q1 * s;
        const float kr = d < HD / 2 ? k1 * c - k2 * s : k2 * c + k1 * s;
        q_s[d] = h2f(f2h(qr));                       // the fp16 rounding the torch path applies
        const uint16_t krh = f2h(kr);
        k_s[d] = h2f(krh);
        v_s[d] = h2f(vh);
        kc[(int64_t)pos * HD + d] = krh;
        vc[(int64_t)pos * HD + d] = vh;
    }
    __syncthreads();

    // scores: 4 lanes per key (32 dims each); key j < pos from the cache (the first 64 already in registers), key pos
    // from LDS
    const float scale = rsqrtf((float)HD);
    for (int j = kj; j <= pos; j += ATT_THREADS / 4) {
        float acc = 0.f;
        if (j == pos) {
#pragma unroll
            for (int e = 0; e < 32; ++e) acc += q_s[part * 32 + e] * k_s[part * 32 + e];
        } else {
            uint4 kw[4];
            if (j == kj) {
#pragma unroll
                for (int v = 0; v < 4; ++v) kw[v] = kspec[v];
            } else {
                const uint4* kr = (const uint4*)(kc + (int64_t)j * HD + part * 32);
#pragma unroll
                for (int v = 0; v < 4; ++v) kw[v] = kr[v];
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t ws[4] = {kw[v].x, kw[v].y, kw[v].z, kw[v].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc += q_s[part * 32 + v * 8 + 2 * e] * h2f((uint16_t)(ws[e] & 0xFFFF));
                    acc += q_s[part * 32 + v * 8 + 2 * e + 1] * h2f((uint16_t)(ws[e] >> 16));
                }
            }
        }
        acc = quad_allsum(acc);
        if (part == 0) sc[j] = h2f(f2h(acc * scale));   // fp16 scores as baddbmm produces them
    }
    __syncthreads();

    // softmax over [0, pos]
    float mx = -INFINITY;
    for (int j = tid; j <= pos; j += ATT_THREADS) mx = fmaxf(mx, sc[j]);
    mx = wave_allmax(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j <= pos; j += ATT_THREADS) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_allsum(sum);
    __syncthreads();          // everyone has read red[] (max) before it is reused
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);

    // out[d] = sum_j p_j * V[j][d]: 16 key groups x 16 lanes of 8 dims (one 256-B V row per 16 lanes and load);
    // a thread walks keys kg, kg + 16, ... so position 64 is 4 iterations deep instead of 32
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int j = kg, i = 0; j <= pos; j += 16, ++i) {
        const float p = h2f(f2h(sc[j] * inv));
        if (j == pos) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += p * v_s[dg * 8 + e];
        } else {
            uint4 w;
            if (i < 4) w = i == 0 ? vspec[0] : i == 1 ? vspec[1] : i == 2 ? vspec[2] : vspec[3];
            else w = *(const uint4*)(vc + (int64_t)j * HD + dg * 8);
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[2 * e] += p * h2f((uint16_t)(ws[e] & 0xFFFF));
                o[2 * e + 1] += p * h2f((uint16_t)(ws[e] >> 16));
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o2[kg * HD + dg * 8 + e] = o[e];
    __syncthreads();
    if (tid < HD) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += o2[g * HD + tid];
        out[h * HD + tid] = f2h(t);
    }
    return true;
}

template <bool ROW>
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_kernel(const uint16_t* __restrict__ qkv,
                                                                  uint16_t* __restrict__ k_cache,
                                                                  uint16_t* __restrict__ v_cache,
                                                                  const int64_t* __restrict__ pos_p,
                                                                  const float* __restrict__ cos_t,
                                                                  const float* __restrict__ sin_t,
                                                                  uint16_t* __restrict__ out, int heads, int max_ctx) {
    extern __shared__ float sm[];
    attn_one_workgroup<ROW>(sm, blockIdx.x, qkv, k_cache, v_cache, pos_p, cos_t, sin_t, out, heads, max_ctx);
}

// LONG CONTEXTS (round 5): one workgroup per head streams a head's whole cache through ONE CU -- 5 us at 72 keys, 13 us at
// 450, ~40 us at 2048 (per layer).  Here the keys of a head are split over up to S workgroups (grid = heads x S, split s of
// head h = block s * heads + h): each computes the scores of its key range, their maximum m, p = exp(score - m), l = sum p
// and the unnormalised o = sum p v in fp32, parks {o[128], m, l} in the workspace (write-through) and bumps the head's
// arrival counter; the LAST arriver -- nobody waits -- merges the parts (o = sum_s o_s e^(m_s - M) / sum_s l_s e^(m_s - M)) and
// writes the head's output, then re-zeroes the counter.  Up to SPLIT_MIN keys a head is ONE workgroup running the
// single-workgroup algorithm above bit for bit (the other splits exit at once): short contexts pay nothing.
// ws: int counters [heads] (zeroed once by the caller; left zeroed), then f32 parts [heads][S][HD + 2] from byte 1024 * ceil(heads / 256).
constexpr int SPLIT_MIN = 128, SPLIT_CHUNK = 64, MAX_SPLITS = 16;
template <bool ROW>
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_split_kernel(const uint16_t* __restrict__ qkv,
                                                                        uint16_t* __restrict__ k_cache,
                                                                        uint16_t* __restrict__ v_cache,
                                                                        const int64_t* __restrict__ pos_p,
                                                                        const float* __restrict__ cos_t,
                                                                        const float* __restrict__ sin_t,
                                                                        uint16_t* __restrict__ out, int heads, int max_ctx, int S,
                                                                        int* __restrict__ cnt, float* __restrict__ part) {
    extern __shared__ float sm[];
    const int h = blockIdx.x % heads, split = blockIdx.x / heads;
    // split 0 starts as the one-workgroup kernel (all its loads fly with the position's); it comes back unhandled only for a
    // long context.  The other splits have nothing to do for a short or refused position.
    if (split == 0 && attn_one_workgroup<ROW>(sm, h, qkv, k_cache, v_cache, pos_p, cos_t, sin_t, out, heads, max_ctx, SPLIT_MIN))
        return;
    const int64_t pos64 = *pos_p;
    if (pos64 < 0 || pos64 >= max_ctx || pos64 + 1 <= SPLIT_MIN) return;
    __syncthreads();                                                    // (split 0: every thread is out of the function's LDS use)
    const int pos = (int)pos64, n = pos + 1;
    // equal key ranges in units of SPLIT_CHUNK keys over the splits that get any
    const int units = (n + SPLIT_CHUNK - 1) / SPLIT_CHUNK, S_max = units < S ? units : S;
    const int per_units = (units + S_max - 1) / S_max, per = per_units * SPLIT_CHUNK;
    // ceil(units / S_max) units per split leave the trailing splits EMPTY when units is not a multiple (513 keys over 8: 9 units,
    // 2 per split, splits 5..7 hold nothing; 1100 over 16: 18 units, splits 9..15): only the splits that hold a key take part,
    // so every counted part has nk >= 1 and a finite maximum (an empty part's {0, -inf, 0} merged correctly only
    // because exp(-inf - M) == 0)
    const int S_eff = (units + per_units - 1) / per_units;
    if (split >= S_eff) return;
    const int j0 = split * per, j1 = min(n, j0 + per);                  // j0 < n for every split < S_eff: the last one may be short, never empty
    float* q_s = sm;
    float* k_s = sm + HD;
    float* v_s = sm + 2 * HD;
    float* sc = sm + 3 * HD;                                            // scores of keys j0 .. j1 - 1
    float* red = sc + max_ctx;
    float* o2 = red + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hidden = heads * HD;
    uint16_t* kc = k_cache + (int64_t)h * max_ctx * HD;
    uint16_t* vc = v_cache + (int64_t)h * max_ctx * HD;
    const bool owns_new = pos >= j0 && pos < j1;                        // this split holds the token's own key
    const int d = tid & (HD - 1), d2 = d & (HD / 2 - 1);
    if (tid < HD) {
        const uint16_t* qh = qkv + h * HD;
        const uint16_t* kh = qkv + hidden + h * HD;
        const float c = ROW ? cos_t[d2] : cos_t[(int64_t)pos * (HD / 2) + d2], s_ = ROW ? sin_t[d2] : sin_t[(int64_t)pos * (HD / 2) + d2];
        const float q1 = h2f(qh[d2]), q2 = h2f(qh[d2 + HD / 2]);
        q_s[d] = h2f(f2h(d < HD / 2 ? q1 * c - q2 * s_ : q2 * c + q1 * s_));
        if (owns_new) {
            const float k1 = h2f(kh[d2]), k2 = h2f(kh[d2 + HD / 2]);
            const uint16_t krh = f2h(d < HD / 2 ? k1 * c - k2 * s_ : k2 * c + k1 * s_), vh = qkv[2 * hidden + h * HD + d];
            k_s[d] = h2f(krh);
            v_s[d] = h2f(vh);
            kc[(int64_t)pos * HD + d] = krh;
            vc[(int64_t)pos * HD + d] = vh;
        }
    }
    __syncthreads();
    const int part_ = tid & 3, kj = tid >> 2;
    const float scale = rsqrtf((float)HD);
    // cached keys [j0, jc) in a branch-free loop (two keys = eight 16-byte loads in flight per thread); the token's own key,
    // which is not in the cache yet for this launch's readers, from LDS afterwards
    const int jc = owns_new ? pos : j1;
    auto kdot = [&](const uint4 (&kw)[4]) {
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const uint32_t ws[4] = {kw[v].x, kw[v].y, kw[v].z, kw[v].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc += q_s[part_ * 32 + v * 8 + 2 * e] * h2f((uint16_t)(ws[e] & 0xFFFF));
                acc += q_s[part_ * 32 + v * 8 + 2 * e + 1] * h2f((uint16_t)(ws[e] >> 16));
            }
        }
        acc = quad_allsum(acc);
        return acc;
    };
    for (int j = j0 + kj; j < jc; j += 2 * (ATT_THREADS / 4)) {
        const int ja = j, jb = j + ATT_THREADS / 4;
        const bool vb = jb < jc;
        const uint4* ka = (const uint4*)(kc + (int64_t)ja * HD + part_ * 32);
        const uint4* kb = (const uint4*)(kc + (int64_t)(vb ? jb : ja) * HD + part_ * 32);
        uint4 wa[4], wb[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) wa[v] = ka[v];
#pragma unroll
        for (int v = 0; v < 4; ++v) wb[v] = kb[v];
        const float sa = kdot(wa), sb = kdot(wb);
        if (part_ == 0) {
            sc[ja - j0] = h2f(f2h(sa * scale));
            if (vb) sc[jb - j0] = h2f(f2h(sb * scale));
        }
    }
    if (owns_new && kj == ((pos - j0) & (ATT_THREADS / 4 - 1))) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) acc += q_s[part_ * 32 + e] * k_s[part_ * 32 + e];
        acc = quad_allsum(acc);
        if (part_ == 0) sc[pos - j0] = h2f(f2h(acc * scale));
    }
    __syncthreads();
    const int nk = j1 - j0;
    float mx = -INFINITY;
    for (int j = tid; j < nk; j += ATT_THREADS) mx = fmaxf(mx, sc[j]);
    mx = wave_allmax(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < nk; j += ATT_THREADS) {
        const float p = __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_allsum(sum);
    __syncthreads();
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float l = red[4] + red[5] + red[6] + red[7];
    const int dg = tid & 15, kg = tid >> 4;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int j = j0 + kg; j < jc; j += 64) {           // four cached V rows in flight per thread (clamped, weight 0 past the end)
        uint4 w[4];
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int jj = j + 16 * u;
            const bool ok = jj < jc;
            w[u] = *(const uint4*)(vc + (int64_t)(ok ? jj : j) * HD + dg * 8);
            p[u] = ok ? sc[jj - j0] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t ws[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[2 * e] += p[u] * h2f((uint16_t)(ws[e] & 0xFFFF));
                o[2 * e + 1] += p[u] * h2f((uint16_t)(ws[e] >> 16));
            }
        }
    }
    if (owns_new && kg == ((pos - j0) & 15)) {
        const float p = sc[pos - j0];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += p * v_s[dg * 8 + e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o2[kg * HD + dg * 8 + e] = o[e];
    __syncthreads();
    // park {o, m, l}: write-through (sc1) stores, drained before the arrival is counted
    float* mine = part + ((int64_t)h * S + split) * (HD + 2);
    if (tid < HD) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += o2[g * HD + tid];
        __hip_atomic_store(mine + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        __hip_atomic_store(mine + HD, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + HD + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int last_s;
    if (tid == 0) last_s = __hip_atomic_fetch_add(cnt + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S_eff - 1;
    __syncthreads();
    if (!last_s) return;
    // the last arriver of the head: every part was drained before its arrival was counted.  ALL the parts' loads go out
    // together (agent-scope: sc1 buffer loads; clamped to the last part, whose weight is then zero) -- read one after the
    // other, 16 parts were 16 dependent round trips, most of the launch at 2000 keys
    if (tid < HD) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(part + (int64_t)h * S * (HD + 2)), 0,
                                                                           S * (HD + 2) * 4, 0x00020000);
        float m_[MAX_SPLITS], l_[MAX_SPLITS], o_[MAX_SPLITS];
#pragma unroll
        for (int s2 = 0; s2 < MAX_SPLITS; ++s2) {
            const int off = (s2 < S_eff ? s2 : S_eff - 1) * (HD + 2) * 4;
            m_[s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + HD * 4, 0, 16));
            l_[s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + (HD + 1) * 4, 0, 16));
            o_[s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + tid * 4, 0, 16));
        }
        float M = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < MAX_SPLITS; ++s2) M = s2 < S_eff ? fmaxf(M, m_[s2]) : M;
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < MAX_SPLITS; ++s2) {
            const float w = s2 < S_eff ? __expf(m_[s2] - M) : 0.f;
            L += w * l_[s2];
            acc += w * o_[s2];
        }
        out[h * HD + tid] = f2h(acc / L);
    }
    if (tid == 0) __hip_atomic_store(cnt + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// cos / sin of the CURRENT position -> one [2][HD/2] row (one launch per token; the position lives in device memory)
// ... and, for the stage that owns the embedding, the token's embedding row in the same launch (tok nullable): a token's
// two table look-ups are one small launch instead of two (each costs ~4.5 us of launch + cold round trip)
__global__ __launch_bounds__(256) void rope_row_kernel(const int64_t* __restrict__ pos_p, const float* __restrict__ cos_t,
                                                       const float* __restrict__ sin_t, float* __restrict__ row, int half_dim,
                                                       int max_ctx, const int64_t* __restrict__ tok, const uint16_t* __restrict__ embed,
                                                       int vocab, int hidden, uint16_t* __restrict__ h_out) {
    int64_t pos = *pos_p;
    pos = pos < 0 ? 0 : pos >= max_ctx ? max_ctx - 1 : pos;    // (an out-of-range position is refused by the attention kernel itself)
    int64_t t = tok ? *tok : 0;
    t = t < 0 ? 0 : t >= vocab ? vocab - 1 : t;
    for (int d = threadIdx.x; d < half_dim; d += 256) {
        row[d] = cos_t[pos * half_dim + d];
        row[half_dim + d] = sin_t[pos * half_dim + d];
    }
    if (tok)
        for (int i = threadIdx.x; i < hidden / 8; i += 256) ((uint4*)h_out)[i] = ((const uint4*)(embed + t * hidden))[i];
}

// Final RMSNorm + lm_head (a plain fp16 Linear in the reference: MXQ quantises the decoder Linears only) + greedy
// argmax for ONE token, as two launches: (1) every workgroup normalises the hidden state like the torch path does
// (fp16(x * rsqrt(mean x^2 + eps)) * norm_w, one fp16 multiply), keeps it in registers -- a lane owns the same 64
// of the K = 4096 columns for every row -- and streams its share of the [V, K] fp16 weight, one wave per row, 8 KiB
// of loads in flight per wave; a logit is the fp32 dot rounded to fp16 (what F.linear returns); the running best
// (value, lowest index) per wave, then per workgroup, goes to a small partial buffer; (2) one wave reduces the
// partials and writes the token id.  Replaces ~12 small torch launches (124 us of a 1.53 ms token).
constexpr int LMH_THREADS = 256, LMH_K = 4096;
__global__ __launch_bounds__(LMH_THREADS) void lmhead_argmax_kernel(const uint16_t* __restrict__ h,
                                                                    const uint16_t* __restrict__ norm_w, float eps,
                                                                    const uint16_t* __restrict__ w, int V,
                                                                    float* __restrict__ part_val,
                                                                    int* __restrict__ part_idx) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    __shared__ float bval[4];
    __shared__ int bidx[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the lane's 64 columns: 8 chunks of 8 at k = 512 i + 8 lane
    h8 x[8], g[8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = *(const h8*)(h + i * 512 + lane * 8);
        g[i] = *(const h8*)(norm_w + i * 512 + lane * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += (float)x[i][j] * (float)x[i][j];
    }
    ss = wave_allsum(ss);
    const float inv = rsqrtf(ss / (float)LMH_K + eps);      // every wave holds the whole row: no cross-wave step
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) x[i][j] = (_Float16)((float)x[i][j] * inv) * g[i][j];

    float best = -INFINITY;
    int besti = 0x7FFFFFFF;
    const int stride = gridDim.x * 4;
    // Two rows in flight per wave (round 3): the next row's 8 loads are issued before the current row is reduced -- a
    // wave that loads, waits, reduces, loads ... pays one loaded memory latency per row (8 rows per wave at V = 32000:
    // 85 us for 262 MB).  Rows past V are loaded from the last row (no branch around a load) and never compared.
    auto load_row = [&](int row, h8 (&wv)[8]) {
        const int rc = row < V ? row : V - 1;
        const uint16_t* wr = w + (int64_t)rc * LMH_K + lane * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) wv[i] = *(const h8*)(wr + i * 512);
    };
    auto take_row = [&](int row, const h8 (&wv)[8]) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = __builtin_amdgcn_fdot2((h2){wv[i][2 * j], wv[i][2 * j + 1]}, (h2){x[i][2 * j], x[i][2 * j + 1]}, acc, false);
        acc = wave_allsum(acc);
        const float logit = (float)(_Float16)acc;           // F.linear's fp16 output
        if (row < V && (logit > best || (logit == best && row < besti))) { best = logit; besti = row; }   // (a NaN logit never wins)
    };
    h8 ra[8], rb[8];
    int row = blockIdx.x * 4 + wave;
    load_row(row, ra);
    for (; row < V; row += 2 * stride) {
        load_row(row + stride, rb);
        take_row(row, ra);
        load_row(row + 2 * stride, ra);
        take_row(row + stride, rb);
    }
    if (lane == 0) { bval[wave] = best; bidx[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k)
            if (bval[k] > best || (bval[k] == best && bidx[k] < besti)) { best = bval[k]; besti = bidx[k]; }
        part_val[blockIdx.x] = best;
        part_idx[blockIdx.x] = besti;
    }
}
// advance (nullable): the device-resident position -- the token id is also appended at generated[*advance] (nullable,
// max_generated slots) and the position moves on by one: the token loop's bookkeeping without its two tiny launches
__global__ __launch_bounds__(64) void lmhead_final_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                                          int n, int64_t* __restrict__ token, int64_t* __restrict__ advance,
                                                          int64_t* __restrict__ generated, int max_generated) {
    const int lane = threadIdx.x;
    float best = -INFINITY;
    int besti = 0x7FFFFFFF;
    // n <= 1024 partials (launcher): all of a lane's 16 loads go out together (clamped, never branched around) instead
    // of one dependent round trip per iteration (7.3 -> ~2 us of every token)
    float pv[16];
    int pi[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = lane + 64 * k;
        const int ic = i < n ? i : n - 1;
        pv[k] = part_val[ic];
        pi[k] = part_idx[ic];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const bool live = lane + 64 * k < n;
        if (live && (pv[k] > best || (pv[k] == best && pi[k] < besti))) { best = pv[k]; besti = pi[k]; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float v = __shfl_xor(best, o, 64);
        const int ix = __shfl_xor(besti, o, 64);
        if (v > best || (v == best && ix < besti)) { best = v; besti = ix; }
    }
    if (lane == 0) {
        const int64_t id = besti == 0x7FFFFFFF ? 0 : besti;
        *token = id;
        if (advance) {
            const int64_t p = *advance;
            if (generated && p >= 0 && p < max_generated) generated[p] = id;
            *advance = p + 1;
        }
    }
}

}   // namespace

int mxq_launch_lmhead_argmax_f16(const void* h, const void* norm_w, float eps, const void* w, int V, int K, void* part,
                                 int part_slots, void* token, void* advance, void* generated, int max_generated,
                                 hipStream_t stream) {
    if (K != LMH_K) return (int)hipErrorInvalidValue;
    int wgs = (V + 3) / 4;
    if (wgs > part_slots) wgs = part_slots;
    if (wgs > 1024) wgs = 1024;        // 4 workgroups per CU: 16 waves x 8 KiB in flight
    if (wgs < 1) return (int)hipErrorInvalidValue;
    float* pv = (float*)part;
    int* pi = (int*)(pv + part_slots);
    lmhead_argmax_kernel<<<wgs, LMH_THREADS, 0, stream>>>((const uint16_t*)h, (const uint16_t*)norm_w, eps,
                                                          (const uint16_t*)w, V, pv, pi);
    lmhead_final_kernel<<<1, 64, 0, stream>>>(pv, pi, wgs, (int64_t*)token, (int64_t*)advance, (int64_t*)generated, max_generated);
    return (int)hipGetLastError();
}

template <bool ROW>
static int launch_attn(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* cos_t, const void* sin_t,
                       void* out, int heads, int head_dim, int max_ctx, hipStream_t stream) {
    if (head_dim != HD) return (int)hipErrorInvalidValue;
    const size_t smem = (size_t)(3 * HD + max_ctx + 8 + 16 * HD) * 4;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_decode_kernel<ROW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    attn_decode_kernel<ROW><<<heads, ATT_THREADS, smem, stream>>>((const uint16_t*)qkv, (uint16_t*)k_cache,
                                                                  (uint16_t*)v_cache, (const int64_t*)pos,
                                                                  (const float*)cos_t, (const float*)sin_t, (uint16_t*)out,
                                                                  heads, max_ctx);
    return (int)hipGetLastError();
}

int mxq_launch_attn_decode_f16(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* cos_t,
                               const void* sin_t, void* out, int heads, int head_dim, int max_ctx, int rope_row,
                               hipStream_t stream) {
    return rope_row ? launch_attn<true>(qkv, k_cache, v_cache, pos, cos_t, sin_t, out, heads, head_dim, max_ctx, stream)
                    : launch_attn<false>(qkv, k_cache, v_cache, pos, cos_t, sin_t, out, heads, head_dim, max_ctx, stream);
}

size_t mxq_attn_split_workspace_bytes_impl(int heads, int splits) {
    return (size_t)((heads + 255) / 256) * 1024 + (size_t)heads * splits * (HD + 2) * sizeof(float);
}
// splits > 1: the long-context kernel (ws = mxq_attn_split_workspace_bytes(heads, splits), counters zeroed once)
int mxq_launch_attn_decode_split_f16(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* rope_row,
                                     void* out, int heads, int head_dim, int max_ctx, int splits, void* ws, hipStream_t stream) {
    if (head_dim != HD || splits > MAX_SPLITS) return (int)hipErrorInvalidValue;
    const size_t smem = (size_t)(3 * HD + max_ctx + 8 + 16 * HD) * 4;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_decode_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    int* cnt = (int*)ws;
    float* part = (float*)((char*)ws + (size_t)((heads + 255) / 256) * 1024);
    attn_decode_split_kernel<true><<<heads * splits, ATT_THREADS, smem, stream>>>(
        (const uint16_t*)qkv, (uint16_t*)k_cache, (uint16_t*)v_cache, (const int64_t*)pos, (const float*)rope_row,
        (const float*)rope_row + HD / 2, (uint16_t*)out, heads, max_ctx, splits, cnt, part);
    return (int)hipGetLastError();
}

int mxq_launch_rope_row_f32(const void* pos, const void* cos_t, const void* sin_t, void* row, int half_dim, int max_ctx,
                            const void* tok, const void* embed, int vocab, int hidden, void* h_out, hipStream_t stream) {
    rope_row_kernel<<<1, 256, 0, stream>>>((const int64_t*)pos, (const float*)cos_t, (const float*)sin_t, (float*)row, half_dim,
                                           max_ctx, (const int64_t*)tok, (const uint16_t*)embed, vocab, hidden, (uint16_t*)h_out);
    return (int)hipGetLastError();
}
