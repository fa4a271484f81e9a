// Probe (round 6): what would it buy the decode layer to make o_proj the TAIL of the attention launch?
// Today: attention (32 workgroups, one per head: ~5.1 us) -> launch boundary -> o_proj GEMV (256 workgroups: 9.4 MB of packed
// weights + the 8-KB attention output: ~5.4 us).  Candidate: ONE launch of 32 + 256 workgroups -- the 256 stream their 36-KB
// weight slabs into registers FIRST, then wait ONCE on a counter the 32 bump when the head outputs are written, read the 8 KB
// and finish.  This probe runs the two schedules on stand-in bodies with the real sizes and dependency structure (two dependent
// 18-KB load phases per head on the producer side -- scores, then values; 36 KB per workgroup + 8 KB shared on the consumer
// side), streaming distinct weights from HBM every iteration:
//   A  two launches (the boundary is the edge)
//   B  one launch, consumers prefetch their slab, then acquire-spin on the counter (agent scope), then read x
//   C  the consumer launch alone (its own floor), D the producer launch alone
// under hipGraph replay.  Every spin is bounded (an expired spin sets err and the workgroup leaves).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/edge_probe.hip -o /tmp/edge_probe && /tmp/edge_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int HEADS = 32, HD = 128, HID = HEADS * HD, KEYS = 72;
constexpr int ROWBLK = 256, SLAB = 36864;                 // o_proj: 256 row blocks x 36 KB of packed weights
constexpr int THREADS = 256;

__device__ __forceinline__ float dot8(uint4 w, float acc) {
    return acc + __uint_as_float((w.x & 0x007fffffu) | 0x3f800000u) + __uint_as_float((w.y & 0x007fffffu) | 0x3f800000u) +
           __uint_as_float((w.z & 0x007fffffu) | 0x3f800000u) + __uint_as_float((w.w & 0x007fffffu) | 0x3f800000u);
}

// stand-in for the attention of one head: q from x_in, two DEPENDENT passes over 18 KB (keys, then values), 128 outputs
template <bool WT = false>
__device__ __forceinline__ void producer_body(const uint4* __restrict__ kv, const float* __restrict__ x_in, float* __restrict__ x_out, int h) {
    __shared__ float red[THREADS];
    const int tid = threadIdx.x;
    const uint4* k = kv + (size_t)h * (2 * KEYS * HD * 2 / 16);
    float q = x_in[h * HD + (tid & (HD - 1))];
    float s = 0.f;
    for (int i = tid; i < KEYS * HD * 2 / 16; i += THREADS) s = dot8(k[i], s);          // "scores"
    red[tid] = s * q;
    __syncthreads();
    float m = red[tid & 63] + red[64 + (tid & 63)] + red[128 + (tid & 63)] + red[192 + (tid & 63)];   // "softmax" needs them all
    const uint4* v = k + KEYS * HD * 2 / 16;
    float o = 0.f;
    for (int i = tid; i < KEYS * HD * 2 / 16; i += THREADS) o = dot8(v[i], o);          // "values", after the scores
    __syncthreads();
    red[tid] = o * m;
    __syncthreads();
    if (tid < HD) {
        if (WT) __hip_atomic_store(x_out + h * HD + tid, red[tid] + red[tid + 128], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
        else x_out[h * HD + tid] = red[tid] + red[tid + 128];
    }
}

// stand-in for one 16-row block of the o_proj GEMV: 36 KB of weights (9 x 16 B per thread), the 8-KB x row, 16 outputs
struct Slab { uint4 w[9]; };
__device__ __forceinline__ void consumer_load(const uint4* __restrict__ wts, int rb, Slab& s) {
    const uint4* p = wts + (size_t)rb * (SLAB / 16) + threadIdx.x;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const u32x4 v = __builtin_nontemporal_load((const u32x4*)(p + i * THREADS));
        s.w[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
}
template <bool WT = false>
__device__ __forceinline__ void consumer_finish(const Slab& s, const float* __restrict__ x, float* __restrict__ y, int rb) {
    __shared__ float xs[HID];
    __shared__ float red2[THREADS];
    const int tid = threadIdx.x;
    if (WT) {
#pragma unroll
        for (int i = 0; i < HID / THREADS; ++i)     // agent-scope loads: past this XCD's L2, all 16 in flight together
            xs[tid + i * THREADS] = __hip_atomic_load(x + tid + i * THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        for (int i = tid; i < HID; i += THREADS) xs[i] = x[i];
    }
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) a = dot8(s.w[i], a) * xs[(tid * 9 + i) & (HID - 1)];
    red2[tid] = a;
    __syncthreads();
    if (tid < 16) {
        float t = 0.f;
        for (int j = 0; j < 16; ++j) t += red2[tid * 16 + j];
        y[rb * 16 + tid] = t;
    }
}

__global__ __launch_bounds__(THREADS) void producer_kernel(const uint4* kv, const float* x_in, float* x_out) {
    producer_body(kv, x_in, x_out, blockIdx.x);
}
__global__ __launch_bounds__(THREADS) void consumer_kernel(const uint4* wts, const float* x, float* y) {
    Slab s;
    consumer_load(wts, blockIdx.x, s);
    consumer_finish(s, x, y, blockIdx.x);
}
// launch floor: a kernel that does nothing / one dependent 4-byte load + store per workgroup
__global__ void empty_kernel() {}
__global__ __launch_bounds__(THREADS) void touch_kernel(const float* x, float* y) {
    if (threadIdx.x == 0) y[blockIdx.x] = x[(blockIdx.x * 37) & (HID - 1)] + 1.f;
}
// one launch: workgroups 0..31 produce, 32..287 consume; `target` = the count that says "all heads of THIS launch written"
template <int SLEEP, bool PREFETCH>
__global__ __launch_bounds__(THREADS) void fused_kernel(const uint4* kv, const float* x_in, float* x_mid, const uint4* wts, float* y,
                                                        unsigned* cnt, unsigned target, unsigned* err) {
    if (blockIdx.x < HEADS) {
        if (SLEEP >= 1000) {   // no fences at all: write-through stores, drained, then a relaxed count
            producer_body<true>(kv, x_in, x_mid, blockIdx.x);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        producer_body(kv, x_in, x_mid, blockIdx.x);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int rb = blockIdx.x - HEADS;
    Slab s;
    if (PREFETCH) consumer_load(wts, rb, s);                 // the slab is on its way before the wait
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int i = 0; i < (1 << 20); ++i) {
            // (relaxed polls: an ACQUIRE load in the loop invalidates this XCD's L2 on every poll -- 256 pollers doing that
            //  made the launch 4 x slower than the two-launch form, 40.8 us; the one acquire fence follows the loop)
            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(SLEEP >= 1000 ? SLEEP - 1000 : SLEEP);
        }
        if (!ok) *err = 1;
        ok_s = ok;
    }
    __syncthreads();
    if (!ok_s) return;
    if (SLEEP >= 1000) {
        if (!PREFETCH) consumer_load(wts, rb, s);
        consumer_finish<true>(s, x_mid, y, rb);
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // every wave: the x row is read past this XCD's L2
    if (!PREFETCH) consumer_load(wts, rb, s);
    consumer_finish(s, x_mid, y, rb);
}

int main() {
    const int COPIES = 64, LAYERS = 64;                       // 64 distinct weight / cache copies: 600 MB streamed per pass
    uint4 *wts, *kv;
    float *x0, *xm, *y;
    unsigned *cnt, *err;
    const size_t wbytes = (size_t)ROWBLK * SLAB, kvbytes = (size_t)HEADS * 2 * KEYS * HD * 2;
    CHECK(hipMalloc(&wts, wbytes * COPIES));
    CHECK(hipMalloc(&kv, kvbytes * COPIES));
    CHECK(hipMalloc(&x0, HID * 4));
    CHECK(hipMalloc(&xm, HID * 4));
    CHECK(hipMalloc(&y, HID * 4));
    CHECK(hipMalloc(&cnt, 256));
    CHECK(hipMalloc(&err, 4));
    CHECK(hipMemset(wts, 0x3c, wbytes * COPIES));
    CHECK(hipMemset(kv, 0x3c, kvbytes * COPIES));
    CHECK(hipMemset(x0, 0, HID * 4));
    CHECK(hipMemset(cnt, 0, 256));
    CHECK(hipMemset(err, 0, 4));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const char* names[12] = {"A two launches (attention-like, then o_proj-like)", "B one launch, consumers prefetch then wait on a counter (s_sleep 2)",
                            "C the consumer launch alone", "D the producer launch alone", "B' as B, s_sleep 32 between polls",
                            "B'' as B, s_sleep 127", "E one launch, consumers wait FIRST, then load (no prefetch; s_sleep 32)",
                            "F one launch, NO fences: write-through x, relaxed count, agent-scope x loads; prefetch (s_sleep 8)",
                            "G as F without the prefetch", "H an EMPTY kernel (1 workgroup)", "I an empty kernel, 256 workgroups",
                            "J 256 workgroups, one dependent 4-byte load + store each"};
    unsigned epoch = 0;
    for (int mode = 0; mode < 12; ++mode) {
        hipGraph_t g;
        hipGraphExec_t ge;
        // the graph: LAYERS dependent steps on distinct copies (as a token walks the layers)
        auto body = [&](unsigned& ep) {
            for (int l = 0; l < LAYERS; ++l) {
                const uint4* w = wts + (size_t)(l % COPIES) * (wbytes / 16);
                const uint4* k = kv + (size_t)(l % COPIES) * (kvbytes / 16);
                if (mode == 9) {
                    empty_kernel<<<1, 64, 0, st>>>();
                } else if (mode == 10) {
                    empty_kernel<<<256, 256, 0, st>>>();
                } else if (mode == 11) {
                    touch_kernel<<<256, 256, 0, st>>>((l & 1) ? xm : x0, (l & 1) ? x0 : xm);
                } else if (mode == 0) {
                    producer_kernel<<<HEADS, THREADS, 0, st>>>(k, x0, xm);
                    consumer_kernel<<<ROWBLK, THREADS, 0, st>>>(w, xm, x0);
                } else if (mode == 1 || (mode >= 4 && mode <= 8)) {
                    ep += HEADS;
                    if (mode == 1) fused_kernel<2, true><<<HEADS + ROWBLK, THREADS, 0, st>>>(k, x0, xm, w, x0, cnt, ep, err);
                    else if (mode == 4) fused_kernel<32, true><<<HEADS + ROWBLK, THREADS, 0, st>>>(k, x0, xm, w, x0, cnt, ep, err);
                    else if (mode == 5) fused_kernel<127, true><<<HEADS + ROWBLK, THREADS, 0, st>>>(k, x0, xm, w, x0, cnt, ep, err);
                    else if (mode == 6) fused_kernel<32, false><<<HEADS + ROWBLK, THREADS, 0, st>>>(k, x0, xm, w, x0, cnt, ep, err);
                    else if (mode == 7) fused_kernel<1008, true><<<HEADS + ROWBLK, THREADS, 0, st>>>(k, x0, xm, w, x0, cnt, ep, err);
                    else fused_kernel<1008, false><<<HEADS + ROWBLK, THREADS, 0, st>>>(k, x0, xm, w, x0, cnt, ep, err);
                } else if (mode == 2) {
                    consumer_kernel<<<ROWBLK, THREADS, 0, st>>>(w, xm, x0);
                } else {
                    producer_kernel<<<HEADS, THREADS, 0, st>>>(k, x0, xm);
                }
            }
        };
        // mode 1's counter runs on monotonically: a graph bakes its targets in, so every replay starts from a zeroed counter
        CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        if (mode == 1 || (mode >= 4 && mode <= 8)) CHECK(hipMemsetAsync(cnt, 0, 4, st));
        unsigned ep = 0;
        body(ep);
        CHECK(hipStreamEndCapture(st, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipGraphLaunch(ge, st));
            CHECK(hipEventRecord(e0, st));
            CHECK(hipGraphLaunch(ge, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        unsigned herr = 0;
        CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%s: %.2f us per layer step (err %u)\n", names[mode], best * 1e3f / LAYERS, herr);
        (void)epoch;
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }
    return 0;
}
