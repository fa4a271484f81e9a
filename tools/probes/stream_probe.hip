// Probe (GPU box): what limits a decode GEMV that streams 36.9 KB row blocks of packed weights?  Times, over a
// >= 600 MB set of distinct buffers (HBM-cold), loads-only kernels in the access patterns of the two GEMV kernels
// against an ideal 16-B-per-lane grid-stride read, with a dummy VALU load of V ops per 64 B to emulate the dequant.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe.hip -o abtmp/stream_probe && abtmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int BLK_DW = 144;

template <int V>
__device__ __forceinline__ unsigned churn(unsigned a, unsigned b) {   // V dependent-ish VALU ops
#pragma unroll
    for (int i = 0; i < V; ++i) a = __builtin_amdgcn_perm(a, b, 0x05010400u + i) ^ (a >> 1);
    return a;
}

// pattern A: lane -> (row r, chunk slot cs), 13 four-byte loads per 576-B block (gemv.hip); THREADS per row block
template <int THREADS, int DEPTH, int V>
__global__ __launch_bounds__(THREADS) void pat_a(const unsigned* __restrict__ w, unsigned* __restrict__ out, int NC) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, cs = lane >> 4;
    constexpr int WAVES = THREADS / 64;
    const unsigned* tiles = w + (size_t)blockIdx.x * NC * BLK_DW;
    const int NC4 = (NC + 3) / 4;
    unsigned acc = 0;
    unsigned buf[DEPTH][13];
    auto load = [&](int c4, unsigned (&t)[13]) {
        const int chunk = c4 * 4 + cs;
        if (chunk < NC) {
            const unsigned* p = tiles + (size_t)chunk * BLK_DW;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = p[i * 16 + r];          // C2[3], C4[2], Z2[3]
            t[8] = p[128 + (r >> 1)];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[9 + i] = p[136 + 2 * (i & 3 ? i - 1 : 0) + (i & 1)];
        } else {
#pragma unroll
            for (int i = 0; i < 13; ++i) t[i] = 0;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(wave + d * WAVES, buf[d]);
    for (int c4 = wave; c4 < NC4; c4 += WAVES * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            unsigned t[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) t[i] = buf[d][i];
            load(c4 + (d + DEPTH) * WAVES, buf[d]);
#pragma unroll
            for (int i = 0; i < 13; ++i) acc ^= t[i];
            acc = churn<V>(acc, t[0]);
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// VALU only: the churn of pattern A's tiles with no memory traffic at all (calibrates what the VALU share costs alone)
template <int THREADS, int V>
__global__ __launch_bounds__(THREADS) void valu_only(unsigned* __restrict__ out, int NC, unsigned seed) {
    const int tid = threadIdx.x, wave = tid >> 6;
    constexpr int WAVES = THREADS / 64;
    const int NC4 = (NC + 3) / 4;
    unsigned acc = seed ^ tid;
    for (int c4 = wave; c4 < NC4; c4 += WAVES) acc = churn<V>(acc, acc * 3u + 1u);
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// pattern A, persistent: gridDim.x workgroups walk the row blocks (rb = blockIdx.x + j * gridDim.x); a wave's tiles of
// consecutive row blocks form ONE stream with DEPTH tiles in flight across row-block boundaries
template <int THREADS, int DEPTH, int V>
__global__ __launch_bounds__(THREADS) void pat_ap(const unsigned* __restrict__ w, unsigned* __restrict__ out, int NC, int RB) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, cs = lane >> 4;
    constexpr int WAVES = THREADS / 64;
    const int NC4 = (NC + 3) / 4;
    const int per_rb = (NC4 - wave + WAVES - 1) / WAVES;            // this wave's tiles per row block
    const int my_rbs = (RB - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = per_rb * my_rbs;
    unsigned acc = 0;
    unsigned buf[DEPTH][13];
    auto load = [&](int i, unsigned (&t)[13]) {
        if (i < total) {
            const int j = i / per_rb, c4 = wave + (i - j * per_rb) * WAVES;
            const int rb = blockIdx.x + j * gridDim.x, chunk = c4 * 4 + cs;
            if (chunk < NC) {
                const unsigned* p = w + ((size_t)rb * NC + chunk) * BLK_DW;
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = p[k * 16 + r];
                t[8] = p[128 + (r >> 1)];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[9 + k] = p[136 + k];
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < 13; ++k) t[k] = 0;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d, buf[d]);
    for (int i = 0; i < total; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            unsigned t[13];
#pragma unroll
            for (int k = 0; k < 13; ++k) t[k] = buf[d][k];
            load(i + d + DEPTH, buf[d]);
#pragma unroll
            for (int k = 0; k < 13; ++k) acc ^= t[k];
            acc = churn<V>(acc, t[0]);
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// pattern B: 16-B loads, lane -> (block b = lane / 4, row quad q), wave role = field group (gemv2.hip)
template <int TEAMS, int V>
__global__ __launch_bounds__(TEAMS * 256) void pat_b(const unsigned* __restrict__ w, unsigned* __restrict__ out, int NC) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, team = wave >> 2, role = wave & 3;
    const int b = lane >> 2, q = lane & 3;
    const unsigned* tiles = w + (size_t)blockIdx.x * NC * BLK_DW;
    const int offA = role == 3 ? 48 + 4 * q : role * 16 + 4 * q, offB = role == 3 ? 64 + 4 * q : 80 + role * 16 + 4 * q;
    const int n_it = (NC + 16 * TEAMS - 1) / (16 * TEAMS);
    unsigned acc = 0;
    u4 a = {}, bb = {};
    u2 s = {}, qq = {};
    auto load = [&](int it) {
        const int chunk = (it * TEAMS + team) * 16 + b;
        if (chunk < NC) {
            const unsigned* p = tiles + (size_t)chunk * BLK_DW;
            a = *(const u4*)(p + offA);
            bb = *(const u4*)(p + offB);
            if (role != 3) { s = *(const u2*)(p + 128 + 2 * q); qq = *(const u2*)(p + 136 + 2 * role); }
        }
    };
    load(0);
    for (int it = 0; it < n_it; ++it) {
        const u4 ca = a, cb = bb;
        const u2 cs_ = s, cq = qq;
        if (it + 1 < n_it) load(it + 1);
        acc ^= ca[0] ^ ca[1] ^ ca[2] ^ ca[3] ^ cb[0] ^ cb[1] ^ cb[2] ^ cb[3] ^ cs_[0] ^ cs_[1] ^ cq[0] ^ cq[1];
        acc = churn<V>(acc, ca[0]);
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// ideal: grid-stride 16-B-per-lane read of the whole buffer, 4 loads in flight per lane
__global__ __launch_bounds__(256) void ideal(const unsigned* __restrict__ w, unsigned* __restrict__ out, size_t n16) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4* p = (const u4*)w;
    unsigned acc = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a[0] ^ a[3] ^ b[1] ^ b[2] ^ c[0] ^ c[3] ^ d[1] ^ d[2];
    }
    for (; i < n16; i += stride) acc ^= p[i][0];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <class F>
double time_it(F&& launch, int nbuf) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<double> ts;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < nbuf; ++i) launch(i);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms * 1e3 / nbuf);
    }
    std::sort(ts.begin(), ts.end());
    return ts[2];
}

int main() {
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int N = cfg == 0 ? 22016 : 4096, K = 4096, NC = K / 64, RB = N / 16;
        const size_t bytes = (size_t)RB * NC * BLK_DW * 4;
        const int nbuf = (int)(640e6 / bytes) + 1;
        std::vector<unsigned*> bufs(nbuf);
        for (auto& b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 1, bytes); }
        unsigned* out;
        hipMalloc(&out, RB * 4 + 4096 * 4);
        hipDeviceSynchronize();
        printf("N=%d K=%d: %.1f MB per buffer x %d buffers (eager launches back to back; us per launch, TB/s)\n", N, K, bytes / 1e6, nbuf);
        auto rep = [&](const char* name, double us) { printf("  %-44s %7.2f us  %5.2f TB/s\n", name, us, bytes / us / 1e6); };
        rep("ideal 16 B/lane grid-stride, 2048 WGs", time_it([&](int i) { ideal<<<2048, 256>>>(bufs[i], out, bytes / 16); }, nbuf));
        rep("A 512 thr, depth 1, loads only", time_it([&](int i) { pat_a<512, 1, 0><<<RB, 512>>>(bufs[i], out, NC); }, nbuf));
        rep("A 512 thr, depth 2, loads only", time_it([&](int i) { pat_a<512, 2, 0><<<RB, 512>>>(bufs[i], out, NC); }, nbuf));
        rep("A 256 thr, depth 2, loads only", time_it([&](int i) { pat_a<256, 2, 0><<<RB, 256>>>(bufs[i], out, NC); }, nbuf));
        rep("A 1024 thr, depth 1, loads only", time_it([&](int i) { pat_a<1024, 1, 0><<<RB, 1024>>>(bufs[i], out, NC); }, nbuf));
        rep("A 512 thr, depth 1, +150 VALU per tile", time_it([&](int i) { pat_a<512, 1, 50><<<RB, 512>>>(bufs[i], out, NC); }, nbuf));
        rep("A 512 thr, depth 1, +345 VALU per tile", time_it([&](int i) { pat_a<512, 1, 115><<<RB, 512>>>(bufs[i], out, NC); }, nbuf));
        rep("A 512 thr, depth 2, +345 VALU per tile", time_it([&](int i) { pat_a<512, 2, 115><<<RB, 512>>>(bufs[i], out, NC); }, nbuf));
        rep("A 256 thr, depth 2, +345 VALU per tile", time_it([&](int i) { pat_a<256, 2, 115><<<RB, 256>>>(bufs[i], out, NC); }, nbuf));
        rep("VALU only, 512 thr, 345 per tile", time_it([&](int i) { valu_only<512, 115><<<RB, 512>>>(out, NC, i); }, nbuf));
        rep("VALU only, 1024 thr, 345 per tile", time_it([&](int i) { valu_only<1024, 115><<<RB, 1024>>>(out, NC, i); }, nbuf));
        rep("VALU only, 512 thr, 150 per tile", time_it([&](int i) { valu_only<512, 50><<<RB, 512>>>(out, NC, i); }, nbuf));
        rep("A 1024 thr, depth 1, +345 VALU per tile", time_it([&](int i) { pat_a<1024, 1, 115><<<RB, 1024>>>(bufs[i], out, NC); }, nbuf));
        rep("A persistent 256x1024thr, depth 2, loads only", time_it([&](int i) { pat_ap<1024, 2, 0><<<256, 1024>>>(bufs[i], out, NC, RB); }, nbuf));
        rep("A persistent 256x1024thr, depth 2, +VALU", time_it([&](int i) { pat_ap<1024, 2, 115><<<256, 1024>>>(bufs[i], out, NC, RB); }, nbuf));
        rep("A persistent 256x1024thr, depth 3, +VALU", time_it([&](int i) { pat_ap<1024, 3, 115><<<256, 1024>>>(bufs[i], out, NC, RB); }, nbuf));
        rep("A persistent 512x512thr, depth 2, +VALU", time_it([&](int i) { pat_ap<512, 2, 115><<<512, 512>>>(bufs[i], out, NC, RB); }, nbuf));
        rep("A persistent 512x512thr, depth 3, +VALU", time_it([&](int i) { pat_ap<512, 3, 115><<<512, 512>>>(bufs[i], out, NC, RB); }, nbuf));
        rep("A persistent 768x512thr, depth 2, +VALU", time_it([&](int i) { pat_ap<512, 2, 115><<<768, 512>>>(bufs[i], out, NC, RB); }, nbuf));
        rep("B 2 teams, loads only", time_it([&](int i) { pat_b<2, 0><<<RB, 512>>>(bufs[i], out, NC); }, nbuf));
        rep("B 1 team, loads only", time_it([&](int i) { pat_b<1, 0><<<RB, 256>>>(bufs[i], out, NC); }, nbuf));
        rep("B 2 teams, +345 VALU per iteration", time_it([&](int i) { pat_b<2, 115><<<RB, 512>>>(bufs[i], out, NC); }, nbuf));
        for (auto b : bufs) hipFree(b);
        hipFree(out);
    }
    return 0;
}
