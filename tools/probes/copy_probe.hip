// Probe (round 6): what is the chip's ceiling for ONE read + ONE write per element -- the traffic of the fake-quant forward
// (0.60 of 8 TB/s, "at device-copy speed")?  Copies 202 M bf16 elements (one decoder block's weights: 405 MB in,
// 405 MB out) with hipMemcpyAsync and with hand-written kernels: 16-byte accesses, default / nontemporal policy, by grid size and
// unroll depth.     hipcc --offload-arch=gfx950 -O3 tools/probes/copy_probe.hip -o /tmp/copy_probe && /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT_ST) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// each workgroup owns a CONTIGUOUS chunk (as a row-resident kernel does) instead of a grid-stride interleave
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_chunk_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, size_t per_wg) {
    const size_t b0 = (size_t)blockIdx.x * per_wg, b1 = b0 + per_wg < n16 ? b0 + per_wg : n16;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256 * UNROLL) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const size_t k = i + u * 256; v[u] = k < b1 ? (NT ? __builtin_nontemporal_load(src + k) : src[k]) : u32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const size_t k = i + u * 256; if (k < b1) { if (NT) __builtin_nontemporal_store(v[u], dst + k); else dst[k] = v[u]; } }
    }
}

int main() {
    const size_t n = 202375168, bytes = n * 2, n16 = bytes / 16;
    void *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto report = [&](const char* name, float ms) { printf("%-64s %7.1f us  %6.2f TB/s (read + write)\n", name, ms * 1e3f, 2.0 * bytes / (ms * 1e-3) / 1e12); };
#define TIME(name, launch)                                                                  \
    {                                                                                       \
        float best = 1e9f;                                                                  \
        for (int rep = 0; rep < 6; ++rep) {                                                 \
            CHECK(hipEventRecord(e0, st));                                                  \
            launch;                                                                         \
            CHECK(hipEventRecord(e1, st));                                                  \
            CHECK(hipStreamSynchronize(st));                                                \
            float ms;                                                                       \
            CHECK(hipEventElapsedTime(&ms, e0, e1));                                        \
            if (rep > 0 && ms < best) best = ms;                                            \
        }                                                                                   \
        report(name, best);                                                                 \
    }
    TIME("hipMemcpyAsync device -> device", CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st)));
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid-stride, %5d workgroups, unroll 4, default policy", g);
        TIME(nm, (copy_kernel<4, false, false><<<g, 256, 0, st>>>((const u32x4*)a, (u32x4*)b, n16)));
        snprintf(nm, sizeof nm, "grid-stride, %5d workgroups, unroll 4, nt loads + nt stores", g);
        TIME(nm, (copy_kernel<4, true, true><<<g, 256, 0, st>>>((const u32x4*)a, (u32x4*)b, n16)));
        snprintf(nm, sizeof nm, "grid-stride, %5d workgroups, unroll 8, nt loads + nt stores", g);
        TIME(nm, (copy_kernel<8, true, true><<<g, 256, 0, st>>>((const u32x4*)a, (u32x4*)b, n16)));
        snprintf(nm, sizeof nm, "grid-stride, %5d workgroups, unroll 4, nt stores only", g);
        TIME(nm, (copy_kernel<4, false, true><<<g, 256, 0, st>>>((const u32x4*)a, (u32x4*)b, n16)));
    }
    for (int g : {2048, 8192, 32768}) {
        char nm[96];
        const size_t per = (n16 + g - 1) / g;
        snprintf(nm, sizeof nm, "contiguous chunk per workgroup, %5d workgroups, unroll 4, nt", g);
        TIME(nm, (copy_chunk_kernel<4, true><<<g, 256, 0, st>>>((const u32x4*)a, (u32x4*)b, n16, per)));
        snprintf(nm, sizeof nm, "contiguous chunk per workgroup, %5d workgroups, unroll 8, default", g);
        TIME(nm, (copy_chunk_kernel<8, false><<<g, 256, 0, st>>>((const u32x4*)a, (u32x4*)b, n16, per)));
    }
    return 0;
}
