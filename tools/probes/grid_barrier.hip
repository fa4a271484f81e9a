// Probe: what does a device-wide barrier cost inside one persistent kernel on MI355X (8 XCDs, per-XCD L2)?
// Decides whether the decode layer chain (5 dependent launches per layer, each paying launch gap + first-load
// latency) could become ONE persistent kernel with grid barriers between the phases.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Every spin is bounded: a barrier that does not complete sets an error flag and the kernel exits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target) {
    for (int i = 0; i < (1 << 22); ++i) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// mode 0: one counter; mode 1: per-XCD counters (blockIdx % 8), the last arriver of an XCD bumps the global one
__global__ void barrier_kernel(unsigned* ctr, unsigned* xcd_ctr, unsigned* data, unsigned* err, unsigned long long* cyc,
                               int iters, int mode, int check) {
    const unsigned nwg = gridDim.x, wg = blockIdx.x;
    const unsigned xcd = wg & 7, per_xcd = (nwg + 7 - xcd) / 8;
    unsigned long long t0 = wall_clock64();
    bool ok = true;
    for (int it = 0; it < iters && ok; ++it) {
        if (check && threadIdx.x == 0) data[wg] = it * 131u + wg;          // plain store, published by the release below
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (mode == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const unsigned old = __hip_atomic_fetch_add(xcd_ctr + xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)(it + 1) * per_xcd - 1)
                    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ok = spin_until(ctr, (unsigned)(it + 1) * (mode == 0 ? nwg : 8u));
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (!ok) *err = 1;
        }
        ok = __syncthreads_or(!ok) == 0;
        if (check && ok && threadIdx.x == 0) {
            const unsigned other = (wg * 37u + 11u + it) % nwg;
            if (data[other] != it * 131u + other) *err = 2;
        }
        if (check) __syncthreads();
    }
    if (threadIdx.x == 0 && wg == 0) *cyc = wall_clock64() - t0;
}

int main() {
    unsigned *ctr, *xcd_ctr, *data, *err;
    unsigned long long* cyc;
    CHECK(hipMalloc(&ctr, 256));
    CHECK(hipMalloc(&xcd_ctr, 8 * 128));
    CHECK(hipMalloc(&data, 4096 * 4));
    CHECK(hipMalloc(&err, 4));
    CHECK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    for (int check = 0; check < 2; ++check)
        for (int mode = 0; mode < 2; ++mode)
            for (int nwg : {256, 512, 1024})
                for (int threads : {256}) {
                    CHECK(hipMemset(ctr, 0, 256));
                    CHECK(hipMemset(xcd_ctr, 0, 8 * 128));
                    CHECK(hipMemset(err, 0, 4));
                    void* args[] = {&ctr, &xcd_ctr, &data, &err, &cyc, (void*)&iters, &mode, &check};
                    hipError_t e = hipLaunchCooperativeKernel((const void*)barrier_kernel, dim3(nwg), dim3(threads), args, 0, 0);
                    if (e != hipSuccess) { printf("nwg %d: cooperative launch refused: %s\n", nwg, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
                    CHECK(hipDeviceSynchronize());
                    unsigned h_err; unsigned long long h_cyc;
                    CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
                    CHECK(hipMemcpy(&h_cyc, cyc, 8, hipMemcpyDeviceToHost));
                    printf("check %d mode %d (%s) workgroups %4d x %d threads: %.2f us per barrier, err %u\n", check, mode,
                           mode ? "per-XCD + global" : "one counter", nwg, threads, h_cyc * 0.01 / iters, h_err);
                }
    return 0;
}
