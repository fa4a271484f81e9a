// Probe (GPU box): how much vector-issue time do MFMAs of the two 16-bit shapes leave for a VALU wave on the same SIMD?
// One workgroup per CU: 8 "matrix" waves (2 per SIMD) issue back-to-back MFMAs on independent accumulators -- either
// v_mfma_f32_16x16x32_f16 (16 cycles each) or v_mfma_f32_32x32x16_f16 (32 cycles each), the same flops per loop trip --
// and 4 "vector" waves (1 per SIMD) run a loop of independent v_fma_f32 (0, 48 or 96 per MFMA-loop trip's worth of time).
// Prints the kernel time of every combination: the prefill kernel's dequant waves live in exactly this situation.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_port.hip -o /tmp/mfma_port && /tmp/mfma_port
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int VALU>   // SHAPE 0: none, 16: 16x16x32, 32: 32x32x16; VALU: v_fma per trip on the vector waves (0 = idle)
__global__ __launch_bounds__(768) void k(float* out, int trips) {
    const int wave = threadIdx.x >> 6;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    float res = 0.f;
    if (wave < 8) {
        if constexpr (SHAPE == 16) {
            f32x4 acc[16];
            for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
            for (int t = 0; t < trips; ++t) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                asm volatile("" ::: "memory");
            }
            for (int i = 0; i < 16; ++i) res += acc[i][0];
        } else if constexpr (SHAPE == 32) {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
            for (int t = 0; t < trips; ++t) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                asm volatile("" ::: "memory");
            }
            for (int i = 0; i < 4; ++i) res += acc[i][0];
        }
    } else if constexpr (VALU > 0) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
        for (int t = 0; t < trips; ++t) {
#pragma unroll
            for (int i = 0; i < VALU; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], 1.0001f, 0.5f);
            asm volatile("" ::: "memory");
        }
        for (int i = 0; i < 8; ++i) res += v[i];
    }
    if (res == 123.456f) out[threadIdx.x] = res;
}

template <int SHAPE, int VALU>
float run(float* out, int trips) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<SHAPE, VALU><<<256, 768>>>(out, trips);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<SHAPE, VALU><<<256, 768>>>(out, trips);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int trips = 20000;      // per trip and matrix wave: 16 x 16-cycle or 8 x 32-cycle MFMAs = 256 pipe cycles; 2 waves per SIMD
    printf("us per launch (%d trips; matrix pipes alone need 2 x 256 cycles per trip and SIMD)\n", trips);
    printf("                      no VALU     48 fma/trip   96 fma/trip   192 fma/trip\n");
    printf("no MFMA            %9.0f   %9.0f   %9.0f   %9.0f\n", 0.f, run<0, 48>(out, trips), run<0, 96>(out, trips), run<0, 192>(out, trips));
    printf("16x16x32 f16       %9.0f   %9.0f   %9.0f   %9.0f\n", run<16, 0>(out, trips), run<16, 48>(out, trips), run<16, 96>(out, trips), run<16, 192>(out, trips));
    printf("32x32x16 f16       %9.0f   %9.0f   %9.0f   %9.0f\n", run<32, 0>(out, trips), run<32, 48>(out, trips), run<32, 96>(out, trips), run<32, 192>(out, trips));
    return 0;
}
