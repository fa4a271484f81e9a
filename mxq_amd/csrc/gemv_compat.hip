// GEMV entry points that accept the OPERAND FORMATS of the reference's CUDA prototype, so
// that its two stand-alone scripts (mxq_quant/cuda_kernel/test_correct_gemv.py,
// test_mxq_gemv.py) run unmodified against `mxq_inference_engine` on ROCm:
//
//   gemv_forward_cuda      uniform 4-bit, group 32/64/128
//                          (csrc/quantization/gemv_cuda.cu:45-242, launcher :346-399)
//   gemv_mxq_forward_cuda  the "MXQ 2.8-bit" format, IC = 4096
//                          (csrc/quantization/gemv_mxq_cuda.cu:39-208, launcher :225-273)
//
// These are written for wave64 from scratch: one wave per output channel, 4 channels per
// workgroup.  For the MXQ prototype format the reference's two 32-lane "iterations" are
// simply the two halves of a wave64 (lane = 32*it + t), and each half reads its OWN
// activation columns 2048*it + 64*t .. +63 -- the reference kernel's missing iteration
// offset (gemv_mxq_cuda.cu:119) is a latent bug that is not reproduced.
// fp32 dequant + fp32 accumulate exactly as the reference does; fp16 output.
#include <hip/hip_runtime.h>

#include "mxq_kernels.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

__device__ __forceinline__ void unpack8h(const uint4 a, float f[8]) {
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = h2f((uint16_t)(w[i] & 0xFFFFu));
        f[2 * i + 1] = h2f((uint16_t)(w[i] >> 16));
    }
}

// ---- uniform 4-bit: kernel [OC, IC/8] int32, nibble j of a word = element j ------------
// A lane takes 16 bytes of codes (4 words = 32 weights = one or a part of one group: group sizes are multiples of
// 32) per iteration and BT batch rows against them, so the weights are read once per BT rows (the reference puts the
// batch on gridDim.z and re-reads them per row, gemv_cuda.cu:346-399) with 16-byte loads (round 2 began with 4-byte
// loads and one batch row per workgroup: 8.7 us at 4096^2).
template <int BT>
__global__ __launch_bounds__(256) void gemv_awq_kernel(const uint16_t* __restrict__ x,
                                                       const uint32_t* __restrict__ kernel,
                                                       const uint16_t* __restrict__ scales,
                                                       const uint32_t* __restrict__ zeros, uint16_t* __restrict__ y,
                                                       int B, int IC, int OC, int group_size, int zeros_w, int sf_w) {
    const int lane = threadIdx.x & 63;
    const int oc = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b0 = blockIdx.y * BT;
    if (oc >= OC) return;
    const int words = IC / 8;
    float psum[BT];
#pragma unroll
    for (int bb = 0; bb < BT; ++bb) psum[bb] = 0.f;
    for (int wi = lane * 4; wi < words; wi += 256) {
        const uint4 wq4 = *(const uint4*)(kernel + (int64_t)oc * words + wi);
        const uint32_t wq[4] = {wq4.x, wq4.y, wq4.z, wq4.w};
        const int g = (wi * 8) / group_size;
        const float s = h2f(scales[(int64_t)oc * sf_w + g]);
        const float z = (float)((zeros[(int64_t)oc * zeros_w + (g >> 3)] >> ((g & 7) * 4)) & 0xFu);
        float dq[32];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int j = 0; j < 8; ++j) dq[w * 8 + j] = s * ((float)((wq[w] >> (4 * j)) & 0xFu) - z);
#pragma unroll
        for (int bb = 0; bb < BT; ++bb) {
            if (b0 + bb >= B) break;
            const uint16_t* xb = x + (int64_t)(b0 + bb) * IC + wi * 8;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float xf[8];
                unpack8h(*(const uint4*)(xb + w * 8), xf);
#pragma unroll
                for (int j = 0; j < 8; ++j) psum[bb] += dq[w * 8 + j] * xf[j];
            }
        }
    }
#pragma unroll
    for (int bb = 0; bb < BT; ++bb) {
        const float v = wave_sum(psum[bb]);
        if (lane == 0 && b0 + bb < B) {
            const _Float16 h = (_Float16)v;
            y[(int64_t)(b0 + bb) * OC + oc] = __builtin_bit_cast(uint16_t, h);
        }
    }
}

// ---- MXQ prototype format, IC == 4096 -------------------------
template <int BT>
__global__ __launch_bounds__(256) void gemv_proto_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ weight, const uint32_t* __restrict__ weight_last,
    const uint32_t* __restrict__ zeros_and_scales, const uint16_t* __restrict__ scales_2nd,
    const uint32_t* __restrict__ zeros_2nd, const uint16_t* __restrict__ scales_4b,
    const uint32_t* __restrict__ zeros_4b, uint16_t* __restrict__ y, int B, int IC, int OC) {
    const int lane = threadIdx.x & 63, t = lane & 31, it = lane >> 5;
    const int oc = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b0 = blockIdx.y * BT;
    if (oc >= OC) return;
    const int weight_w = IC / 64 * 4, last_w = IC / 64;
    const uint4 pw = *(const uint4*)(weight + (int64_t)oc * weight_w + it * (weight_w / 2) + t * 4);
    const uint32_t pl = weight_last[(int64_t)oc * last_w + it * (last_w / 2) + t];
    const uint32_t zs = zeros_and_scales[(int64_t)oc * 32 + t];
    const uint32_t z1p = (zs >> (16 * it)) & 0xFFu, s1p = (zs >> (16 * it + 8)) & 0xFFu;
    const uint32_t z2p = (zeros_2nd[(int64_t)(oc / 4) * 32 + t] >> (8 * it)) & 0xFFu;
    const float s4 = h2f(scales_4b[oc]);
    const float z4 = (float)((zeros_4b[oc / 8] >> ((oc % 8) * 4)) & 0xFu);
    const uint32_t w2[3] = {pw.x, pw.y, pw.z};
    // the lane's 64 dequantised weights, once for all BT batch rows (the reference re-reads and re-converts them per row:
    // batch on gridDim.z, gemv_mxq_cuda.cu:261-262)
    float dq[64];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float z1 = (float)((z1p >> (2 * g)) & 3u), s1 = (float)((s1p >> (2 * g)) & 3u);
        const float z2 = (float)((z2p >> (2 * g)) & 3u);
        const float sf = h2f(scales_2nd[(int64_t)(oc / 4) * 192 + it * 96 + t * 3 + g]) * (s1 - z2);
#pragma unroll
        for (int j = 0; j < 16; ++j) dq[g * 16 + j] = sf * ((float)((w2[g] >> (2 * j)) & 3u) - z1);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dq[48 + j] = s4 * ((float)((pw.w >> (4 * j)) & 0xFu) - z4);
        dq[56 + j] = s4 * ((float)((pl >> (4 * j)) & 0xFu) - z4);
    }
#pragma unroll
    for (int bb = 0; bb < BT; ++bb) {
        if (b0 + bb >= B) break;
        const uint16_t* xl = x + (int64_t)(b0 + bb) * IC + 2048 * it + 64 * t;   // this lane's 64 columns
        float psum = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            float xf[8];
            unpack8h(*(const uint4*)(xl + h * 8), xf);
#pragma unroll
            for (int j = 0; j < 8; ++j) psum += dq[h * 8 + j] * xf[j];
        }
        psum = wave_sum(psum);
        if (lane == 0) {
            const _Float16 hh = (_Float16)psum;
            y[(int64_t)(b0 + bb) * OC + oc] = __builtin_bit_cast(uint16_t, hh);
        }
    }
}

}   // namespace

int mxq_launch_gemv_awq_f16(const void* x, const void* kernel, const void* scales, const void* zeros, void* y,
                            int B, int IC, int OC, int group_size, hipStream_t stream) {
    const int packed = (IC / group_size + 7) / 8;
    const int zeros_w = (packed + 3) / 4 * 4;   // gemv_cuda.cu:54-59
    const int sf_w = zeros_w * 8;
    if (B <= 1) {
        gemv_awq_kernel<1><<<dim3((OC + 3) / 4, B), 256, 0, stream>>>(
            (const uint16_t*)x, (const uint32_t*)kernel, (const uint16_t*)scales, (const uint32_t*)zeros, (uint16_t*)y, B,
            IC, OC, group_size, zeros_w, sf_w);
    } else {
        gemv_awq_kernel<4><<<dim3((OC + 3) / 4, (B + 3) / 4), 256, 0, stream>>>(
            (const uint16_t*)x, (const uint32_t*)kernel, (const uint16_t*)scales, (const uint32_t*)zeros, (uint16_t*)y, B,
            IC, OC, group_size, zeros_w, sf_w);
    }
    return (int)hipGetLastError();
}

int mxq_launch_gemv_proto_f16(const void* x, const void* weight, const void* weight_last,
                              const void* zeros_and_scales, const void* scales_2nd, const void* zeros_2nd,
                              const void* scales_4b, const void* zeros_4b, void* y, int B, int IC, int OC,
                              hipStream_t stream) {
    if (B <= 1) {
        gemv_proto_kernel<1><<<dim3((OC + 3) / 4, B), 256, 0, stream>>>(
            (const uint16_t*)x, (const uint32_t*)weight, (const uint32_t*)weight_last, (const uint32_t*)zeros_and_scales,
            (const uint16_t*)scales_2nd, (const uint32_t*)zeros_2nd, (const uint16_t*)scales_4b,
            (const uint32_t*)zeros_4b, (uint16_t*)y, B, IC, OC);
    } else {
        gemv_proto_kernel<4><<<dim3((OC + 3) / 4, (B + 3) / 4), 256, 0, stream>>>(
            (const uint16_t*)x, (const uint32_t*)weight, (const uint32_t*)weight_last, (const uint32_t*)zeros_and_scales,
            (const uint16_t*)scales_2nd, (const uint32_t*)zeros_2nd, (const uint16_t*)scales_4b,
            (const uint32_t*)zeros_4b, (uint16_t*)y, B, IC, OC);
    }
    return (int)hipGetLastError();
}
