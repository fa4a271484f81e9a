// SymQuantizer / AsymQuantizer forward: the dynamic-range fake quantisers LLM-QAT applies to
// activations and to the KV cache (reference LLM-QAT/models/utils_quant.py:31-86 and 105-182; call
// sites utils_quant.py:716-724, modeling_llama_quant.py:323-329).  HBM-bound elementwise work.
//
//   symmetric : s = qmax / (max|x| + 1e-6);  out = round(x * s) / (s + 1e-6)          qmax = 2^(bits-1) - 1
//   asymmetric: e = (max - min) + 1e-8;      out = round((x - min) / e * L) / L * e + min   L = 2^bits - 1
// where the range is taken over a SEGMENT of the tensor.  The reference's slicing quirks are kept by the
// host (mxq_amd/utils_quant.py) and reach the kernels as plain geometry:
//   * group kernel: 2-D [rows, cols], groups of 8 / 128 consecutive columns; columns at or beyond
//     `covered` (= cols rounded down to a whole group) get the range 0 the reference leaves there;
//   * row kernel (segments of <= 4096 elements, >= 1024 of them: one wave per segment, one pass) and
//     segment kernels: n_seg contiguous segments of seg_len elements (a token row of a 3-D activation,
//     a (batch, head) slab of a 4-D tensor, or the whole tensor for layerwise=True); segment i is "live"
//     iff i % period < live, the others get range 0 (3-D inputs: the reference slices dim 1 with a group
//     count derived from the last dim, so only the first `live` tokens of every sequence are ranged).
//     Pass 1 reduces ranges into order-preserving integer keys with atomics (any segment length, any
//     segment count fills the chip), pass 2 applies them; the second read comes from L2 / MALL.
// Every reference op is evaluated in fp32 and rounded to the tensor dtype (mxq_fq_types.h), so outputs
// are bit-identical to PyTorch for fp32 / bf16 / fp16.
#include <hip/hip_runtime.h>
#include <math.h>

#include "mxq_fq_types.h"
#include "mxq_kernels.h"

namespace {

using namespace mxq_fq;

// torch.max / torch.min propagate NaN; fmaxf / fminf do not
__device__ __forceinline__ float nan_max(float a, float b) { return (b > a || b != b) ? b : a; }
__device__ __forceinline__ float nan_min(float a, float b) { return (b < a || b != b) ? b : a; }

// x[j] <- x[j] / d for the lane's VEC values, where the quotient is rounded to T right afterwards.  An IEEE fp32
// division is ~10 VALU ops; x * v_rcp_f32(d) is two, and fl32(x * rcp(d)) rounds to the same T value as the correctly
// rounded quotient unless it lies within a few fp32 ulps of one of T's rounding boundaries (mxq_fq_types.h,
// boundary_key: the screen of the weight fake-quant kernel).  Any lane of the wave near a boundary (or outside the
// normal range, or fp32 tensors, whose quotients are not re-rounded) sends the wave through the real division.
template <typename T>
__device__ __forceinline__ void div_vec(float (&x)[T::VEC], float d) {
    if constexpr (T::HAS_FAST_DIV) {
        const float r = __builtin_amdgcn_rcpf(d);
        float q[T::VEC];
        uint32_t key = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < T::VEC; ++j) {
            q[j] = x[j] * r;
            key = min(key, T::boundary_key(q[j]));
        }
        if (!__any(key < T::KEY_LIMIT)) {
#pragma unroll
            for (int j = 0; j < T::VEC; ++j) x[j] = q[j];
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < T::VEC; ++j) x[j] = x[j] / d;
}
template <typename T>
__device__ __forceinline__ void rnd_vec(float (&x)[T::VEC]) {   // roundings go pairwise (T::rnd2: one v_cvt_pk per two values)
#pragma unroll
    for (int j = 0; j < T::VEC; j += 2) T::rnd2(x[j], x[j + 1]);
}

template <typename T>
__device__ __forceinline__ void sym_apply(const float (&v)[T::VEC], float mx, float qmax, float (&o)[T::VEC]) {
    // `qmax / tensor` is Tensor.__rtruediv__ = tensor.reciprocal() * qmax: two roundings, not one division
    const float s = T::rnd(T::rnd(1.0f / T::rnd(mx + 1e-6f)) * qmax);
    const float s2 = T::rnd(s + 1e-6f);
#pragma unroll
    for (int j = 0; j < T::VEC; ++j) o[j] = v[j] * s;
    rnd_vec<T>(o);
#pragma unroll
    for (int j = 0; j < T::VEC; ++j) o[j] = rintf(o[j]);
    div_vec<T>(o, s2);
    rnd_vec<T>(o);
}

template <typename T>
__device__ __forceinline__ void asym_apply(const float (&v)[T::VEC], float mn, float mx, float L, float (&o)[T::VEC]) {
    const float alpha = T::rnd(mx - mn);
    const float e = T::rnd(alpha + 1e-8f);
#pragma unroll
    for (int j = 0; j < T::VEC; ++j) o[j] = v[j] - mn;
    rnd_vec<T>(o);
    div_vec<T>(o, e);
    rnd_vec<T>(o);
#pragma unroll
    for (int j = 0; j < T::VEC; ++j) o[j] *= L;
    rnd_vec<T>(o);
#pragma unroll
    for (int j = 0; j < T::VEC; ++j) o[j] = rintf(o[j]);
    div_vec<T>(o, L);
    rnd_vec<T>(o);
#pragma unroll
    for (int j = 0; j < T::VEC; ++j) o[j] *= e;
    rnd_vec<T>(o);
#pragma unroll
    for (int j = 0; j < T::VEC; ++j) o[j] += mn;
    rnd_vec<T>(o);
}

// ------------------------------------------------------------------------------------------------
// group kernel: one pass, the group's range comes from a xor-shuffle over its g / VEC lanes
// ------------------------------------------------------------------------------------------------
template <typename T, bool SYM>
__global__ __launch_bounds__(256) void mxq_actquant_group_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                                 int64_t rows, int cols, int lpg, int tpr, int covered,
                                                                 float lv) {
    constexpr int VEC = T::VEC;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / tpr;
    const int c = (int)(t - row * tpr) * VEC;
    const bool valid = row < rows && c < cols;
    float v[VEC], o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = 0.f;
    if (valid) T::load(x, row * cols + c, v);
    float mx = SYM ? 0.f : -INFINITY, mn = INFINITY;
    if (valid) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            mx = nan_max(mx, SYM ? fabsf(v[j]) : v[j]);
            if (!SYM) mn = nan_min(mn, v[j]);
        }
    }
    for (int d = 1; d < lpg; d <<= 1) {   // lpg is a power of two <= 32 and tpr % lpg == 0: groups never straddle waves
        mx = nan_max(mx, __shfl_xor(mx, d));
        if (!SYM) mn = nan_min(mn, __shfl_xor(mn, d));
    }
    if (!valid) return;
    if (c >= covered) { mx = 0.f; mn = 0.f; }
    if (SYM) sym_apply<T>(v, mx, lv, o);
    else asym_apply<T>(v, mn, mx, lv, o);
    T::store(out, row * cols + c, o);
}

// ------------------------------------------------------------------------------------------------
// segment kernels
// ------------------------------------------------------------------------------------------------
// order-preserving float -> uint32 key: atomicMax on keys == max on floats, and +NaN beats +inf
__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

constexpr int SEG_UNROLL = 4;   // capi.hip sizes the grid with the same 256 * VEC * 4 elements per workgroup

template <typename T, bool SYM>
__global__ __launch_bounds__(256) void mxq_actquant_range_kernel(const void* __restrict__ x, uint32_t* __restrict__ keys,
                                                                 int64_t seg_len, int chunks) {
    constexpr int VEC = T::VEC;
    const int64_t seg = blockIdx.x / chunks;
    const int chunk = blockIdx.x - (int)(seg * chunks);
    const int64_t base = seg * seg_len;
    float mx = SYM ? 0.f : -INFINITY, mn = INFINITY;
#pragma unroll
    for (int u = 0; u < SEG_UNROLL; ++u) {
        const int64_t e = ((int64_t)(chunk * SEG_UNROLL + u) * 256 + threadIdx.x) * VEC;
        if (e < seg_len) {
            float v[VEC];
            T::load(x, base + e, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                mx = nan_max(mx, SYM ? fabsf(v[j]) : v[j]);
                if (!SYM) mn = nan_min(mn, v[j]);
            }
        }
    }
    for (int d = 1; d < 64; d <<= 1) {
        mx = nan_max(mx, __shfl_xor(mx, d));
        if (!SYM) mn = nan_min(mn, __shfl_xor(mn, d));
    }
    __shared__ float smx[4], smn[4];
    if ((threadIdx.x & 63) == 0) {
        smx[threadIdx.x >> 6] = mx;
        smn[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mx = nan_max(mx, smx[w]);
            mn = nan_min(mn, smn[w]);
        }
        if (mx != mx) mx = __uint_as_float(0x7FC00000u);   // canonical +NaN: the largest key
        atomicMax(keys + 2 * seg, f2key(mx));
        if (!SYM) atomicMax(keys + 2 * seg + 1, ~f2key(mn));   // inverted key: max == float min
    }
}

template <typename T, bool SYM>
__global__ __launch_bounds__(256) void mxq_actquant_apply_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                                 const uint32_t* __restrict__ keys, int64_t seg_len,
                                                                 int chunks, int64_t period, int64_t live, float lv) {
    constexpr int VEC = T::VEC;
    const int64_t seg = blockIdx.x / chunks;
    const int chunk = blockIdx.x - (int)(seg * chunks);
    const int64_t base = seg * seg_len;
    float mx = 0.f, mn = 0.f;
    if (seg % period < live) {
        mx = key2f(keys[2 * seg]);
        if (!SYM) mn = key2f(~keys[2 * seg + 1]);
    }
#pragma unroll
    for (int u = 0; u < SEG_UNROLL; ++u) {
        const int64_t e = ((int64_t)(chunk * SEG_UNROLL + u) * 256 + threadIdx.x) * VEC;
        if (e < seg_len) {
            float v[VEC], o[VEC];
            T::load(x, base + e, v);
            if (SYM) sym_apply<T>(v, mx, lv, o);
            else asym_apply<T>(v, mn, mx, lv, o);
            T::store(out, base + e, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// row kernel: segments of up to NI * 64 * VEC elements (4096 for 16-bit dtypes: a token row at Llama's hidden
// size) stay in registers -- one wave per segment, one HBM read, no scratch, no second launch
// ------------------------------------------------------------------------------------------------
template <typename T, bool SYM, int NI>
__global__ __launch_bounds__(256) void mxq_actquant_row_kernel(const void* __restrict__ x, void* __restrict__ out,
                                                               int64_t n_seg, int seg_len, int64_t period, int64_t live,
                                                               float lv) {
    constexpr int VEC = T::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= n_seg) return;   // wave-uniform
    const int64_t base = seg * seg_len;
    uint4 raw[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e0 = (i * 64 + lane) * VEC;
        raw[i] = make_uint4(0, 0, 0, 0);
        if (e0 < seg_len) raw[i] = T::load_raw(x, base + e0);
    }
    float mx = SYM ? 0.f : -INFINITY, mn = INFINITY;
    if (seg % period < live) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 < seg_len) {
                float v[VEC];
                T::unpack(raw[i], v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    mx = nan_max(mx, SYM ? fabsf(v[j]) : v[j]);
                    if (!SYM) mn = nan_min(mn, v[j]);
                }
            }
        }
        for (int d = 1; d < 64; d <<= 1) {
            mx = nan_max(mx, __shfl_xor(mx, d));
            if (!SYM) mn = nan_min(mn, __shfl_xor(mn, d));
        }
    } else {
        mx = 0.f;
        mn = 0.f;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e0 = (i * 64 + lane) * VEC;
        if (e0 < seg_len) {
            float v[VEC], o[VEC];
            T::unpack(raw[i], v);
            if (SYM) sym_apply<T>(v, mx, lv, o);
            else asym_apply<T>(v, mn, mx, lv, o);
            T::store(out, base + e0, o);
        }
    }
}

template <typename T, bool SYM>
int launch_group(const void* x, void* out, int64_t rows, int cols, int group, float lv, hipStream_t stream) {
    const int lpg = group / T::VEC;
    const int per_row = cols / T::VEC;
    const int tpr = (per_row + lpg - 1) / lpg * lpg;
    const int64_t threads = rows * tpr;
    mxq_actquant_group_kernel<T, SYM><<<(unsigned)((threads + 255) / 256), 256, 0, stream>>>(
        x, out, rows, cols, lpg, tpr, cols / group * group, lv);
    return (int)hipGetLastError();
}

template <typename T, bool SYM>
int launch_seg(const void* x, void* out, void* range_ws, int64_t n_seg, int64_t seg_len, int64_t period, int64_t live,
               float lv, hipStream_t stream) {
    if (seg_len <= 8 * 64 * T::VEC && n_seg >= 1024) {   // enough rows to fill the chip with one wave each
        mxq_actquant_row_kernel<T, SYM, 8><<<(unsigned)((n_seg + 3) / 4), 256, 0, stream>>>(x, out, n_seg, (int)seg_len,
                                                                                          period, live, lv);
        return (int)hipGetLastError();
    }
    const int64_t per_chunk = (int64_t)256 * T::VEC * SEG_UNROLL;
    const int chunks = (int)((seg_len + per_chunk - 1) / per_chunk);
    hipError_t e = hipMemsetAsync(range_ws, 0, (size_t)n_seg * 8, stream);   // key 0 is below every float's key
    if (e != hipSuccess) return (int)e;
    const unsigned grid = (unsigned)(n_seg * chunks);
    mxq_actquant_range_kernel<T, SYM><<<grid, 256, 0, stream>>>(x, (uint32_t*)range_ws, seg_len, chunks);
    mxq_actquant_apply_kernel<T, SYM><<<grid, 256, 0, stream>>>(x, out, (const uint32_t*)range_ws, seg_len, chunks,
                                                                period, live, lv);
    return (int)hipGetLastError();
}

}   // namespace

int mxq_launch_actquant_group(const void* x, void* out, int64_t rows, int cols, int group, int num_bits, int symmetric,
                              int dtype, hipStream_t stream) {
    const float lv = symmetric ? (float)((1ll << (num_bits - 1)) - 1) : (float)((1ll << num_bits) - 1);
    switch (dtype * 2 + (symmetric ? 1 : 0)) {
        case MXQ_DTYPE_F32 * 2 + 1: return launch_group<F32, true>(x, out, rows, cols, group, lv, stream);
        case MXQ_DTYPE_F32 * 2: return launch_group<F32, false>(x, out, rows, cols, group, lv, stream);
        case MXQ_DTYPE_F16 * 2 + 1: return launch_group<F16, true>(x, out, rows, cols, group, lv, stream);
        case MXQ_DTYPE_F16 * 2: return launch_group<F16, false>(x, out, rows, cols, group, lv, stream);
        case MXQ_DTYPE_BF16 * 2 + 1: return launch_group<BF16, true>(x, out, rows, cols, group, lv, stream);
        case MXQ_DTYPE_BF16 * 2: return launch_group<BF16, false>(x, out, rows, cols, group, lv, stream);
    }
    return (int)hipErrorInvalidValue;
}

int mxq_launch_actquant_seg(const void* x, void* out, void* range_ws, int64_t n_seg, int64_t seg_len, int64_t period,
                            int64_t live, int num_bits, int symmetric, int dtype, hipStream_t stream) {
    const float lv = symmetric ? (float)((1ll << (num_bits - 1)) - 1) : (float)((1ll << num_bits) - 1);
    switch (dtype * 2 + (symmetric ? 1 : 0)) {
        case MXQ_DTYPE_F32 * 2 + 1: return launch_seg<F32, true>(x, out, range_ws, n_seg, seg_len, period, live, lv, stream);
        case MXQ_DTYPE_F32 * 2: return launch_seg<F32, false>(x, out, range_ws, n_seg, seg_len, period, live, lv, stream);
        case MXQ_DTYPE_F16 * 2 + 1: return launch_seg<F16, true>(x, out, range_ws, n_seg, seg_len, period, live, lv, stream);
        case MXQ_DTYPE_F16 * 2: return launch_seg<F16, false>(x, out, range_ws, n_seg, seg_len, period, live, lv, stream);
        case MXQ_DTYPE_BF16 * 2 + 1: return launch_seg<BF16, true>(x, out, range_ws, n_seg, seg_len, period, live, lv, stream);
        case MXQ_DTYPE_BF16 * 2: return launch_seg<BF16, false>(x, out, range_ws, n_seg, seg_len, period, live, lv, stream);
    }
    return (int)hipErrorInvalidValue;
}
