// Fused MXAsymQuantizer forward and STE backward for QAT
// (reference LLM-QAT/models/utils_quant.py:316-462 forward, :464-475 backward; the live
// branch only: 2-D input, layerwise=False, in_features % 64 == 0 -- SURVEY.md H6).
//
// The reference runs ~1.7k-4.5k tiny torch kernels per weight (a Python loop over 64-wide
// chunks x 3 groups).  Here: ONE kernel, one wave per weight row, HBM-bound
// (read + write = 2 * sizeof(T) bytes per element):
//   pass 1  row min/max of the gathered 4-bit arm (last 16 of every 64 columns, an fp32
//           buffer in the reference, :347,369-377) -- only those 32/64-B quarters are read;
//   pass 2  coalesced 16-B-per-lane sweep of the row: 2-bit groups (16 columns) live in 2
//           (16-bit dtypes) or 4 (fp32) adjacent lanes -> xor-shuffle min/max; every
//           reference op is done in fp32 and rounded to the tensor dtype T, which is what
//           PyTorch does for bf16/fp16 tensors -> bit-identical output.
// The second read of the row hits L2 (a row is 8-44 KB).  For 16-bit dtypes and rows of up to
// 4096 columns a register-resident single-pass variant is used instead.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "mxq_fq_types.h"
#include "mxq_kernels.h"

namespace {

using namespace mxq_fq;

// Shared per-group math of pass 2: v[] (VEC values of one lane) -> o[].  mn/mx are the group
// min/max (already reduced over the lanes of the group).
template <typename T, bool FASTQ>
__device__ __forceinline__ void quant_vec(const float (&v)[T::VEC], float alpha, float beta, float L,
                                          float (&o)[T::VEC]) {
    constexpr int VEC = T::VEC;
    const float invL = 1.0f / L;
    const float e = T::rnd(alpha + 1e-8f);
    float xs[VEC], xn[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j += 2) {   // roundings go pairwise (T::rnd2)
        xs[j] = v[j] - beta;
        xs[j + 1] = v[j + 1] - beta;
        T::rnd2(xs[j], xs[j + 1]);
    }
    // (w - beta) / e must be the correctly rounded fp32 quotient re-rounded to T.  Fast path:
    // multiply by v_rcp_f32(e) and screen; any lane near a rounding boundary (or out of the
    // normal range) sends the wave through the IEEE divide (~5 % of the iterations).
    bool slow = !T::HAS_FAST_DIV;
    if constexpr (T::HAS_FAST_DIV) {
        const float r = __builtin_amdgcn_rcpf(e);
        uint32_t key = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            xn[j] = xs[j] * r;
            key = min(key, T::boundary_key(xn[j]));
        }
        slow = __any(key < T::KEY_LIMIT);
    }
    if (slow) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) xn[j] = xs[j] / e;
    }
#pragma unroll
    for (int j = 0; j < VEC; j += 2) {
        float a = xn[j], b = xn[j + 1];
        T::rnd2(a, b);
        a *= L; b *= L;
        T::rnd2(a, b);
        a = rintf(a); b = rintf(b);
        a = FASTQ ? a * invL : a / L;
        b = FASTQ ? b * invL : b / L;
        T::rnd2(a, b);
        a *= e; b *= e;
        T::rnd2(a, b);
        a += beta; b += beta;
        T::rnd2(a, b);
        o[j] = a;
        o[j + 1] = b;
    }
}

// Register-resident variant (16-bit dtypes): a row is split over WPR waves of the workgroup (1, 2 or 4; each
// part a multiple of 64 columns and at most NI * 64 * VEC = 4096 elements), every part is loaded ONCE with all
// its 16-B loads in flight, the 4-bit-arm min/max comes from registers (combined across the row's waves
// through LDS when WPR > 1) and so do the group min/max; then everything is quantised and stored.  HBM sees
// exactly one read and one write per element and no second pass exists.
// (Round 2 tried to overlap a wave's arithmetic with the next row's loads -- a second register set, then an LDS-DMA
// prefetch: both slower, profiles/r02_fakequant_pipe_ab.txt.  At ~30 VALU ops per element the kernel is co-limited
// by VALU issue and HBM, and occupancy is worth more than the prefetch.)
template <typename T, bool FASTQ, int NI, int WPR>
__global__ __launch_bounds__(256) void mxq_fakequant_fwd_reg_kernel(const void* __restrict__ w,
                                                                    void* __restrict__ out, int rows, int cols,
                                                                    float L2) {
    constexpr int VEC = T::VEC;
    constexpr int LPG = 16 / VEC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * (4 / WPR) + wave / WPR;
    const bool active = row < rows;   // wave-uniform; inactive waves still reach the barrier below
    const int part = cols / WPR;
    const int64_t base = (int64_t)row * cols + (wave % WPR) * part;
    uint4 raw[NI];   // the part stays in its storage format: 4 VGPRs per 16-B load
    const bool is4 = ((lane * VEC) & 63) >= 48;   // 64 * VEC elements per iteration keep the chunk phase
    float mn4 = INFINITY, mx4 = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e0 = (i * 64 + lane) * VEC;
        raw[i] = make_uint4(0, 0, 0, 0);
        if (active && e0 < part) raw[i] = T::load_raw_nt(w, base + e0);
    }
    // min / max of every 16-column group (kept for the quantisation below); the 4-bit arm's row-wide min / max is the
    // min / max over the groups of the is4 lanes -- no separate pass over the row
    float gmn[NI], gmx[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e0 = (i * 64 + lane) * VEC;
        float t[VEC];
        T::unpack(raw[i], t);
        float mn = t[0], mx = t[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) { mn = fminf(mn, t[j]); mx = fmaxf(mx, t[j]); }
#pragma unroll
        for (int o = 1; o < LPG; o <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, o, 64));
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        }
        gmn[i] = mn;
        gmx[i] = mx;
        const bool in4 = active && e0 < part && is4;
        mn4 = fminf(mn4, in4 ? mn : INFINITY);
        mx4 = fmaxf(mx4, in4 ? mx : -INFINITY);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        mn4 = fminf(mn4, __shfl_xor(mn4, o, 64));
        mx4 = fmaxf(mx4, __shfl_xor(mx4, o, 64));
    }
    if constexpr (WPR > 1) {
        __shared__ float part_mn[4], part_mx[4];
        if (lane == 0) {
            part_mn[wave] = mn4;
            part_mx[wave] = mx4;
        }
        __syncthreads();
        const int w0 = wave / WPR * WPR;
#pragma unroll
        for (int k = 0; k < WPR; ++k) {
            mn4 = fminf(mn4, part_mn[w0 + k]);
            mx4 = fmaxf(mx4, part_mx[w0 + k]);
        }
    }
    if (!active) return;
    const float alpha4 = T::rnd(mx4 - mn4);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e0 = (i * 64 + lane) * VEC;
        if (e0 < part) {   // a 16-column group is never split by the part end (part % 64 == 0)
            float vi[VEC];
            T::unpack(raw[i], vi);
            const float mn = gmn[i], mx = gmx[i];
            float o[VEC];
            quant_vec<T, FASTQ>(vi, is4 ? alpha4 : T::rnd(mx - mn), is4 ? mn4 : mn, is4 ? 15.0f : L2, o);
            T::store_nt(out, base + e0, o);
        }
    }
}

// FASTQ: q / L may be computed as q * (1/L) -- the launcher has checked on the host that this
// rounds to the same T value for every integer 0 <= q <= L of both arms.
template <typename T, bool FASTQ>
__global__ __launch_bounds__(256) void mxq_fakequant_fwd_kernel(const void* __restrict__ w, void* __restrict__ out,
                                                                int rows, int cols, float L2) {
    constexpr int VEC = T::VEC;
    constexpr int LPG = 16 / VEC;   // lanes per 16-column group (2 or 4)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;   // wave-uniform
    const int64_t base = (int64_t)row * cols;
    const int NC = cols / 64;
    float v[VEC];

    // pass 1: min / max over the gathered 4-bit slice of the row
    float mn4 = INFINITY, mx4 = -INFINITY;
    for (int c = lane / LPG; c < NC; c += 64 / LPG) {
        T::load(w, base + (int64_t)c * 64 + 48 + (lane % LPG) * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { mn4 = fminf(mn4, v[j]); mx4 = fmaxf(mx4, v[j]); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        mn4 = fminf(mn4, __shfl_xor(mn4, o, 64));
        mx4 = fmaxf(mx4, __shfl_xor(mx4, o, 64));
    }
    const float alpha4 = T::rnd(mx4 - mn4);   // fp32 subtraction, cast on assignment (:369-384)

    // pass 2
    for (int e0 = lane * VEC; e0 < cols; e0 += 64 * VEC) {
        T::load(w, base + e0, v);
        float mn = v[0], mx = v[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) { mn = fminf(mn, v[j]); mx = fmaxf(mx, v[j]); }
#pragma unroll
        for (int o = 1; o < LPG; o <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, o, 64));
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        }
        const bool is4 = (e0 & 63) >= 48;
        const float alpha = is4 ? alpha4 : T::rnd(mx - mn);
        const float beta = is4 ? mn4 : mn;
        float o[VEC];
        quant_vec<T, FASTQ>(v, alpha, beta, is4 ? 15.0f : L2, o);
        T::store_nt(out, base + e0, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mxq_fakequant_bwd_kernel(const void* __restrict__ gout,
                                                                const void* __restrict__ w, void* __restrict__ gin,
                                                                int64_t nvec, float lo, float hi) {
    constexpr int VEC = T::VEC;
    float g[VEC], x[VEC];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        T::load_nt(gout, i * VEC, g);
        T::load_nt(w, i * VEC, x);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (x[j] >= hi || x[j] <= lo) g[j] = 0.0f;
        T::store_nt(gin, i * VEC, g);
    }
}

// host copies of the roundings, used only to validate the q * (1/L) shortcut
inline float host_rnd(float x, int dtype) {
    if (dtype == MXQ_DTYPE_F16) return (float)(_Float16)x;
    if (dtype == MXQ_DTYPE_BF16) {
        uint32_t u;
        memcpy(&u, &x, 4);
        u += 0x7FFFu + ((u >> 16) & 1u);
        u &= 0xFFFF0000u;
        memcpy(&x, &u, 4);
    }
    return x;
}
inline bool mul_matches_div(float L, int dtype) {
    if (dtype == MXQ_DTYPE_F32 || L > 4096.0f) return false;
    volatile float inv = 1.0f / L;
    for (float q = 0.0f; q <= L; q += 1.0f) {
        volatile float a = q / L, b = q * inv;
        if (host_rnd(a, dtype) != host_rnd(b, dtype)) return false;
    }
    return true;
}

template <typename T>
int launch_fwd(const void* w, void* out, int rows, int cols, int num_bits, int dtype, hipStream_t stream) {
    // the reference stores s = 2**num_bits - 1 in a tensor of the weight dtype (utils_quant.py:342,366)
    const float L2 = host_rnd((float)(exp2((double)num_bits) - 1.0), dtype);
    const bool fastq = mul_matches_div(L2, dtype) && mul_matches_div(15.0f, dtype);
    const dim3 grid((rows + 3) / 4);
    // register-resident whenever a row splits into 1, 2 or 4 parts of whole 64-column chunks of <= 4096 elements
    // (a single-wave 24-iteration instance for rows up to 12288 columns was measured slower than the two-pass
    // kernel: hipcc keeps the unpacked row live, 231 VGPRs, 2 waves/SIMD)
    if (fastq && T::VEC == 8 && cols <= 4096)
        mxq_fakequant_fwd_reg_kernel<T, true, 8, 1><<<grid, 256, 0, stream>>>(w, out, rows, cols, L2);
    else if (fastq && T::VEC == 8 && cols <= 8192 && cols % 128 == 0)
        mxq_fakequant_fwd_reg_kernel<T, true, 8, 2><<<(rows + 1) / 2, 256, 0, stream>>>(w, out, rows, cols, L2);
    else if (fastq && T::VEC == 8 && cols <= 16384 && cols % 256 == 0)
        mxq_fakequant_fwd_reg_kernel<T, true, 8, 4><<<rows, 256, 0, stream>>>(w, out, rows, cols, L2);
    else if (fastq) mxq_fakequant_fwd_kernel<T, true><<<grid, 256, 0, stream>>>(w, out, rows, cols, L2);
    else mxq_fakequant_fwd_kernel<T, false><<<grid, 256, 0, stream>>>(w, out, rows, cols, L2);
    return (int)hipGetLastError();
}

template <typename T>
int launch_bwd(const void* gout, const void* w, void* gin, int64_t n, float lo, float hi, hipStream_t stream) {
    const int64_t nvec = n / T::VEC;
    int64_t blocks = (nvec + 255) / 256;
#ifndef MXQ_FQ_BWD_CAP
#define MXQ_FQ_BWD_CAP (256 * 64)
#endif
    // cap + grid-stride.  Round 6: the cap was 2048 workgroups (8 per CU); a plain copy of the same bytes with the same 16-byte
    // nt accesses runs at 4.6 TB/s from 2048 workgroups, 6.0-6.5 from 8192 and 6.7 from 16384 (tools/probes/copy_probe.hip,
    // profiles/r06_copy_probe.txt): short-lived workgroups, each touching a few 4-KiB lines spread over the whole tensor, keep
    // more HBM channels busy than few long-lived ones walking with a large stride
    if (blocks > MXQ_FQ_BWD_CAP) blocks = MXQ_FQ_BWD_CAP;
    if (blocks < 1) blocks = 1;
    mxq_fakequant_bwd_kernel<T><<<(unsigned)blocks, 256, 0, stream>>>(gout, w, gin, nvec, lo, hi);
    return (int)hipGetLastError();
}

}   // namespace

int mxq_launch_fakequant_fwd(const void* w, void* out, int rows, int cols, int num_bits, int dtype,
                             hipStream_t stream) {
    switch (dtype) {
        case MXQ_DTYPE_F32: return launch_fwd<F32>(w, out, rows, cols, num_bits, dtype, stream);
        case MXQ_DTYPE_F16: return launch_fwd<F16>(w, out, rows, cols, num_bits, dtype, stream);
        case MXQ_DTYPE_BF16: return launch_fwd<BF16>(w, out, rows, cols, num_bits, dtype, stream);
    }
    return (int)hipErrorInvalidValue;
}

int mxq_launch_fakequant_bwd(const void* grad_out, const void* w, void* grad_in, int64_t n, float lo, float hi,
                             int dtype, hipStream_t stream) {
    switch (dtype) {
        case MXQ_DTYPE_F32: return launch_bwd<F32>(grad_out, w, grad_in, n, lo, hi, stream);
        case MXQ_DTYPE_F16: return launch_bwd<F16>(grad_out, w, grad_in, n, lo, hi, stream);
        case MXQ_DTYPE_BF16: return launch_bwd<BF16>(grad_out, w, grad_in, n, lo, hi, stream);
    }
    return (int)hipErrorInvalidValue;
}
