// gemm_forward_cuda's operand format (gemm_cuda.h:3-4; gemm8a.hip's header) for FEW tokens (<= 32): a streaming kernel instead
// of MFMA tiles + a combine launch.  Round 6: at 16 tokens x 4096^2 the 64-token tiles cut into K slices (gemm8aq.hip) took
// 15.0-15.7 us -- 11 in the main launch, 4.9 in the combine launch -- where torch's fp16 GEMM on a dense weight takes 9.4; this
// kernel: 11.0 (4096 -> 11008: 23.5 -> 18.2; 11008 -> 4096: 21.5 -> 15.6, torch 20.4).  From 48 tokens on the tiles win again
// (same-process sweep, profiles/r06_awq_gemm_bench.txt), so the dispatch hands it up to 32.
//
// One workgroup = 128 output channels x one K SLICE (S slices per channel block, so that blocks x S fills the chip):
//   * the slice's code words -- rows of 64 bytes, K-major -- go into LDS in one go: every wave DMAs the 32-row groups IT will
//     convert (two 1-KiB LDS-DMAs per group, whole 64-byte row segments), waits for its own DMAs and never meets a barrier before
//     the end; inside a group the 16-byte chunks are placed so that the four k octets a wave instruction reads sit in four
//     different bank groups;
//   * lane (r = lane & 15, q = lane >> 4) builds EXACTLY the MFMA A-operand it owns -- channel 16 b + r, k = 8 q .. 8 q + 7 of a
//     32-row group -- from 8 words: one v_perm_b32 puts the byte that holds the channel's nibble in rows k and k + 1 side by side,
//     (xx >> 4 hi) & 0x000f000f | 0x6400 6400 reads (1024 + q_k, 1024 + q_k+1), minus (1024 + z), times the scale: the reference's
//     fp16((q - z) * s), one rounding (gemm_cuda_gen.cu:134-141) -- 21 vector-ALU ops per fragment of 16 x 32 weights;
//   * the B operand (8 consecutive activations of token lane & 15) comes straight from global memory / L2, one 16-byte load per
//     lane and token block (x is at most 64 x IC fp16);
//   * v_mfma_f32_16x16x32_f16, D^T = W . x^T: a lane ends with 4 consecutive channels of one token, per channel block of 16 and
//     token block of 16; the four waves' partial sums meet in LDS;
//   * S > 1: the workgroup parks its fp32 partial [128 channels x tokens] in the workspace (write-through), counts itself on
//     the channel block's counter, and the LAST arriver -- nobody waits -- adds the S partials in slice order and writes fp16 y,
//     then re-zeroes the counter.  Deterministic: the order of every sum is fixed.
#include <hip/hip_runtime.h>

#include "mxq_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int BN = 128, NB = BN / 16;
constexpr int CNT_BYTES = 64 * 1024;     // the workspace's counter head (shared with the stream-K kernels: zero when idle)

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t m, uint32_t o) {   // (a & m) | o: mask in an SGPR, constant in a VGPR
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m), "v"(o));
    return r;
}

// byte of a code word that holds channel e of its octet, and whether it is the byte's high nibble: channel e sits in nibble
// (0, 4, 1, 5, 2, 6, 3, 7)[e] (dequantize.cuh:35-51)
__device__ __forceinline__ int chan_byte(int e) { return ((e & 1) << 1) | (e >> 2); }
__device__ __forceinline__ int chan_high(int e) { return (e >> 1) & 1; }

// WAVES = 8 (two waves per SIMD: twice the conversion rate of 4, and each wave owns ONE channel block of the result); the kernel
// is written for 4 as well (more token blocks per wave need the registers and the LDS) but from 48 tokens on the tile kernels win
template <int NJ, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mxq_skinny_awq_f16_kernel(const uint16_t* __restrict__ x,
                                                                     const uint32_t* __restrict__ kernel,
                                                                     const uint16_t* __restrict__ scales,
                                                                     const uint32_t* __restrict__ zeros,
                                                                     uint16_t* __restrict__ y, int M, int IC, int OC,
                                                                     uint32_t gmul, int ngroups, int S, int rows_per_slice,
                                                                     int* __restrict__ cnt, float* __restrict__ part) {
    constexpr int BPW = NB / WAVES;                              // channel blocks of the result a wave sums and owns: 2 | 1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int blk = blockIdx.x / S, sl = blockIdx.x - blk * S;
    const int n0 = blk * BN, OC8 = OC >> 3;
    const uint32_t row_bytes = (uint32_t)OC >> 1;
    const int k0 = sl * rows_per_slice;
    const int rows = min(rows_per_slice, IC - k0);              // a multiple of 64 (launcher)
    const int groups = rows >> 5;                               // 32-row groups of this slice: wave w converts groups w, w + 4, ...

    // ---- the wave's groups of code words into LDS: chunk (row 8 qq + i, column quad c) of a group at position (i 16 + c 4 + qq) * 16
    const rsrc_t rs_q = make_rsrc(kernel, (uint32_t)IC * row_bytes);
    {
        uint32_t voff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int P = j * 64 + lane, i = P >> 4, c = (P >> 2) & 3, qq = P & 3;
            int o0 = (n0 >> 3) + 4 * c;
            o0 = o0 < OC8 - 4 ? o0 : OC8 - 4;                   // (a quad past the last channel re-reads the last one: never stored)
            voff[j] = (uint32_t)(8 * qq + i) * row_bytes + (uint32_t)o0 * 4u;
        }
        for (int g = wave; g < groups; g += WAVES) {
            const uint32_t so = (uint32_t)(k0 + g * 32) * row_bytes;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, (__attribute__((address_space(3))) void*)(smem + g * 2048 + j * 1024), 16,
                                                         voff[j], so, 0, 0);
        }
    }

    // ---- per-lane constants of the conversion: which byte / nibble of a word holds channel r & 7 of its octet
    const int e = r & 7;
    const uint32_t selx = 0x0c040c00u + (uint32_t)chan_byte(e) * 0x00010001u;     // [lo.byte, 0, hi.byte, 0]
    const uint32_t selz = 0x0c000c00u + (uint32_t)chan_byte(e) * 0x00010001u;     // the same byte of ONE word twice
    const uint32_t sh4 = 4u * (uint32_t)chan_high(e);
    uint32_t magic;
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(magic));
    constexpr uint32_t LO = 0x000f000fu;
    // the lane's word inside a group's 2 KiB: row 8 q + i, octet 2 b + (r >> 3) -> chunk (i, c = b >> 1, q), word ((2 b) & 3) + (r >> 3)
    const uint32_t rd0 = (uint32_t)(q * 16 + (r >> 3) * 4);
    const rsrc_t rs_s = make_rsrc(scales, (uint32_t)ngroups * (uint32_t)OC * 2u);    // whole tensors [IC / G, OC] and [IC / G, OC / 8]
    const rsrc_t rs_z = make_rsrc(zeros, (uint32_t)ngroups * (uint32_t)OC8 * 4u);
    // the lane's B-operand rows: token 16 j + r (clamped: rows beyond M compute garbage nobody stores)
    const uint16_t* xrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int m = j * 16 + r;
        m = m < M ? m : M - 1;
        xrow[j] = x + (int64_t)m * IC + k0 + q * 8;
    }

    f32x4 acc[NB][NJ];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    half8 xb[NJ];
    if (wave < groups) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) xb[j] = *(const half8*)(xrow[j] + wave * 32);
    }
    // scale / zero-point of every channel block for one quantisation group (wave-uniform per 32-row group: G >= 32); the raw
    // words are fetched BEFORE the wait for the DMAs (first group) or a group ahead, and turned into packed constants when used
    uint32_t zw[NB], sw[NB];
    auto fetch_zs = [&](uint32_t qg) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int oct = (n0 >> 3) + 2 * b + (r >> 3);
            oct = oct < OC8 ? oct : OC8 - 1;
            zw[b] = __builtin_amdgcn_raw_buffer_load_b32(rs_z, (qg * (uint32_t)OC8 + (uint32_t)oct) * 4u, 0, 0);
            sw[b] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs_s, (qg * (uint32_t)OC + (uint32_t)oct * 8u + (uint32_t)e) * 2u, 0, 0);
        }
    };
    uint32_t gprev = __builtin_amdgcn_readfirstlane(__umulhi((uint32_t)(k0 + (wave < groups ? wave : 0) * 32), gmul));
    fetch_zs(gprev);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's own DMAs, its first x fragments and group constants have landed
    h2 zc[NB], sp[NB];
    bool fresh = true;
    for (int g = wave; g < groups; g += WAVES) {
        const uint32_t qg = __builtin_amdgcn_readfirstlane(__umulhi((uint32_t)(k0 + g * 32), gmul));
        if (qg != gprev) {
            gprev = qg;
            fetch_zs(qg);
            fresh = true;
        }
        if (fresh) {
            fresh = false;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const uint32_t zz = __builtin_amdgcn_perm(zw[b], zw[b], selz);
                zc[b] = __builtin_bit_cast(h2, and_or(zz >> sh4, LO, magic));
                sp[b] = __builtin_bit_cast(h2, __builtin_amdgcn_perm(sw[b], sw[b], 0x01000100u));
            }
        }
        // next group's activations in flight during this group's math
        half8 xn[NJ];
        const int gn = g + WAVES < groups ? g + WAVES : g;
#pragma unroll
        for (int j = 0; j < NJ; ++j) xn[j] = *(const half8*)(xrow[j] + gn * 32);
        const char* grp = smem + g * 2048 + rd0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) w[i] = *(const uint32_t*)(grp + i * 256 + (b >> 1) * 64 + (b & 1) * 8);
            uint32_t a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t xx = __builtin_amdgcn_perm(w[2 * m + 1], w[2 * m], selx);
                const h2 qv = __builtin_bit_cast(h2, and_or(xx >> sh4, LO, magic));
                a[m] = __builtin_bit_cast(uint32_t, (qv - zc[b]) * sp[b]);
            }
            const half8 wf = __builtin_bit_cast(half8, (u32x4){a[0], a[1], a[2], a[3]});
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xb[j], acc[b][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) xb[j] = xn[j];
    }

    // ---- the four waves' partial sums meet in LDS (the word ring is idle once every wave is here): [wave][b][j][lane] x 16 B
    __syncthreads();
    f32x4* red = (f32x4*)smem;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < NJ; ++j) red[((wave * NB + b) * NJ + j) * 64 + lane] = acc[b][j];
    __syncthreads();
    // wave w sums and owns channel blocks BPW w .. BPW w + BPW - 1
    f32x4 tot[BPW][NJ];
#pragma unroll
    for (int bb = 0; bb < BPW; ++bb)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 t = red[((0 * NB + BPW * wave + bb) * NJ + j) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < WAVES; ++w2) t = t + red[((w2 * NB + BPW * wave + bb) * NJ + j) * 64 + lane];
            tot[bb][j] = t;
        }
    auto store_y = [&](const f32x4 (&t)[BPW][NJ]) {   // lane: token 16 j + r, channels n0 + 16 (BPW wave + bb) + 4 q .. + 3
#pragma unroll
        for (int bb = 0; bb < BPW; ++bb)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int m = j * 16 + r, n = n0 + 16 * (BPW * wave + bb) + 4 * q;
                if (m < M && n < OC) {
                    const half4 h = {(_Float16)t[bb][j][0], (_Float16)t[bb][j][1], (_Float16)t[bb][j][2], (_Float16)t[bb][j][3]};
                    *(half4*)(y + (int64_t)m * OC + n) = h;
                }
            }
    };
    if (S == 1) {
        store_y(tot);
        return;
    }
    // ---- park the partial (write-through), count, and let the channel block's last arriver add the S partials in slice order
    constexpr int SLOT = BN * 16 * NJ;                           // floats of one partial: [channel block][j][lane][4]
    const rsrc_t rs_p = make_rsrc(part + (int64_t)blk * S * SLOT, (uint32_t)S * SLOT * 4u);
#pragma unroll
    for (int bb = 0; bb < BPW; ++bb)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tot[bb][j]), rs_p,
                                                   (uint32_t)((sl * SLOT) + (((BPW * wave + bb) * NJ + j) * 64 + lane) * 4) * 4u, 0u, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int last_s;
    if (tid == 0) last_s = __hip_atomic_fetch_add(cnt + blk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
    __syncthreads();
    if (!last_s) return;
#pragma unroll
    for (int bb = 0; bb < BPW; ++bb)
#pragma unroll
        for (int j = 0; j < NJ; ++j) tot[bb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < S; ++s2) {
#pragma unroll
        for (int bb = 0; bb < BPW; ++bb)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_p, (uint32_t)((s2 * SLOT) + (((BPW * wave + bb) * NJ + j) * 64 + lane) * 4) * 4u,
                                                                       0u, 16);      // agent scope: past this XCD's L2
                tot[bb][j] = tot[bb][j] + __builtin_bit_cast(f32x4, v);
            }
    }
    store_y(tot);
    if (tid == 0) __hip_atomic_store(cnt + blk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NJ>
int launch_nj(const void* x, const void* kernel, const void* scales, const void* zeros, void* y, int M, int IC, int OC, int G,
              void* workspace, size_t ws_bytes, hipStream_t stream) {
    constexpr int WAVES = NJ <= 2 ? 8 : 4, THREADS = WAVES * 64;
    const int nblk = (OC + BN - 1) / BN, NT = IC / 64;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    // K slices: one workgroup per CU (blocks x S <= CUs) -- or two, where that leaves a quarter of the chip idle (86 blocks: 2 slices
    // = 172 workgroups) and a slice's code words (64 B per row) and the reduction scratch (32 KiB per token block), which share
    // the LDS, leave room for two workgroups on a CU
    int S = cus / nblk;
    if (S < 1) S = 1;
    if (S > NT) S = NT;
    if (nblk * S * 4 < cus * 3) {
        int S2 = 2 * cus / nblk;
        S2 = S2 > NT ? NT : S2;
        const size_t words2 = (size_t)((NT + S2 - 1) / S2) * 64 * 64, red2 = (size_t)WAVES * NB * NJ * 64 * 16;
        if (S2 > S && (words2 > red2 ? words2 : red2) <= 78 * 1024) S = S2;
    }
    const size_t slot = (size_t)BN * 16 * NJ * sizeof(float);
    if (!workspace || nblk * (int)sizeof(int) > 16384) S = 1;
    while (S > 1 && ws_bytes < (size_t)CNT_BYTES + (size_t)nblk * S * slot) --S;
    int steps = (NT + S - 1) / S;
    while (steps * 64 > 2304) {                              // the slice must fit the LDS: more slices if there is a workspace for them
        if (!workspace || nblk * (int)sizeof(int) > 16384) return MXQ_NOT_MY_SHAPE;
        ++S;
        if (ws_bytes < (size_t)CNT_BYTES + (size_t)nblk * S * slot) return MXQ_NOT_MY_SHAPE;
        steps = (NT + S - 1) / S;
    }
    S = (NT + steps - 1) / steps;                            // no empty slice
    const int rows_per_slice = steps * 64;
    size_t smem = (size_t)rows_per_slice * 64;
    const size_t red = (size_t)WAVES * NB * NJ * 64 * 16;
    if (red > smem) smem = red;
    hipError_t e = mxq_set_dyn_lds_once<&mxq_skinny_awq_f16_kernel<NJ, WAVES>>(160 * 1024 - 64);
    if (e != hipSuccess) return (int)e;
    const uint32_t gmul = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)G) + 1u;
    mxq_skinny_awq_f16_kernel<NJ, WAVES><<<nblk * S, THREADS, smem, stream>>>(
        (const uint16_t*)x, (const uint32_t*)kernel, (const uint16_t*)scales, (const uint32_t*)zeros, (uint16_t*)y, M, IC, OC, gmul, IC / G, S,
        rows_per_slice, (int*)workspace, workspace ? (float*)((char*)workspace + CNT_BYTES) : nullptr);
    return (int)hipGetLastError();
}

}   // namespace

// 1 <= M <= 32; IC % 64 == 0, OC % 8 == 0, G % 32 == 0 (one quantisation group per 32-row step), IC % G == 0.  MXQ_NOT_MY_SHAPE:
// more tokens, or a K range that does not fit the LDS without slices and no workspace to slice with (caller: the tile kernel).
int mxq_launch_skinny_awq_f16(const void* x, const void* kernel, const void* scales, const void* zeros, void* y, int M, int IC, int OC,
                              int G, void* workspace, size_t ws_bytes, hipStream_t stream) {
    if (M > 32) return MXQ_NOT_MY_SHAPE;
    if (M < 1 || IC < 64 || IC % 64 != 0 || OC % 8 != 0 || OC < 32 || G < 32 || G % 32 != 0 || IC % G != 0 || G >= 4096 || IC >= (1 << 20))
        return -1;
    if ((int64_t)IC * OC / 2 >= ((int64_t)1 << 31) || (int64_t)(IC / G) * OC * 2 >= ((int64_t)1 << 31)) return -1;
    if (M <= 16) return launch_nj<1>(x, kernel, scales, zeros, y, M, IC, OC, G, workspace, ws_bytes, stream);
    return launch_nj<2>(x, kernel, scales, zeros, y, M, IC, OC, G, workspace, ws_bytes, stream);
}
