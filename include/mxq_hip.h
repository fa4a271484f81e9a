/* libmxq_hip.so -- C ABI of the MI355X-native MXQ hot path (gfx950 / CDNA4).
 *
 * This is the drop-in boundary: the entry points below are what a binding of the
 * reference's native module would call.  The reference's boundary is the pybind11 torch
 * extension `mxq_inference_engine`
 *     mxq_quant/cuda_kernel/csrc/pybind.cpp:6-10        (module, two exports)
 *     mxq_quant/cuda_kernel/csrc/quantization/gemv_cuda.h:4-9       gemv_forward_cuda
 *     mxq_quant/cuda_kernel/csrc/quantization/gemv_mxq_cuda.h:4-12  gemv_mxq_forward_cuda
 *     mxq_quant/cuda_kernel/csrc/quantization/gemm_cuda.h:3-4       gemm_forward_cuda (declared, unexported)
 * plus the Python ops the kernels must agree with
 *     mxq_quant/lib/quantizer.py:14-20, 61-147   quantize / dequantize / find_params
 *     mxq_quant/lib/mxqgpt.py:387-448            MXQGPT.fasterquant (layout)
 *     LLM-QAT/models/utils_quant.py:316-475      MXAsymQuantizer forward / backward
 *
 * Conventions
 *   - plain pointers and sizes; no torch types; every pointer is a DEVICE pointer
 *     (hipMalloc'd or a torch tensor's data_ptr()) unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - every function returns 0 on success, a positive hipError_t if the HIP runtime
 *     reported one at launch, or a negative MXQ_E* code for rejected arguments;
 *   - nothing here allocates or frees device memory, and nothing synchronises except mxq_workspace_status: outputs
 *     and scratch are caller-allocated and the library is re-entrant (safe under hipGraph capture).  The library keeps
 *     exactly two pieces of process-global HOST state, both caches of device facts that never change: the CU count of
 *     the first device a stream-K launcher ran on (csrc/gemm8.hip cu_count(): one device model per process is assumed),
 *     and one bit per (kernel, device ordinal) recording that hipFuncSetAttribute(MaxDynamicSharedMemorySize) was made
 *     (csrc/mxq_kernels.h mxq_set_dyn_lds_once).  No data-dependent state survives a call;
 *   - shapes: weight W[N, K] row-major (nn.Linear.weight), N % 16 == 0, K % 64 == 0.
 */
#ifndef MXQ_HIP_H
#define MXQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MXQ_DTYPE_F32 0
#define MXQ_DTYPE_F16 1
#define MXQ_DTYPE_BF16 2

#define MXQ_E_SHAPE (-1)   /* N % 16, K % 64, sizes <= 0, unsupported group size ... */
#define MXQ_E_NULL (-2)    /* a required pointer is NULL */
#define MXQ_E_DTYPE (-3)   /* unknown dtype code */
#define MXQ_E_ALIGN (-4)   /* a pointer is not 16-byte aligned */

/* Library / packed-format version: (format << 16) | api.  No reference counterpart
 * (the reference has no packed checkpoint format). */
int mxq_version(void);

/* Bytes of the packed buffers for W[N, K] (format v1, csrc/mxq_format.h): qweight holds
 * codes + per-group metadata (~4.4 bit/weight), rowmeta one float4 per row.  Host-only
 * helpers, no device work.  Return 0 for invalid shapes. */
size_t mxq_qweight_bytes(int N, int K);
size_t mxq_rowmeta_bytes(int N);

/* Fused quantise-and-pack of a weight matrix on device.
 * Replaces the Python loop of MXQGPT.fasterquant (mxqgpt.py:404-443: 3 x
 * Quantizer(bits=2, qq_scale_bits=4) per 64-column chunk + one Quantizer(bits=4) over the
 * gathered last-16 columns) and produces the packed form instead of fake-quant fp16.
 * W: [N, K] of `w_dtype`; dead: optional uint8[K] mask of columns to zero first
 * (diag(H) == 0, mxqgpt.py:401-403), may be NULL. */
int mxq_quantize_pack(const void* W, int w_dtype, const uint8_t* dead, void* qweight, void* rowmeta, int N, int K,
                      void* stream);

/* Pack the lossless parameterisation (as produced by Quantizer.quantize / find_params,
 * quantizer.py:14-16, 114-121) into format v1.  Shapes (G = 3*K/64 two-bit groups per row):
 *   codes2 u8[N, 3K/4] (chunk-major, 48 per chunk), sc2 u8[N, G], zero2 f32[N, G],
 *   qs2/qz2 f32[N/16, G], codes4 u8[N, K/4], sc4 u8[N], zero4 f32[N], qs4/qz4 f32[N/16]. */
int mxq_pack_codes(const uint8_t* codes2, const uint8_t* sc2, const float* zero2, const float* qs2,
                   const float* qz2, const uint8_t* codes4, const uint8_t* sc4, const float* zero4,
                   const float* qs4, const float* qz4, void* qweight, void* rowmeta, int N, int K, void* stream);

/* Integer unpack: exact inverse of mxq_pack_codes (the bit-exact unpack contract:
 * codes == Quantizer.quantize output, quantizer.py:14-16). */
int mxq_unpack(const void* qweight, const void* rowmeta, uint8_t* codes2, uint8_t* sc2, float* zero2, float* qs2,
               float* qz2, uint8_t* codes4, uint8_t* sc4, float* zero4, float* qs4, float* qz4, int N, int K,
               void* stream);

/* Dequantise to a dense fp16 [N, K] matrix: bit-identical to the weight that
 * MXQGPT.fasterquant writes back (scale*(q-zero) in fp32, one rounding to fp16;
 * quantizer.py:19-20, mxqgpt.py:448). */
int mxq_dequant_f16(const void* qweight, const void* rowmeta, void* w16, int N, int K, void* stream);

/* y[M, N] (fp16) = x[M, K] (fp16) . dequant(qweight)[N, K]^T, fp32 accumulation.
 * The fused unpack + per-group scale/zero + W2/4 x A16 product behind a quantised
 * nn.Linear.  Counterpart of gemm_forward_cuda (gemm_cuda.h:3-4, MFMA path, any M) and of
 * gemv_mxq_forward_cuda (gemv_mxq_cuda.h:4-12, streaming path, chosen for M <= 4).
 * Without a workspace the K range is never split: streaming GEMV (M <= 4), skinny MFMA kernel (<= 64), mid-M kernel
 * with one slice per tile (<= 256), prefill kernel with whole tiles beyond; mxq_linear_f16_ws is the fast entry for
 * 20 < M <= 1024.  x, y, qweight must be 16-byte aligned; y must not alias x. */
int mxq_linear_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   void* stream);
/* The code paths of mxq_linear_f16, exposed for benchmarking / testing: streaming GEMV (M <= 4), skinny MFMA kernel
 * (5..40 tokens in the dispatch: every packed byte read once, a lane dequantises the MFMA operand it owns; the reference
 * re-reads the weights once per batch row, gemv_mxq_cuda.cu:261-262), prefill GEMM.  mxq_skinny_f16 accepts 1 <= M <= 64
 * and any layout (MXQ_LAYOUT_*, below: mixed with exact or compact metadata, W2G16, W4ROW). */
int mxq_skinny_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, int layout,
                   void* stream);
int mxq_gemm_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                 void* stream);
int mxq_gemv_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                 void* stream); /* M <= 4 */
/* The same Linear with a caller-owned scratch buffer, which lets the prefill GEMM balance launches
 * whose tile count is not a multiple of the CU count (stream-K tail, csrc/gemm8.hip): e.g. 512 tokens
 * x 4096^2 is 64 tiles, a quarter of the 256 CUs, unless every CU takes a quarter of a tile's K range.
 * `workspace` is device memory of at least mxq_gemm_workspace_bytes() bytes, 16-byte aligned, whose
 * first 64 KiB the caller zeroes ONCE (hipMemset) before first use; the kernels leave it zeroed, so
 * it can be reused by every later launch on the SAME stream (launches on different streams need
 * different workspaces).  workspace == NULL selects the workspace-free schedule of mxq_linear_f16; a workspace too small
 * for the mid-M kernel's partial tiles hands 41..256 tokens to the prefill kernel.
 * The tail is only split where that pays (about 20 idle K-steps per CU: for 256 < M <= 1024 on the Llama shapes and for
 * gate/up at M = 2048).  Beyond 64 tokens, launches of up to 176 tiles of 128 x 128 run the same kernel built with a
 * 128-token tile (csrc/gemm8h.hip; csrc/gemm8n.hip, a 128 x 64 tile, where K <= 6144): up to 64 tiles (e.g. 4096^2 at 65-256
 * tokens) with K cut into one slice per idle CU, fp32 slabs in the workspace beyond its first 64 KiB and a combine launch; 65..176 tiles (gate/up at 65-256 tokens, 4096^2 at
 * 257-640) in one launch, stream-K over the otherwise idle CUs.  Up to 64 tokens, launches of 65..176 tiles of 64 x 128
 * (gate/up) run its 64-token build (csrc/gemm8q.hip) the same way.
 * mxq_gemm_f16_ws runs ONE named schedule (enum mxq_gemm_variant below) at any token count: the entry behind the parity
 * tests of every kernel build and behind tools/'s A/B timings; production callers use mxq_linear_f16_auto.  Results of every
 * variant agree to fp32-summation-order rounding and are run-to-run deterministic; an unknown code is MXQ_E_SHAPE.
 * (Profiling-only ablation builds live in libmxq_hip_prof.so, `make prof`, used by tools/ alone: not part of this ABI.) */
enum mxq_gemm_variant {
    MXQ_GEMM_DEFAULT = 0,           /* by token count: 128 x 128-tile kernel (no workspace) or the 256 x 128 fused kernel */
    MXQ_GEMM_TILE128 = 1,           /* csrc/gemm.hip: 128 x 128 tile, two LDS stages; workspace ignored */
    MXQ_GEMM_FUSED256 = 8,          /* csrc/gemm8.hip: 256 x 128 tile, wave-specialised, persistent; tail split where it pays */
    MXQ_GEMM_FUSED256_SPLIT = 9,    /* ... tail split whenever that is structurally possible */
    MXQ_GEMM_MIDM = 10,             /* csrc/midm.hip: split-K slices + combine launch */
    MXQ_GEMM_FUSED128 = 12,         /* csrc/gemm8h.hip: the fused kernel's 128-token build, tail split where it pays */
    MXQ_GEMM_FUSED128_SPLIT = 13,   /* ... always split */
    MXQ_GEMM_FUSED128_SLICES = 14,  /* ... K slices + combine launch */
    MXQ_GEMM_FUSED64_SPLIT = 16,    /* csrc/gemm8q.hip: the 64-token build, stream-K tail always split */
    MXQ_GEMM_FUSED64_SLICES = 17,   /* ... K slices + combine launch */
    MXQ_GEMM_FUSED128N64_SPLIT = 20,  /* csrc/gemm8n.hip: the 128 x 64-tile build, tail always split */
    MXQ_GEMM_FUSED128N64_SLICES = 21  /* ... K slices + combine launch */
};
size_t mxq_gemm_workspace_bytes(void);
/* A stream-K launch makes workgroups wait for other workgroups of the SAME launch (an owner for the lower-numbered units of
 * its tile; in the all-contributors reduction, any contributor for any other).  That needs the launch's workgroups resident
 * together -- the grid is at most one workgroup per CU -- so a stream-K launch must not share the device with another
 * kernel that ALSO waits on its own unscheduled workgroups (two stream-K launches on two streams can starve each other;
 * ordinary kernels only delay it).  Every wait is bounded (~4 M polls, seconds).  A wait that gives up is never silent:
 * the tile (or token block) whose partial sums did not arrive is written as NaN, and the last four ints of the workspace's
 * 64-KiB head become {code, tail-tile index, workgroup, count seen} (code 1: an owner's wait, 2: the all-contributors
 * reduction's; 0: nothing happened).  After a non-zero code the counters are inconsistent: zero the head again before the
 * workspace is reused.  mxq_workspace_status copies the four ints to `status4` (HOST memory) after synchronising `stream`
 * -- the one entry point of this library that synchronises. */
int mxq_workspace_status(const void* workspace, size_t workspace_bytes, int* status4, void* stream);
/* Measurement helper: one launch of 8 single-wave workgroups (one per XCD) writing {s_memtime, s_memrealtime (100 MHz), XCC id, 0}
 * as 4 x uint64 each into `out32_u64` (32 x uint64, device memory).  Two stamps on one stream around a timed region give the
 * shader clock the chip held in between, d(s_memtime) / d(s_memrealtime) x 100 MHz per XCD (bench.py: roofline.sclk_MHz). */
int mxq_clock_stamp(void* out32_u64, void* stream);
/* Bytes of workspace the dispatch of mxq_linear_f16_auto can use for a call of this shape (0: it never touches one --
 * GEMV, skinny kernel, or, with `hoisting` != 0, the hoisted mode from mxq_hoist_min_tokens() tokens on); at most
 * mxq_gemm_workspace_bytes().  For callers that own one workspace per captured graph.  Host-only helper. */
size_t mxq_linear_workspace_need(int M, int N, int K, int layout, int hoisting);
/* hipStreamGetCaptureInfo through this library's HIP runtime: *active = 1 and *id = the capture sequence's id while
 * `stream` is being captured, else *active = 0, *id = 0.  Host-only helper (no device work). */
int mxq_stream_capture_id(void* stream, int* active, unsigned long long* id);
int mxq_linear_f16_ws(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                      void* workspace, size_t workspace_bytes, void* stream);
int mxq_gemm_f16_ws(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                    int variant, void* workspace, size_t workspace_bytes, void* stream);
/* mxq_linear_f16_ws for a weight in any layout (MXQ_LAYOUT_*: mixed with exact or compact metadata, W2G16, W4ROW).
 * Dispatch by token count: <= 4 the streaming GEMV, then the skinny MFMA kernel (mixed layouts: up to 40 tokens, 20 for
 * weights of more than 24 M elements, where its time has overtaken the split-K kernel's; uniform layouts: up to 48, where
 * the prefill kernel takes over), beyond 64 tokens launches of up to 176 tiles of 128 x 128 the fused kernel's 128-token
 * build (every layout; slices mode up to 64 tiles, stream-K mode beyond), and otherwise up to 256 tokens the mid-M split-K
 * kernel (csrc/midm.hip: K cut into slices over the workgroups, fp32 partial tiles in the workspace beyond its first
 * 64 KiB, summed in slice order by a second launch -- the reference launcher's split_k_iters regime,
 * gemm_cuda_gen.cu:429-475) for the mixed layouts, the prefill kernel otherwise.  The mid-M kernel does not touch
 * the workspace's counters.  workspace == NULL: ONE workspace-free schedule for every layout -- GEMV (<= 4 tokens), the skinny
 * kernel (mixed layouts <= 64, uniform <= 48), the mid-M kernel unsplit (mixed layouts, <= 256), whole tiles of the prefill
 * kernel beyond (csrc/capi.hip linear_noworkspace). */
int mxq_linear_f16_layout_ws(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                             int layout, void* workspace, size_t workspace_bytes, void* stream);

/* Hoisted-dequant mode of the same Linear, for launches that cover many token tiles (batch x seq >= 4096 tokens, e.g.
 * BASELINE configs[4]; mxq_hoist_min_tokens()): the fused kernel dequantises each 128 x 64 weight tile once per 256 tokens; here the bit-exact
 * dequant kernel writes fp16 weights into `w16_scratch` (>= mxq_hoist_scratch_bytes(N, K) = 2*N*K bytes of device
 * memory, 16-byte aligned, overwritten) ONCE and the MFMA kernel streams fp16 tiles from it -- same products, same
 * summation order, results bit-identical to mxq_gemm_f16_layout.  The scratch is transient (nothing is cached
 * between calls); layout as below (0 mixed, 1 W2G16, 2 W4ROW, 3 mixed with compact metadata). */
size_t mxq_hoist_scratch_bytes(int N, int K);
int mxq_linear_f16_hoisted(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                           int layout, void* w16_scratch, size_t scratch_bytes, void* stream);

/* ONE call for the whole dispatch -- the counterpart of the reference's single entry, whose launcher picks its schedule
 * inside (gemm_cuda.h:3-4, gemm_cuda_gen.cu:424-478: tile kernel by OC, split_k_iters): streaming GEMV, skinny MFMA kernel,
 * mid-M split-K kernel, fused prefill GEMM with its stream-K tail (mxq_linear_f16_layout_ws with `workspace`, which may
 * be NULL), and -- from mxq_hoist_min_tokens() tokens on, when `w16_scratch` holds mxq_hoist_scratch_bytes(N, K) bytes --
 * the hoisted-dequant mode above.  A NULL or too small scratch never fails: the fused kernel runs instead. */
int mxq_hoist_min_tokens(void);
int mxq_linear_f16_auto(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, int layout,
                        void* workspace, size_t workspace_bytes, void* w16_scratch, size_t scratch_bytes, void* stream);

/* The second half of the hoisted mode on its own = the reference's implicit nn.Linear on the fake-quant fp16 weight
 * (mxq_quant/main.py:85 -> lib/eval.py:54): y[M, N] = x[M, K] . w16[N, K]^T, fp16 operands, fp32
 * accumulation in K order, fp16 result; w16 e.g. from mxq_dequant_f16.  variant 0: kernel by shape (the 256 x 256-tile
 * kernel for launches of >= 2 tiles per CU with an even number of 64-deep K-tiles, else the 256 x 128 one), 1 / 2: that
 * kernel explicitly (2 -> MXQ_E_SHAPE if K % 128 != 0).  Both kernels give bit-identical results. */
int mxq_dense_f16(const void* x, const void* w16, void* y, int M, int N, int K, int variant, void* stream);

/* Uniform-bit-width layouts for the W2A16 / W4A16 / mixed sweep (BASELINE config 5); the mixed
 * layout is layout 0.  1 = W2G16: Quantizer(bits=2, qq_scale_bits=4) on every 16-column group
 * (quantizer.py:61-147), 4.5 bit/weight; 2 = W4ROW: Quantizer(bits=4, qq_scale_bits=4) per row
 * (the mixed layout's 4-bit arm on all columns), 4 bit/weight.  Same block grid / bit order. */
#define MXQ_LAYOUT_MIXED 0
#define MXQ_LAYOUT_W2G16 1
#define MXQ_LAYOUT_W4ROW 2
/* 3 = MIXEDC: the mixed layout with COMPACT metadata ("format v2", csrc/mxq_format.h): same codes / scale codes /
 * (qs, qz), 2-bit zero-points stored as fp16 -> 480-B blocks, 3.75 bit/weight instead of 4.5.  The integer unpack
 * stays bit-exact on the codes; the dequantised weight becomes fp16(scale * (q - float(fp16(zero)))), which moves
 * the GEMM result by ~4e-4 relative (inside the 1e-3 budget).  The reference's own kernel
 * format (gemv_mxq_cuda.cu:54-62, integer zeros) is coarser still and is served by mxq_gemv_proto_f16. */
#define MXQ_LAYOUT_MIXEDC 3
size_t mxq_qweight_bytes_layout(int N, int K, int layout);
/* exact (layout 0, from mxq_quantize_pack / mxq_pack_codes) -> compact (layout 3) blocks; rowmeta is shared. */
int mxq_compact(const void* qweight_exact, void* qweight_compact, int N, int K, void* stream);
/* mxq_unpack / mxq_dequant_f16 / mxq_gemv_fused_f16 on a compact qweight (same arguments; zero2 comes back as the
 * stored fp16 values widened to fp32).  GEMM / GEMV: mxq_gemm_f16_layout / mxq_gemv_f16_layout with layout 3. */
int mxq_unpack_compact(const void* qweight, const void* rowmeta, uint8_t* codes2, uint8_t* sc2, float* zero2, float* qs2,
                       float* qz2, uint8_t* codes4, uint8_t* sc4, float* zero4, float* qs4, float* qz4, int N, int K,
                       void* stream);
int mxq_dequant_f16_compact(const void* qweight, const void* rowmeta, void* w16, int N, int K, void* stream);
int mxq_gemv_fused_f16_compact(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K,
                               int prologue, const void* norm_w, float eps, const void* residual, void* stream);
int mxq_quantize_pack_layout(const void* W, int w_dtype, void* qweight, void* rowmeta, int N, int K, int layout,
                             void* stream);
/* Uniform layouts only: dequantise to fp16 [N, K] (w16, nullable) and / or integer-unpack (codes u8
 * [N, K], nullable, with sc u8 / zero f32 [N, K/16] and qs / qz f32 [N/16, K/16] for W2G16, or sc / zero
 * [N] and qs / qz [N/16] for W4ROW). */
int mxq_expand_layout(const void* qweight, const void* rowmeta, void* w16, uint8_t* codes, uint8_t* sc, float* zero,
                      float* qs, float* qz, int N, int K, int layout, void* stream);
/* MFMA dequant-GEMM (the wave-specialised kernel) and the streaming GEMV (M <= 4 tokens) on any layout:
 * one kernel template each, the layouts differ in the field offsets and in the group / quarter loop. */
int mxq_gemm_f16_layout(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                        int layout, void* stream);
int mxq_gemv_f16_layout(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                        int layout, void* stream);

/* Decode-time variant of mxq_gemv_f16 for ONE token with the neighbouring elementwise ops of a
 * Llama decoder layer fused into the activation staging / the store (no reference counterpart:
 * the reference's decode is HF transformers on fake-quant weights):
 *   prologue 0: none; 1: x <- RMSNorm(x) * norm_w (fp16 norm_w[K], eps); 2: x[2K] = (gate, up),
 *   x <- silu(gate) * up.  residual (nullable, fp16 [N]): y <- residual + W.x. */
int mxq_gemv_fused_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K, int prologue,
                       const void* norm_w, float eps, const void* residual, void* stream);
/* The decoder MLP's two launches with the SwiGLU moved from the consumer's staging into the producer's final reduction
 * (round 5; no reference counterpart).  mxq_gemv_swiglu_f16: x fp16[K] -> RMSNorm prologue (norm_w, eps) -> the weight
 * [N2 = 2 I, K] = gate stacked on up; a workgroup takes the same 16 rows of gate and of up, and writes
 * act = fp16(silu(gate)) * up -- one 16-column scale group of the down projection's input -- in the kernels' staged order
 * (per eight: x0, x2, x1, x3, x4, x6, x5, x7) into `act` (fp16 [I]) with the group's fp32 sum in `act_sum` (f32 [I / 16]).
 * mxq_gemv_staged_f16: y fp16[N] = residual (nullable) + W[N, K] . act for such a staged row.  Bit for bit the results of
 * mxq_gemv_fused_f16 prologue 1 followed by prologue 2.  N2 % 32 == 0, K % 256 == 0 (first call). */
int mxq_gemv_swiglu_f16(const void* x, const void* qweight, const void* rowmeta, void* act, void* act_sum, int N2, int K,
                        const void* norm_w, float eps, int compact, void* stream);
int mxq_gemv_staged_f16(const void* x_staged, const void* x_sum, const void* qweight, const void* rowmeta, void* y, int N, int K,
                        const void* residual, int compact, void* stream);
/* Decode-harness glue (BASELINE config 3), not part of the reference's hot path: rotary embedding
 * of q/k, KV-cache append at *pos and single-query attention for one token; one workgroup per
 * head, head_dim 128.  qkv fp16 [3*heads*128]; caches fp16 [heads][max_ctx][128]; pos int64[1]
 * (device); cos/sin f32 [max_ctx][64]; out fp16 [heads*128].
 * Precondition: 0 <= *pos < max_ctx at the time the kernel runs (the caller tracks the position it increments
 * on the device, e.g. mxq_amd/llama_decode.py's ContextWindow).  A position outside the cache is not written
 * anywhere: that launch leaves the caches untouched and fills `out` with NaN. */
int mxq_attn_decode_f16(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* cos_t,
                        const void* sin_t, void* out, int heads, int head_dim, int max_ctx, void* stream);

/* The same with the rotary table rows of the CURRENT position gathered once per token instead of once per layer:
 * mxq_rope_row_f32 writes row = {cos_t[*pos][0..half_dim), sin_t[*pos][0..half_dim)} (f32 [2 * half_dim]; *pos clamped
 * into [0, max_ctx)), and mxq_attn_decode_row_f16 takes that row in place of the two tables -- its launch then has no
 * load that depends on another load's result except the cache rows beyond the 64th key.  Same arithmetic, same results. */
int mxq_rope_row_f32(const void* pos, const void* cos_t, const void* sin_t, void* row, int half_dim, int max_ctx, void* stream);
/* ... and the token's embedding row in the same launch, for the stage that owns the embedding: h_out f16[hidden] =
 * embed f16[vocab, hidden][*token] (token int64[1] on the device, clamped into [0, vocab)), next to the rotary row. */
int mxq_embed_rope_row(const void* token, const void* embed, int vocab, int hidden, void* h_out, const void* pos, const void* cos_t,
                       const void* sin_t, void* row, int half_dim, int max_ctx, void* stream);
int mxq_attn_decode_row_f16(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* rope_row, void* out,
                            int heads, int head_dim, int max_ctx, void* stream);

/* The same for LONG contexts (round 5): one workgroup per head streams a head's whole cache through one CU (5 us per layer at
 * 72 keys, 13 at 450, ~40 at 2048); here a head's keys are split over up to `splits` (<= 16) workgroups (grid heads x splits), each
 * parks its partial softmax {o[128], max, sum} in `workspace`, and the head's LAST arriver (an arrival counter: nobody waits)
 * merges them.  Up to 128 keys a head is ONE workgroup running mxq_attn_decode_row_f16's algorithm bit for bit, so short
 * contexts pay nothing; beyond, the result differs from the one-workgroup kernel by fp32 summation order and by not
 * rounding the probabilities to fp16.  workspace: mxq_attn_split_workspace_bytes(heads, splits) bytes, 16-byte aligned, its
 * first 1024 * ceil(heads / 256) bytes (the counters) zeroed ONCE by the caller; the kernel leaves them zeroed. */
size_t mxq_attn_split_workspace_bytes(int heads, int splits);
int mxq_attn_decode_split_f16(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* rope_row, void* out,
                              int heads, int head_dim, int max_ctx, int splits, void* workspace, void* stream);

/* Decode-harness glue (BASELINE config 3), not part of the reference's hot path: final RMSNorm + the fp16 lm_head
 * Linear + greedy argmax of ONE token.  h fp16[K], norm_w fp16[K], w fp16[V, K] (row-major nn.Linear weight), K = 4096;
 * the normalised row is fp16(h * rsqrt(mean h^2 + eps)) * norm_w, a logit the fp32 dot rounded to fp16, the result
 * the lowest index of the largest logit (NaN logits never win), written to *token (int64, device).  part: scratch of
 * part_slots * 8 bytes (>= 256 slots recommended; fewer slots = fewer workgroups). */
int mxq_lmhead_argmax_f16(const void* h, const void* norm_w, float eps, const void* w, int V, int K, void* part,
                          int part_slots, void* token, void* stream);
/* The same, plus the token loop's bookkeeping in the second launch: the id is also stored at generated[*pos] (int64
 * [max_generated], nullable; skipped when *pos is outside it) and the device-resident position *pos (int64[1]) moves on by
 * one -- a decoded token then costs no separate "position += 1" and "append the id" launches. */
int mxq_lmhead_argmax_advance_f16(const void* h, const void* norm_w, float eps, const void* w, int V, int K, void* part,
                                  int part_slots, void* token, void* pos, void* generated, int max_generated, void* stream);

/* MXAsymQuantizer.forward (utils_quant.py:316-462; 2-D, layerwise=False branch):
 * fake-quantise w[rows, cols] of `dtype` into out (same shape/dtype), bit-identical to
 * the reference in fp32 / bf16 / fp16.  cols % 64 == 0. */
int mxq_fakequant_fwd(const void* w, void* out, int rows, int cols, int num_bits, int dtype, void* stream);

/* MXAsymQuantizer.backward (utils_quant.py:464-475): grad_in = grad_out where
 * lo < w < hi, else 0.  n elements, n % 8 == 0 (n % 4 for fp32). */
int mxq_fakequant_bwd(const void* grad_out, const void* w, void* grad_in, int64_t n, float lo, float hi, int dtype,
                      void* stream);

/* SymQuantizer.forward (symmetric = 1; utils_quant.py:31-86) and AsymQuantizer.forward (symmetric = 0;
 * utils_quant.py:105-182): the dynamic-range fake quantisers of activations / KV cache
 * (utils_quant.py:716-724, modeling_llama_quant.py:323-329), bit-identical in fp32 / bf16 / fp16.
 *   symmetric : s = (2^(bits-1) - 1) * (1 / (max|x| + 1e-6));  out = round(x * s) / (s + 1e-6)
 *   asymmetric: e = (max - min) + 1e-8;  out = round((x - min) / e * L) / L * e + min,  L = 2^bits - 1
 * mxq_actquant_group_fwd: x[rows, cols], one range per `group` consecutive columns (the reference's
 *   2-D branch: group 128 symmetric, 8 asymmetric); columns beyond the last whole group get range 0,
 *   as the reference leaves them.  cols % 8 == 0 (% 4 for fp32); group / (8 or 4) a power of two <= 32.
 * mxq_actquant_fwd: x is n_seg contiguous segments of seg_len elements, one range per segment
 *   (a token row of a 3-D activation, a (batch, head) slab of a 4-D tensor, the whole tensor for
 *   layerwise = True); segment i is ranged iff i % period < live, the others get range 0 (3-D inputs:
 *   the reference slices tokens by a group count taken from the hidden size).  seg_len % 8 == 0
 *   (% 4 for fp32).  range_ws: 8 * n_seg bytes of device scratch (overwritten).
 * The backward of both is mxq_fakequant_bwd (same straight-through clip, utils_quant.py:88-102). */
int mxq_actquant_group_fwd(const void* x, void* out, int64_t rows, int cols, int group, int num_bits, int symmetric,
                           int dtype, void* stream);
int mxq_actquant_fwd(const void* x, void* out, void* range_ws, int64_t n_seg, int64_t seg_len, int64_t period,
                     int64_t live, int num_bits, int symmetric, int dtype, void* stream);

/* gemv_forward_cuda(in_feats, kernel, scaling_factors, zeros, group_size)
 * (gemv_cuda.h:4-9; operand layout gemv_cuda.cu:45-59): x f16[B, IC], kernel i32[OC, IC/8],
 * scales f16[OC, sf_w], zeros i32[OC, zeros_w], group_size in {32, 64, 128}; y f16[B, OC].
 * x and kernel 16-byte aligned and IC % 32 == 0 (the codes are read with 16-byte loads), else MXQ_E_ALIGN / _SHAPE. */
int mxq_gemv_awq_f16(const void* x, const void* kernel, const void* scales, const void* zeros, void* y, int B,
                     int IC, int OC, int group_size, void* stream);

/* gemm_forward_cuda(in_feats, kernel, scaling_factors, zeros, split_k_iters) (gemm_cuda.h:3-4; launcher and its checks
 * gemm_cuda_gen.cu:424-478; nibble order dequantize.cuh:35-51) -- declared by the reference, never compiled into its module
 * (setup.py:37-41).  x f16[M, IC]; kernel i32[IC, OC/8] (a word = the 4-bit codes of 8 consecutive output channels at one
 * input channel, channel e in nibble (0, 4, 1, 5, 2, 6, 3, 7)[e]); scales f16[IC/G, OC]; zeros i32[IC/G, OC/8] (packed like
 * the codes); weight = fp16(fp16(q - z) * s) as the reference computes it (gemm_cuda_gen.cu:134-141), fp32 accumulation;
 * y f16[M, OC].  Runs on the fused prefill kernel's skeleton with dequant waves for this operand format (csrc/gemm8a.hip,
 * gemm8aq.hip).  The reference launcher's split_k_iters is a SCHEDULE (K slices summed by the caller), not part of the
 * operator: this entry schedules K itself -- up to 32 tokens a streaming kernel (128 channels x one K slice per workgroup, the last
 * arriver of a channel block sums the slices; csrc/skinny_awq.hip), then 64- / 128-token tiles by tile count (K slices + a combine
 * launch, or stream-K), beyond that 256-token tiles with a stream-K tail -- through `workspace` (mxq_gemm_workspace_bytes() bytes, 16-byte
 * aligned, counter head zeroed as for mxq_linear_f16_ws; NULL: whole tiles only).  Deterministic.  MXQ_E_SHAPE for the
 * launcher's rejections (OC % 64, group_size % 32, OC % group_size) and for IC % 64, IC % group_size. */
int mxq_gemm_awq_f16(const void* x, const void* kernel, const void* scales, const void* zeros, void* y, int M, int IC, int OC,
                     int group_size, void* workspace, size_t workspace_bytes, void* stream);

/* gemv_mxq_forward_cuda(in_feats, kernel, kernel_last, zeros_and_scales, scales_2nd,
 * zeros_2nd, scales_4b, zeros_4b, group_size) (gemv_mxq_cuda.h:4-12; operand layout
 * gemv_mxq_cuda.cu:54-62,96-201).  IC must be 4096 and group_size 16, as in the reference. */
int mxq_gemv_proto_f16(const void* x, const void* weight, const void* weight_last, const void* zeros_and_scales,
                       const void* scales_2nd, const void* zeros_2nd, const void* scales_4b, const void* zeros_4b,
                       void* y, int B, int IC, int OC, int group_size, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MXQ_HIP_H */
