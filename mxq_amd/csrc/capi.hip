// extern "C" surface of libmxq_hip.so (declared in include/mxq_hip.h): argument
// validation + dispatch to the launchers.  Never throws, allocates or synchronises.
#include <hip/hip_runtime.h>

#include "../../include/mxq_hip.h"
#include "mxq_format.h"
#include "mxq_kernels.h"

namespace {
inline bool shape_ok(int N, int K) { return N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0; }
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }
inline bool dtype_ok(int d) { return d == MXQ_DTYPE_F32 || d == MXQ_DTYPE_F16 || d == MXQ_DTYPE_BF16; }
inline bool layout_ok(int l) {
    return l == MXQ_LAYOUT_MIXED || l == MXQ_LAYOUT_W2G16 || l == MXQ_LAYOUT_W4ROW || l == MXQ_LAYOUT_MIXEDC;
}
}   // namespace

extern "C" {

int mxq_version(void) { return (MXQ_FORMAT_VERSION << 16) | 1; }

size_t mxq_qweight_bytes(int N, int K) {
    if (!shape_ok(N, K)) return 0;
    return (size_t)(N / 16) * (K / 64) * MXQ_BLK_BYTES;
}

size_t mxq_rowmeta_bytes(int N) { return N > 0 && N % 16 == 0 ? (size_t)N * 16 : 0; }

int mxq_quantize_pack(const void* W, int w_dtype, const uint8_t* dead, void* qweight, void* rowmeta, int N, int K,
                      void* stream) {
    if (!W || !qweight || !rowmeta) return MXQ_E_NULL;
    if (!shape_ok(N, K)) return MXQ_E_SHAPE;
    if (!dtype_ok(w_dtype)) return MXQ_E_DTYPE;
    if (!aligned16(W) || !aligned16(qweight) || !aligned16(rowmeta)) return MXQ_E_ALIGN;
    return mxq_launch_quantize_pack(W, w_dtype, dead, qweight, rowmeta, N, K, (hipStream_t)stream);
}

int mxq_pack_codes(const uint8_t* codes2, const uint8_t* sc2, const float* zero2, const float* qs2,
                   const float* qz2, const uint8_t* codes4, const uint8_t* sc4, const float* zero4,
                   const float* qs4, const float* qz4, void* qweight, void* rowmeta, int N, int K, void* stream) {
    if (!codes2 || !sc2 || !zero2 || !qs2 || !qz2 || !codes4 || !sc4 || !zero4 || !qs4 || !qz4 || !qweight ||
        !rowmeta)
        return MXQ_E_NULL;
    if (!shape_ok(N, K)) return MXQ_E_SHAPE;
    if (!aligned16(qweight) || !aligned16(rowmeta)) return MXQ_E_ALIGN;
    return mxq_launch_pack_codes(codes2, sc2, zero2, qs2, qz2, codes4, sc4, zero4, qs4, qz4, qweight, rowmeta, N, K,
                                 (hipStream_t)stream);
}

int mxq_unpack(const void* qweight, const void* rowmeta, uint8_t* codes2, uint8_t* sc2, float* zero2, float* qs2,
               float* qz2, uint8_t* codes4, uint8_t* sc4, float* zero4, float* qs4, float* qz4, int N, int K,
               void* stream) {
    if (!codes2 || !sc2 || !zero2 || !qs2 || !qz2 || !codes4 || !sc4 || !zero4 || !qs4 || !qz4 || !qweight ||
        !rowmeta)
        return MXQ_E_NULL;
    if (!shape_ok(N, K)) return MXQ_E_SHAPE;
    if (!aligned16(qweight) || !aligned16(rowmeta)) return MXQ_E_ALIGN;
    return mxq_launch_unpack(qweight, rowmeta, codes2, sc2, zero2, qs2, qz2, codes4, sc4, zero4, qs4, qz4, N, K, 0,
                             (hipStream_t)stream);
}

int mxq_unpack_compact(const void* qweight, const void* rowmeta, uint8_t* codes2, uint8_t* sc2, float* zero2,
                       float* qs2, float* qz2, uint8_t* codes4, uint8_t* sc4, float* zero4, float* qs4, float* qz4,
                       int N, int K, void* stream) {
    if (!codes2 || !sc2 || !zero2 || !qs2 || !qz2 || !codes4 || !sc4 || !zero4 || !qs4 || !qz4 || !qweight ||
        !rowmeta)
        return MXQ_E_NULL;
    if (!shape_ok(N, K)) return MXQ_E_SHAPE;
    if (!aligned16(qweight) || !aligned16(rowmeta)) return MXQ_E_ALIGN;
    return mxq_launch_unpack(qweight, rowmeta, codes2, sc2, zero2, qs2, qz2, codes4, sc4, zero4, qs4, qz4, N, K, 1,
                             (hipStream_t)stream);
}

int mxq_compact(const void* qweight_exact, void* qweight_compact, int N, int K, void* stream) {
    if (!qweight_exact || !qweight_compact) return MXQ_E_NULL;
    if (!shape_ok(N, K)) return MXQ_E_SHAPE;
    if (!aligned16(qweight_exact) || !aligned16(qweight_compact)) return MXQ_E_ALIGN;
    return mxq_launch_compact(qweight_exact, qweight_compact, N, K, (hipStream_t)stream);
}

int mxq_dequant_f16(const void* qweight, const void* rowmeta, void* w16, int N, int K, void* stream) {
    if (!qweight || !rowmeta || !w16) return MXQ_E_NULL;
    if (!shape_ok(N, K)) return MXQ_E_SHAPE;
    if (!aligned16(qweight) || !aligned16(rowmeta) || !aligned16(w16)) return MXQ_E_ALIGN;
    return mxq_launch_dequant_f16(qweight, rowmeta, w16, N, K, 0, (hipStream_t)stream);
}

int mxq_dequant_f16_compact(const void* qweight, const void* rowmeta, void* w16, int N, int K, void* stream) {
    if (!qweight || !rowmeta || !w16) return MXQ_E_NULL;
    if (!shape_ok(N, K)) return MXQ_E_SHAPE;
    if (!aligned16(qweight) || !aligned16(rowmeta) || !aligned16(w16)) return MXQ_E_ALIGN;
    return mxq_launch_dequant_f16(qweight, rowmeta, w16, N, K, 1, (hipStream_t)stream);
}

static int linear_check(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K) {
    if (!x || !qweight || !rowmeta || !y) return MXQ_E_NULL;
    if (!shape_ok(N, K) || M <= 0) return MXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(rowmeta) || !aligned16(y)) return MXQ_E_ALIGN;
    return 0;
}

int mxq_gemm_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                 void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    return mxq_launch_gemm_f16(x, qweight, rowmeta, y, M, N, K, (hipStream_t)stream);
}

int mxq_gemv_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                 void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (M > 4) return MXQ_E_SHAPE;
    return mxq_launch_gemv_f16(x, qweight, rowmeta, y, M, N, K, (hipStream_t)stream);
}

int mxq_skinny_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, int layout,
                   void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (M > 64 || !layout_ok(layout)) return MXQ_E_SHAPE;
    return mxq_launch_skinny_f16(x, qweight, rowmeta, y, M, N, K, layout, (hipStream_t)stream);
}

int mxq_gemv_fused_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K, int prologue,
                       const void* norm_w, float eps, const void* residual, void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, 1, N, K)) return e;
    if (prologue < 0 || prologue > 2) return MXQ_E_SHAPE;
    if (prologue == 1 && !norm_w) return MXQ_E_NULL;
    return mxq_launch_gemv_fused_f16(x, qweight, rowmeta, y, N, K, prologue, norm_w, eps, residual, 0,
                                     (hipStream_t)stream);
}

int mxq_gemv_fused_f16_compact(const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K,
                               int prologue, const void* norm_w, float eps, const void* residual, void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, 1, N, K)) return e;
    if (prologue < 0 || prologue > 2) return MXQ_E_SHAPE;
    if (prologue == 1 && !norm_w) return MXQ_E_NULL;
    return mxq_launch_gemv_fused_f16(x, qweight, rowmeta, y, N, K, prologue, norm_w, eps, residual, 1,
                                     (hipStream_t)stream);
}

int mxq_gemv_swiglu_f16(const void* x, const void* qweight, const void* rowmeta, void* act, void* act_sum, int N2, int K,
                        const void* norm_w, float eps, int compact, void* stream) {
    if (!x || !qweight || !rowmeta || !act || !act_sum || !norm_w) return MXQ_E_NULL;
    if (!shape_ok(N2, K) || N2 % 32 != 0 || K % 256 != 0) return MXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(rowmeta) || !aligned16(act) || ((uintptr_t)act_sum & 3)) return MXQ_E_ALIGN;
    return mxq_launch_gemv_swiglu_f16(x, qweight, rowmeta, act, act_sum, N2, K, norm_w, eps, compact != 0, (hipStream_t)stream);
}

int mxq_gemv_staged_f16(const void* x_staged, const void* x_sum, const void* qweight, const void* rowmeta, void* y, int N, int K,
                        const void* residual, int compact, void* stream) {
    if (int e = linear_check(x_staged, qweight, rowmeta, y, 1, N, K)) return e;
    if (!x_sum) return MXQ_E_NULL;
    if ((uintptr_t)x_sum & 3) return MXQ_E_ALIGN;
    return mxq_launch_gemv_staged_f16(x_staged, x_sum, qweight, rowmeta, y, N, K, residual, compact != 0, (hipStream_t)stream);
}

int mxq_attn_decode_f16(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* cos_t,
                        const void* sin_t, void* out, int heads, int head_dim, int max_ctx, void* stream) {
    if (!qkv || !k_cache || !v_cache || !pos || !cos_t || !sin_t || !out) return MXQ_E_NULL;
    if (heads <= 0 || head_dim != 128 || max_ctx <= 0 || max_ctx > 32768) return MXQ_E_SHAPE;
    return mxq_launch_attn_decode_f16(qkv, k_cache, v_cache, pos, cos_t, sin_t, out, heads, head_dim, max_ctx, 0,
                                      (hipStream_t)stream);
}

int mxq_rope_row_f32(const void* pos, const void* cos_t, const void* sin_t, void* row, int half_dim, int max_ctx, void* stream) {
    if (!pos || !cos_t || !sin_t || !row) return MXQ_E_NULL;
    if (half_dim <= 0 || half_dim > 4096 || max_ctx <= 0) return MXQ_E_SHAPE;
    return mxq_launch_rope_row_f32(pos, cos_t, sin_t, row, half_dim, max_ctx, nullptr, nullptr, 0, 0, nullptr, (hipStream_t)stream);
}

int mxq_embed_rope_row(const void* token, const void* embed, int vocab, int hidden, void* h_out, const void* pos, const void* cos_t,
                       const void* sin_t, void* row, int half_dim, int max_ctx, void* stream) {
    if (!token || !embed || !h_out || !pos || !cos_t || !sin_t || !row) return MXQ_E_NULL;
    if (vocab <= 0 || hidden <= 0 || hidden % 8 != 0 || half_dim <= 0 || half_dim > 4096 || max_ctx <= 0) return MXQ_E_SHAPE;
    if (!aligned16(embed) || !aligned16(h_out)) return MXQ_E_ALIGN;
    return mxq_launch_rope_row_f32(pos, cos_t, sin_t, row, half_dim, max_ctx, token, embed, vocab, hidden, h_out, (hipStream_t)stream);
}

size_t mxq_attn_split_workspace_bytes(int heads, int splits) {
    return heads > 0 && splits > 0 ? ::mxq_attn_split_workspace_bytes_impl(heads, splits) : 0;
}

int mxq_attn_decode_split_f16(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* rope_row, void* out,
                              int heads, int head_dim, int max_ctx, int splits, void* workspace, void* stream) {
    if (!qkv || !k_cache || !v_cache || !pos || !rope_row || !out || !workspace) return MXQ_E_NULL;
    if (heads <= 0 || head_dim != 128 || max_ctx <= 0 || max_ctx > 32768 || splits < 1 || splits > 16) return MXQ_E_SHAPE;
    if (!aligned16(workspace)) return MXQ_E_ALIGN;
    return mxq_launch_attn_decode_split_f16(qkv, k_cache, v_cache, pos, rope_row, out, heads, head_dim, max_ctx, splits, workspace,
                                            (hipStream_t)stream);
}

int mxq_attn_decode_row_f16(const void* qkv, void* k_cache, void* v_cache, const void* pos, const void* rope_row, void* out,
                            int heads, int head_dim, int max_ctx, void* stream) {
    if (!qkv || !k_cache || !v_cache || !pos || !rope_row || !out) return MXQ_E_NULL;
    if (heads <= 0 || head_dim != 128 || max_ctx <= 0 || max_ctx > 32768) return MXQ_E_SHAPE;
    return mxq_launch_attn_decode_f16(qkv, k_cache, v_cache, pos, rope_row, (const float*)rope_row + head_dim / 2, out, heads,
                                      head_dim, max_ctx, 1, (hipStream_t)stream);
}

int mxq_lmhead_argmax_f16(const void* h, const void* norm_w, float eps, const void* w, int V, int K, void* part,
                          int part_slots, void* token, void* stream) {
    if (!h || !norm_w || !w || !part || !token) return MXQ_E_NULL;
    if (V <= 0 || K != 4096 || part_slots < 1) return MXQ_E_SHAPE;
    if (!aligned16(h) || !aligned16(norm_w) || !aligned16(w)) return MXQ_E_ALIGN;
    return mxq_launch_lmhead_argmax_f16(h, norm_w, eps, w, V, K, part, part_slots, token, nullptr, nullptr, 0, (hipStream_t)stream);
}

int mxq_lmhead_argmax_advance_f16(const void* h, const void* norm_w, float eps, const void* w, int V, int K, void* part,
                                  int part_slots, void* token, void* pos, void* generated, int max_generated, void* stream) {
    if (!h || !norm_w || !w || !part || !token || !pos) return MXQ_E_NULL;
    if (V <= 0 || K != 4096 || part_slots < 1 || max_generated < 0 || (generated && max_generated == 0)) return MXQ_E_SHAPE;
    if (!aligned16(h) || !aligned16(norm_w) || !aligned16(w)) return MXQ_E_ALIGN;
    return mxq_launch_lmhead_argmax_f16(h, norm_w, eps, w, V, K, part, part_slots, token, pos, generated, max_generated,
                                        (hipStream_t)stream);
}

// ---- uniform layouts of the config-5 sweep ---------------------------------------------------

size_t mxq_qweight_bytes_layout(int N, int K, int layout) {
    if (!shape_ok(N, K) || !layout_ok(layout)) return 0;
    return (size_t)(N / 16) * (K / 64) * mxq_layout_blk_dw(layout) * 4;
}

int mxq_quantize_pack_layout(const void* W, int w_dtype, void* qweight, void* rowmeta, int N, int K, int layout,
                             void* stream) {
    if (!W || !qweight || !rowmeta) return MXQ_E_NULL;
    // (compact metadata is derived from the exact form: mxq_quantize_pack, then mxq_compact)
    if (!shape_ok(N, K) || !layout_ok(layout) || layout == MXQ_LAYOUT_MIXEDC) return MXQ_E_SHAPE;
    if (!dtype_ok(w_dtype)) return MXQ_E_DTYPE;
    if (!aligned16(W) || !aligned16(qweight) || !aligned16(rowmeta)) return MXQ_E_ALIGN;
    if (layout == MXQ_LAYOUT_MIXED)
        return mxq_launch_quantize_pack(W, w_dtype, nullptr, qweight, rowmeta, N, K, (hipStream_t)stream);
    return mxq_launch_quantize_uniform(W, w_dtype, qweight, rowmeta, N, K, layout, (hipStream_t)stream);
}

int mxq_expand_layout(const void* qweight, const void* rowmeta, void* w16, uint8_t* codes, uint8_t* sc, float* zero,
                      float* qs, float* qz, int N, int K, int layout, void* stream) {
    if (!qweight || !rowmeta || (!w16 && !codes)) return MXQ_E_NULL;
    if (codes && (!sc || !zero || !qs || !qz)) return MXQ_E_NULL;
    if (!shape_ok(N, K) || (layout != MXQ_LAYOUT_W2G16 && layout != MXQ_LAYOUT_W4ROW)) return MXQ_E_SHAPE;
    if (!aligned16(qweight) || !aligned16(rowmeta) || (w16 && !aligned16(w16))) return MXQ_E_ALIGN;
    return mxq_launch_uniform_expand(qweight, rowmeta, w16, codes, sc, zero, qs, qz, N, K, layout,
                                     (hipStream_t)stream);
}

int mxq_gemv_f16_layout(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                        int layout, void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (!layout_ok(layout) || M > 4) return MXQ_E_SHAPE;
    return mxq_launch_gemv_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, (hipStream_t)stream);
}

int mxq_gemm_f16_layout(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                        int layout, void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (!layout_ok(layout)) return MXQ_E_SHAPE;
    return mxq_launch_gemm8_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, nullptr, 0, (hipStream_t)stream);
}

// y = x . w16^T on a dense fp16 weight: variant 0 = by shape, 1 = the 256 x 128 kernel (gemm8.hip, dense instantiation),
// 2 = the 256 x 256 kernel (dense256.hip; MXQ_E_SHAPE when the K-tile count is odd)
static int dense_f16(const void* x, const void* w16, void* y, int M, int N, int K, int variant, hipStream_t stream) {
    if (variant != 1) {
        const int e = mxq_launch_dense256_f16(x, w16, y, M, N, K, variant == 2, stream);
        if (e != MXQ_NOT_MY_SHAPE) return e;
        if (variant == 2) return MXQ_E_SHAPE;
    }
    return mxq_launch_gemm8_dense_f16(x, w16, y, M, N, K, stream);
}

size_t mxq_hoist_scratch_bytes(int N, int K) { return shape_ok(N, K) ? (size_t)N * K * 2 : 0; }

int mxq_linear_f16_hoisted(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                           int layout, void* w16_scratch, size_t scratch_bytes, void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (!layout_ok(layout)) return MXQ_E_SHAPE;
    if (!w16_scratch) return MXQ_E_NULL;
    if (!aligned16(w16_scratch)) return MXQ_E_ALIGN;
    if (scratch_bytes < (size_t)N * K * 2) return MXQ_E_SHAPE;
    int e;
    if (layout == MXQ_LAYOUT_MIXED || layout == MXQ_LAYOUT_MIXEDC)
        e = mxq_launch_dequant_f16(qweight, rowmeta, w16_scratch, N, K, layout == MXQ_LAYOUT_MIXEDC, (hipStream_t)stream);
    else
        e = mxq_launch_uniform_expand(qweight, rowmeta, w16_scratch, nullptr, nullptr, nullptr, nullptr, nullptr, N, K,
                                      layout, (hipStream_t)stream);
    if (e) return e;
    return dense_f16(x, w16_scratch, y, M, N, K, 0, (hipStream_t)stream);
}

int mxq_dense_f16(const void* x, const void* w16, void* y, int M, int N, int K, int variant, void* stream) {
    if (!x || !w16 || !y) return MXQ_E_NULL;
    if (M < 0 || !shape_ok(N, K)) return MXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(w16) || !aligned16(y)) return MXQ_E_ALIGN;
    if (variant < 0 || variant > 2) return MXQ_E_SHAPE;
    if (M == 0) return 0;
    return dense_f16(x, w16, y, M, N, K, variant, (hipStream_t)stream);
}

// Token counts the mid-M split-K kernel (midm.hip) may serve: above the skinny kernel's range, below the point where the
// prefill kernel's 256 x 128 tiles win again (profiles/r03_midM.txt).  Since round 4 the fused kernel's 128- / 64-token builds
// (gemm8h_mode / gemm8q_mode below) take most of that range when a full workspace is given: what is left to this kernel is
// 21 / 41 .. 64 tokens on launches of <= 64 tiles (and up to 256 tokens where the workspace is too small for those builds,
// or absent).  The whole map against hipBLASLt: profiles/r04_dispatch_map.txt.
static const int MIDM_MAX_TOKENS = 256;
// Token count from which hoisting the dequant out of the token loop (dequant pass + dense 256 x 256 kernel) beats the
// fused kernel on the Llama shapes (tools/ab_gemm.py; profiles/r03_dense256.txt)
static const int HOIST_MIN_TOKENS = 4096;
// ... and the token count up to which the skinny kernel (one workgroup per 16-row block, every wave reads all of x from
// L2) still beats it: its time grows with tokens x weight size, the split-K kernel's is flat up to 64 tokens.  Measured
// crossovers (profiles/r03_midM.txt): ~44 tokens at 4096^2, ~22 at 11008 x 4096 and 4096 x 11008.
static int skinny_max_tokens(int N, int K) { return (int64_t)N * K > ((int64_t)24 << 20) ? 20 : 40; }
// ... and where the fused kernel's 128-token build (gemm8h.hip: 128 x 128 tiles) beats both neighbours, beyond 64 tokens:
//   * launches of <= 64 such tiles (a tile would be shared by >= 4 CUs): SLICES mode -- K cut into one slice per idle CU,
//     fp32 slabs through the workspace, a combine launch: the mid-M kernel's schedule with the fused kernel's wave roles.
//     128 tokens x 4096^2 19.3 us (mid-M kernel 21.1, hipBLASLt fp16 19.5), 256 tokens 22.4 (26.6), 4096 x 11008 at
//     128 / 256 tokens 23.8 / 34.8 (25.5 / 40.7);
//   * launches of 65 .. 176 tiles: ONE launch, stream-K over the otherwise idle CUs (at most 4 CUs per tile): gate/up
//     (11008 x 4096) at 128 / 192 / 256 tokens 25.2 / 33.4 / 34.8 us against 30.7 / 46.2 / 47.2 (mid-M kernel); 4096^2 at
//     384 / 512 tokens 26.2 / 28.8 against 34.9 / 35.7 (256-token tile).
// Same-process A/B on the Llama shapes: profiles/r04_gemm8h.txt.  Returns 0 (neither), 1 (stream-K), 2 (slices).
static int gemm8h_mode(int M, int N, size_t ws_bytes) {
    if (M <= 64 || ws_bytes < mxq_gemm8h_workspace_bytes()) return 0;
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    return tiles <= 64 ? 2 : tiles <= 176 ? 1 : 0;
}
// Up to 64 tokens the same again with a 64-token tile (gemm8q.hip), where it wins: 65 .. 176 tiles of 64 x 128 in stream-K
// mode -- gate/up at 21-64 tokens: 20.6-20.9 us against 23.6-23.9 (skinny / mid-M kernel); with few tiles the mid-M kernel's
// 64-token form is as fast (4096^2 at 64 tokens 13.6 against 14.5) and stays.  `no_midm`: the uniform layouts, which that
// kernel does not serve, take the slices mode there (4096^2 at 49-64 tokens: 14.5 us against 30 on the 256-token tile).
static int gemm8q_mode(int M, int N, size_t ws_bytes, bool no_midm) {
    if (M > 64 || ws_bytes < mxq_gemm8q_workspace_bytes()) return 0;
    const int tiles = (N + 127) / 128;
    return tiles <= 64 ? (no_midm ? 2 : 0) : tiles <= 176 ? 1 : 0;
}
static int small_tile_launch(int M, int mode, const void* x, const void* qweight, const void* rowmeta, void* y, int N, int K,
                             int layout, void* workspace, size_t ws_bytes, hipStream_t stream) {
    // <= 64 tiles of 128 x 128 and a short K (<= 96 K-steps): the fp32 partial tiles are the cost, and the 128 x 64 tile
    // (gemm8n.hip) halves them for the same number of workgroups -- 4096^2 at 65-128 tokens 16.0-16.2 us against 18.4-18.6
    // (its slices mode: <= 64 tiles of 128 x 64), at 192 / 256 tokens 20.5 / 21.0 against 22.2 / 22.3 (stream-K mode); with
    // K = 11008 its longer K loop costs more than the bytes save (24-25 against 22-23 us): profiles/r04_gemm8h.txt
    if (M > 64 && mode == 2 && K <= 6144) {
        const int tiles64 = ((M + 127) / 128) * ((N + 63) / 64);
        return tiles64 <= 64
                   ? mxq_launch_gemm8n_slices_f16(x, qweight, rowmeta, y, M, N, K, layout, workspace, ws_bytes, 0, stream)
                   : mxq_launch_gemm8n_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, workspace, ws_bytes, stream);
    }
    if (M <= 64)
        return mode == 2 ? mxq_launch_gemm8q_slices_f16(x, qweight, rowmeta, y, M, N, K, layout, workspace, ws_bytes, 0, stream)
                         : mxq_launch_gemm8q_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, workspace, ws_bytes, stream);
    return mode == 2 ? mxq_launch_gemm8h_slices_f16(x, qweight, rowmeta, y, M, N, K, layout, workspace, ws_bytes, 0, stream)
                     : mxq_launch_gemm8h_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, workspace, ws_bytes, stream);
}

// The workspace-free schedule of EVERY layout (mxq_linear_f16, and the _ws entries when `workspace` is NULL): the K range is
// never split.  Streaming GEMV (<= 4 tokens), skinny MFMA kernel as far as it goes (mixed layouts 64 tokens: 19.5 us at
// 4096^2 where the 128 x 128-tile kernel took 57; uniform layouts 48), then -- mixed layouts -- the mid-M kernel with one
// slice per tile up to 256 tokens (44-54 us at 4096^2: use a _ws entry where it matters; a shape beyond its 32-bit offsets
// falls through), and the prefill kernel on whole tiles beyond.
static int linear_noworkspace(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, int layout,
                              hipStream_t stream) {
    const bool mixed = mxq_layout_is_mixed(layout);
    if (M <= 4) return mxq_launch_gemv_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, stream);
    if (M <= (mixed ? 64 : 48)) return mxq_launch_skinny_f16(x, qweight, rowmeta, y, M, N, K, layout, stream);
    if (mixed && M <= MIDM_MAX_TOKENS) {
        const int e = mxq_launch_midm_f16(x, qweight, rowmeta, y, M, N, K, layout, nullptr, 0, 0, 0, stream);
        if (e != MXQ_E_SHAPE) return e;
    }
    if (layout == MXQ_LAYOUT_MIXED) return mxq_launch_gemm_f16(x, qweight, rowmeta, y, M, N, K, stream);
    return mxq_launch_gemm8_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, nullptr, 0, stream);
}

int mxq_linear_f16(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    return linear_noworkspace(x, qweight, rowmeta, y, M, N, K, MXQ_LAYOUT_MIXED, (hipStream_t)stream);
}

size_t mxq_gemm_workspace_bytes(void) { return mxq_gemm8_workspace_bytes(); }

// Bytes of workspace the dispatch of mxq_linear_f16_auto can use for THIS call (0: the call never touches a workspace --
// GEMV, skinny kernel, hoisted mode with a scratch): what a caller that owns one workspace per captured graph allocates
// instead of the maximum (mxq_gemm_workspace_bytes()).  Mirrors the dispatch above.
size_t mxq_linear_workspace_need(int M, int N, int K, int layout, int hoisting) {
    if (!shape_ok(N, K) || M <= 0 || !layout_ok(layout)) return 0;
    if (M <= 4 || (hoisting && M >= HOIST_MIN_TOKENS)) return 0;
    const bool mixed = mxq_layout_is_mixed(layout);
    if (M <= (mixed ? skinny_max_tokens(N, K) : 48)) return 0;
    const size_t full = mxq_gemm8_workspace_bytes();
    if (M > 64 ? gemm8h_mode(M, N, full) : gemm8q_mode(M, N, full, layout == MXQ_LAYOUT_W2G16 || layout == MXQ_LAYOUT_W4ROW)) {
        size_t need = M > 64 ? mxq_gemm8h_workspace_bytes() : mxq_gemm8q_workspace_bytes();
        if (M > 64 && mxq_gemm8n_workspace_bytes() > need) need = mxq_gemm8n_workspace_bytes();
        return need;
    }
    if (mixed && M <= MIDM_MAX_TOKENS) {   // the mid-M kernel's partial tiles: one slice per idle CU, at most
        const size_t tiles = (size_t)((M + 127) / 128) * ((N + 127) / 128);
        size_t slabs = tiles > 256 ? tiles : 256 + tiles;
        return (size_t)65536 + slabs * 128 * 128 * sizeof(float);
    }
    return full;
}

// hipStreamGetCaptureInfo through THIS library's HIP runtime (the process's own: a caller that dlopens "libamdhip64.so" by
// name may load a second runtime next to the one its framework bundles): *active = 1 and *id = the capture sequence's
// id while `stream` is being captured, else *active = 0.
int mxq_stream_capture_id(void* stream, int* active, unsigned long long* id) {
    if (!active || !id) return MXQ_E_NULL;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    const hipError_t e = hipStreamGetCaptureInfo((hipStream_t)stream, &st, &cid);
    if (e != hipSuccess) return (int)e;
    *active = st == hipStreamCaptureStatusActive;
    *id = *active ? cid : 0;
    return 0;
}

// Clock stamps (measurement helper): workgroup b of 8 -- one per XCD under the round-robin workgroup placement -- writes
// {s_memtime (shader-clock ticks), s_memrealtime (100 MHz), XCC id, 0} as 4 x u64 at out[4 b].  Two stamps on one stream
// around a timed region give the shader clock the chip held in between: d(memtime) / d(memrealtime) x 100 MHz per XCD
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).
__global__ void mxq_clock_stamp_kernel(unsigned long long* out) {
    if (threadIdx.x != 0) return;
    const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15u;     // HW_REG_XCC_ID[3:0]
    out[4 * blockIdx.x + 0] = t;
    out[4 * blockIdx.x + 1] = r;
    out[4 * blockIdx.x + 2] = xcc;
    out[4 * blockIdx.x + 3] = 0;
}

int mxq_clock_stamp(void* out32_u64, void* stream) {
    if (!out32_u64) return MXQ_E_NULL;
    mxq_clock_stamp_kernel<<<8, 64, 0, (hipStream_t)stream>>>((unsigned long long*)out32_u64);
    return (int)hipGetLastError();
}

// The stream-K kernels' status words (csrc/gemm8.hip, SK_STATUS_OFF: the last 16 bytes of the 64-KiB head).  The ONE entry
// that synchronises: it waits for `stream`, then copies the four ints to the host.
int mxq_workspace_status(const void* workspace, size_t workspace_bytes, int* status4, void* stream) {
    if (!workspace || !status4) return MXQ_E_NULL;
    if (workspace_bytes < 65536) return MXQ_E_SHAPE;
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpy(status4, (const char*)workspace + 65536 - 16, 16, hipMemcpyDeviceToHost);
}

static int gemm_ws(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                   void* workspace, size_t ws_bytes, hipStream_t stream) {
    // with a workspace the wave-specialised kernel wins at every token count (its stream-K split keeps all
    // CUs busy when there are few tiles: 64 tokens x 4096^2 34 us vs 57 us for the 128-row-tile kernel);
    // without one (mxq_gemm_f16_ws variant 0 only: mxq_linear_f16_ws forwards to mxq_linear_f16) the 128 x 128 kernel
    if (!workspace) return mxq_launch_gemm_f16(x, qweight, rowmeta, y, M, N, K, stream);
    return mxq_launch_gemm8_f16(x, qweight, rowmeta, y, M, N, K, workspace, ws_bytes, 0, stream);
}

// the mid-M kernel needs the workspace beyond its 64-KiB head for at least two slices' partial tiles per output tile;
// with less it would run unsplit (44-54 us at 4096^2 where the prefill kernel's stream-K tail takes 34)
static bool midm_ws_ok(int M, int N, size_t ws_bytes) {
    const size_t tiles = (size_t)((M + 127) / 128) * ((N + 127) / 128);
    return ws_bytes >= (size_t)65536 + tiles * 2 * 128 * 128 * sizeof(float);
}

int mxq_linear_f16_ws(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace) return mxq_linear_f16(x, qweight, rowmeta, y, M, N, K, stream);   // the workspace-free dispatch
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (!aligned16(workspace)) return MXQ_E_ALIGN;
    if (M <= 4) return mxq_launch_gemv_f16(x, qweight, rowmeta, y, M, N, K, (hipStream_t)stream);
    if (M <= skinny_max_tokens(N, K))
        return mxq_launch_skinny_f16(x, qweight, rowmeta, y, M, N, K, MXQ_LAYOUT_MIXED, (hipStream_t)stream);
    if (const int mode = M > 64 ? gemm8h_mode(M, N, workspace_bytes) : gemm8q_mode(M, N, workspace_bytes, false))
        return small_tile_launch(M, mode, x, qweight, rowmeta, y, N, K, MXQ_LAYOUT_MIXED, workspace, workspace_bytes,
                                 (hipStream_t)stream);
    if (M <= MIDM_MAX_TOKENS && midm_ws_ok(M, N, workspace_bytes)) {
        const int e = mxq_launch_midm_f16(x, qweight, rowmeta, y, M, N, K, MXQ_LAYOUT_MIXED, workspace, workspace_bytes, 0, 0,
                                          (hipStream_t)stream);
        if (e != MXQ_E_SHAPE) return e;      // a shape beyond its 32-bit offsets: the prefill kernel takes it
    }
    return gemm_ws(x, qweight, rowmeta, y, M, N, K, workspace, workspace_bytes, (hipStream_t)stream);
}

int mxq_linear_f16_layout_ws(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                             int layout, void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (!layout_ok(layout)) return MXQ_E_SHAPE;
    if (workspace && !aligned16(workspace)) return MXQ_E_ALIGN;
    if (!workspace) return linear_noworkspace(x, qweight, rowmeta, y, M, N, K, layout, (hipStream_t)stream);
    if (layout == MXQ_LAYOUT_MIXED)
        return mxq_linear_f16_ws(x, qweight, rowmeta, y, M, N, K, workspace, workspace_bytes, stream);
    if (M <= 4) return mxq_launch_gemv_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, (hipStream_t)stream);
    // every layout: the skinny MFMA kernel up to the token count where the split-K / prefill kernels overtake it
    if (M <= (layout == MXQ_LAYOUT_MIXEDC ? skinny_max_tokens(N, K) : 48))
        return mxq_launch_skinny_f16(x, qweight, rowmeta, y, M, N, K, layout, (hipStream_t)stream);
    if (const int mode = M > 64 ? gemm8h_mode(M, N, workspace_bytes)
                                : gemm8q_mode(M, N, workspace_bytes, layout != MXQ_LAYOUT_MIXEDC))
        return small_tile_launch(M, mode, x, qweight, rowmeta, y, N, K, layout, workspace, workspace_bytes, (hipStream_t)stream);
    if (layout == MXQ_LAYOUT_MIXEDC) {
        if (M <= MIDM_MAX_TOKENS && midm_ws_ok(M, N, workspace_bytes)) {
            const int e = mxq_launch_midm_f16(x, qweight, rowmeta, y, M, N, K, layout, workspace, workspace_bytes, 0, 0,
                                              (hipStream_t)stream);
            if (e != MXQ_E_SHAPE) return e;
        }
    }
    return mxq_launch_gemm8_layout_f16(x, qweight, rowmeta, y, M, N, K, layout, workspace, workspace_bytes,
                                       (hipStream_t)stream);
}

int mxq_hoist_min_tokens(void) { return HOIST_MIN_TOKENS; }

int mxq_linear_f16_auto(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K, int layout,
                        void* workspace, size_t workspace_bytes, void* w16_scratch, size_t scratch_bytes, void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (!layout_ok(layout)) return MXQ_E_SHAPE;
    if (w16_scratch && !aligned16(w16_scratch)) return MXQ_E_ALIGN;
    if (M >= HOIST_MIN_TOKENS && w16_scratch && scratch_bytes >= (size_t)N * K * 2)
        return mxq_linear_f16_hoisted(x, qweight, rowmeta, y, M, N, K, layout, w16_scratch, scratch_bytes, stream);
    return mxq_linear_f16_layout_ws(x, qweight, rowmeta, y, M, N, K, layout, workspace, workspace_bytes, stream);
}

int mxq_gemm_f16_ws(const void* x, const void* qweight, const void* rowmeta, void* y, int M, int N, int K,
                    int variant, void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = linear_check(x, qweight, rowmeta, y, M, N, K)) return e;
    if (workspace && !aligned16(workspace)) return MXQ_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {   // enum mxq_gemm_variant (include/mxq_hip.h)
        case MXQ_GEMM_DEFAULT: return gemm_ws(x, qweight, rowmeta, y, M, N, K, workspace, workspace_bytes, st);
        case MXQ_GEMM_TILE128: return mxq_launch_gemm1_f16(x, qweight, rowmeta, y, M, N, K, st);
        case MXQ_GEMM_FUSED256:
        case MXQ_GEMM_FUSED256_SPLIT:
            return mxq_launch_gemm8_f16(x, qweight, rowmeta, y, M, N, K, workspace, workspace_bytes, variant == MXQ_GEMM_FUSED256_SPLIT, st);
        case MXQ_GEMM_MIDM:
            return mxq_launch_midm_f16(x, qweight, rowmeta, y, M, N, K, MXQ_LAYOUT_MIXED, workspace, workspace_bytes, 0, 0, st);
        case MXQ_GEMM_FUSED128:
        case MXQ_GEMM_FUSED128_SPLIT:
            return mxq_launch_gemm8h_f16(x, qweight, rowmeta, y, M, N, K, workspace, workspace_bytes, variant == MXQ_GEMM_FUSED128_SPLIT, st);
        case MXQ_GEMM_FUSED128_SLICES:
            return mxq_launch_gemm8h_slices_f16(x, qweight, rowmeta, y, M, N, K, MXQ_LAYOUT_MIXED, workspace, workspace_bytes, 0, st);
        case MXQ_GEMM_FUSED64_SPLIT:
            return mxq_launch_gemm8q_f16(x, qweight, rowmeta, y, M, N, K, workspace, workspace_bytes, 1, st);
        case MXQ_GEMM_FUSED64_SLICES:
            return mxq_launch_gemm8q_slices_f16(x, qweight, rowmeta, y, M, N, K, MXQ_LAYOUT_MIXED, workspace, workspace_bytes, 0, st);
        case MXQ_GEMM_FUSED128N64_SPLIT:
            return mxq_launch_gemm8n_f16(x, qweight, rowmeta, y, M, N, K, workspace, workspace_bytes, 1, st);
        case MXQ_GEMM_FUSED128N64_SLICES:
            return mxq_launch_gemm8n_slices_f16(x, qweight, rowmeta, y, M, N, K, MXQ_LAYOUT_MIXED, workspace, workspace_bytes, 0, st);
    }
    return MXQ_E_SHAPE;
}

int mxq_fakequant_fwd(const void* w, void* out, int rows, int cols, int num_bits, int dtype, void* stream) {
    if (!w || !out) return MXQ_E_NULL;
    if (rows <= 0 || cols <= 0 || cols % 64 != 0 || num_bits < 1 || num_bits > 31) return MXQ_E_SHAPE;
    if (!dtype_ok(dtype)) return MXQ_E_DTYPE;
    if (!aligned16(w) || !aligned16(out)) return MXQ_E_ALIGN;
    return mxq_launch_fakequant_fwd(w, out, rows, cols, num_bits, dtype, (hipStream_t)stream);
}

int mxq_actquant_group_fwd(const void* x, void* out, int64_t rows, int cols, int group, int num_bits, int symmetric,
                           int dtype, void* stream) {
    if (!x || !out) return MXQ_E_NULL;
    if (!dtype_ok(dtype)) return MXQ_E_DTYPE;
    const int vec = dtype == MXQ_DTYPE_F32 ? 4 : 8;
    const int lpg = group > 0 && group % vec == 0 ? group / vec : 0;
    if (rows <= 0 || cols <= 0 || cols % vec != 0 || lpg == 0 || lpg > 32 || (lpg & (lpg - 1)) != 0 || num_bits < 1 ||
        num_bits > 31 || rows * (int64_t)(cols / vec + lpg) >= ((int64_t)1 << 39))
        return MXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(out)) return MXQ_E_ALIGN;
    return mxq_launch_actquant_group(x, out, rows, cols, group, num_bits, symmetric, dtype, (hipStream_t)stream);
}

int mxq_actquant_fwd(const void* x, void* out, void* range_ws, int64_t n_seg, int64_t seg_len, int64_t period,
                     int64_t live, int num_bits, int symmetric, int dtype, void* stream) {
    if (!x || !out || !range_ws) return MXQ_E_NULL;
    if (!dtype_ok(dtype)) return MXQ_E_DTYPE;
    const int vec = dtype == MXQ_DTYPE_F32 ? 4 : 8;
    if (n_seg <= 0 || seg_len <= 0 || seg_len % vec != 0 || period <= 0 || live < 0 || num_bits < 1 || num_bits > 31)
        return MXQ_E_SHAPE;
    const int64_t per_chunk = (int64_t)256 * vec * 4;
    if (n_seg * ((seg_len + per_chunk - 1) / per_chunk) >= ((int64_t)1 << 31)) return MXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(out) || ((uintptr_t)range_ws & 7)) return MXQ_E_ALIGN;
    return mxq_launch_actquant_seg(x, out, range_ws, n_seg, seg_len, period, live, num_bits, symmetric, dtype,
                                   (hipStream_t)stream);
}

int mxq_fakequant_bwd(const void* grad_out, const void* w, void* grad_in, int64_t n, float lo, float hi, int dtype,
                      void* stream) {
    if (!grad_out || !w || !grad_in) return MXQ_E_NULL;
    if (!dtype_ok(dtype)) return MXQ_E_DTYPE;
    if (n <= 0 || n % (dtype == MXQ_DTYPE_F32 ? 4 : 8) != 0) return MXQ_E_SHAPE;
    if (!aligned16(grad_out) || !aligned16(w) || !aligned16(grad_in)) return MXQ_E_ALIGN;
    return mxq_launch_fakequant_bwd(grad_out, w, grad_in, n, lo, hi, dtype, (hipStream_t)stream);
}

int mxq_gemv_awq_f16(const void* x, const void* kernel, const void* scales, const void* zeros, void* y, int B,
                     int IC, int OC, int group_size, void* stream) {
    if (!x || !kernel || !scales || !zeros || !y) return MXQ_E_NULL;
    // the code matrix is read with 16-byte loads, one group index per 32 weights: rows of IC / 8 dwords must be
    // whole 16-byte units (IC % 32 == 0) and the matrix itself 16-byte aligned
    if (B <= 0 || OC <= 0 || IC <= 0 || IC % 32 != 0) return MXQ_E_SHAPE;
    if (group_size != 32 && group_size != 64 && group_size != 128) return MXQ_E_SHAPE;   // gemv_cuda.cu:371-397
    if (IC % group_size != 0) return MXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(kernel)) return MXQ_E_ALIGN;
    return mxq_launch_gemv_awq_f16(x, kernel, scales, zeros, y, B, IC, OC, group_size, (hipStream_t)stream);
}

int mxq_gemm_awq_f16(const void* x, const void* kernel, const void* scales, const void* zeros, void* y, int M, int IC, int OC,
                     int group_size, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !kernel || !scales || !zeros || !y) return MXQ_E_NULL;
    // the reference launcher's checks (gemm_cuda_gen.cu:447-454) + this kernel's 64-deep K-step
    if (M <= 0 || IC <= 0 || OC <= 0 || OC % 64 != 0 || OC % 8 != 0) return MXQ_E_SHAPE;
    if (group_size <= 0 || group_size % 32 != 0 || OC % group_size != 0) return MXQ_E_SHAPE;
    if (IC % 64 != 0 || IC % group_size != 0) return MXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(scales) || !aligned16(y) || ((uintptr_t)kernel & 3) || ((uintptr_t)zeros & 3)) return MXQ_E_ALIGN;
    if (workspace && !aligned16(workspace)) return MXQ_E_ALIGN;
    if (workspace && workspace_bytes < 65536) return MXQ_E_SHAPE;
    // Up to 32 tokens: a streaming kernel (one workgroup per 128 channels x K slice, last arriver sums the slices).  Beyond -- and as
    // its fallback -- the fused kernel's dispatch by tile count (gemm8h_mode / gemm8q_mode above) on this operand format's builds:
    // 64-token tiles, then 128-token tiles while there are at most 176 of them (<= 1024 tokens) -- <= 64 tiles: every tile's K range cut so that tiles x slices fill the
    // chip, partial tiles summed by a combine launch; 65 .. 176 tiles: one launch, stream-K over the otherwise idle CUs (both need
    // the workspace) -- beyond that 256-token tiles, persistent, stream-K tail.
    hipStream_t st = (hipStream_t)stream;
    const int tn = (OC + 127) / 128;
    if (M <= 32) {     // few tokens: the streaming kernel (skinny_awq.hip), unless the K range does not fit its LDS unsliced
        const int e = mxq_launch_skinny_awq_f16(x, kernel, scales, zeros, y, M, IC, OC, group_size, workspace, workspace_bytes, st);
        if (e != MXQ_NOT_MY_SHAPE) return e;
    }
    if (M <= 64 && workspace && workspace_bytes >= mxq_gemm8aq_workspace_bytes() && tn <= 176)
        return mxq_launch_gemm8aq_f16(x, kernel, scales, zeros, y, M, IC, OC, group_size, workspace, workspace_bytes, tn <= 64 ? -1 : -2, st);
    if (M <= 64 && !workspace) return mxq_launch_gemm8aq_f16(x, kernel, scales, zeros, y, M, IC, OC, group_size, nullptr, 0, 0, st);
    const int t128 = ((M + 127) / 128) * tn;
    if (M <= 1024 && workspace && workspace_bytes >= mxq_gemm8ah_workspace_bytes() && t128 <= 176)
        return mxq_launch_gemm8ah_f16(x, kernel, scales, zeros, y, M, IC, OC, group_size, workspace, workspace_bytes, t128 <= 64 ? -1 : -2, st);
    return mxq_launch_gemm8a_f16(x, kernel, scales, zeros, y, M, IC, OC, group_size, workspace, workspace_bytes, 0, st);
}

int mxq_gemv_proto_f16(const void* x, const void* weight, const void* weight_last, const void* zeros_and_scales,
                       const void* scales_2nd, const void* zeros_2nd, const void* scales_4b, const void* zeros_4b,
                       void* y, int B, int IC, int OC, int group_size, void* stream) {
    if (!x || !weight || !weight_last || !zeros_and_scales || !scales_2nd || !zeros_2nd || !scales_4b || !zeros_4b ||
        !y)
        return MXQ_E_NULL;
    if (group_size != 16 || IC != 4096 || B <= 0 || OC <= 0 || OC % 8 != 0) return MXQ_E_SHAPE;
    if (!aligned16(x) || !aligned16(weight)) return MXQ_E_ALIGN;
    return mxq_launch_gemv_proto_f16(x, weight, weight_last, zeros_and_scales, scales_2nd, zeros_2nd, scales_4b,
                                     zeros_4b, y, B, IC, OC, (hipStream_t)stream);
}

}   // extern "C"
