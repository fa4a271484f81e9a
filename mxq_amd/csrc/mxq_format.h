// MXQ packed weight format v1 ("B16x64") -- shared by every kernel and by the host.
//
// The reference has no packer; this layout is the build's own,
// derived from the lossless parameterisation of MXQGPT.fasterquant
// (reference mxq_quant/lib/mxqgpt.py:404-443, lib/quantizer.py:94-121):
//   per 64 input channels: 3 groups x 16 two-bit codes + 16 four-bit codes,
//   per (row, 2-bit group): 4-bit scale code + fp32 zero-point,
//   per (16-row block, 2-bit group): fp32 (qs, qz) of the second-order scale quantiser,
//   per row: 4-bit-arm scale code + fp32 zero-point; per 16-row block: (qs4, qz4).
//
// Weight W[N, K], N % 16 == 0, K % 64 == 0.  NC = K/64 chunks.
// qweight = [N/16][NC] blocks; one block covers 16 rows x 1 chunk (64 input channels) and
// is 144 dwords (576 B, 16-B aligned, 4.5 bit/weight):
//
//   C2 [g:3][r:16] u32   16 two-bit codes of group g, row r           dword   0
//   C4 [h:2][r:16] u32    8 four-bit codes (half h of the last 16)    dword  48
//   Z2 [g:3][r:16] f32   zero-point of that group                     dword  80
//   SC [r:16]      u16   scale codes: g0 | g1 << 4 | g2 << 8          dword 128
//   QQ [g:4][2]    f32   (qs, qz) of the 16-row block, g = 3 unused   dword 136
//
// A block is the unit both hot kernels consume: the prefill GEMM copies one block per
// (16 rows, K-step) into LDS with a single 36-lane global_load_lds_dwordx4 (every field
// is 16-B aligned), the decode GEMV streams 4 consecutive blocks (2304 contiguous bytes)
// per wave iteration with lane -> (r = lane & 15, chunk slot = lane >> 4).
//
// Bit order inside a code word is "byte-spread" so that a whole-word mask yields one
// code per byte (feeds v_perm_b32 / v_cvt_f32_ubyteN directly):
//   2-bit word: code of element k (0..15) sits at bit 8*(k&3) + 2*(k>>2)
//   4-bit word: code of element k (0..7)  sits at bit 8*(k&3) + 4*(k>>2)
//
// Two uniform layouts exist for the W2 / W4 / mixed sweep of BASELINE config 5 (same [N/16][K/64]
// block grid, same code-word bit order, same second-order scale coding):
//   W2G16: every 16-column quarter is a 2-bit group:  C2 [g:4][r:16] @0, Z2 [g:4][r:16] @64,
//          SC [r:16] u16 (4 nibbles) @128, QQ [g:4][2] @136 -- 144 dwords, 4.5 bit/weight.
//   W4ROW: 4-bit codes with ONE scale/zero per row (the mixed layout's 4-bit arm applied to all
//          columns): C4 [q:4][h:2][r:16] @0 -- 128 dwords, 4.0 bit/weight + rowmeta.
//
// rowmeta = [N] float4 {zero4, (float)scale_code4, qs4, qz4} (qs4/qz4 replicated over
// the 16 rows of a block).
#pragma once
#include <stdint.h>

#define MXQ_FORMAT_VERSION 1
#define MXQ_BLK_ROWS 16
#define MXQ_CHUNK 64
#define MXQ_BLK_DW 144
#define MXQ_BLK_BYTES 576
#define MXQ_OFF_C2 0
#define MXQ_OFF_C4 48
#define MXQ_OFF_Z2 80
#define MXQ_OFF_SC 128 /* dword offset; the field is u16[16] */
#define MXQ_OFF_QQ 136

#if defined(__HIPCC__)
#define MXQ_HD __host__ __device__ __forceinline__
#else
#define MXQ_HD inline
#endif

MXQ_HD int64_t mxq_blk_index(int row, int chunk, int K) {
    return (int64_t)(row / MXQ_BLK_ROWS) * (K / MXQ_CHUNK) + chunk;
}
MXQ_HD int mxq_bit2(int k) { return 8 * (k & 3) + 2 * (k >> 2); }   // k in 0..15
MXQ_HD int mxq_bit4(int k) { return 8 * (k & 3) + 4 * (k >> 2); }   // k in 0..7

// dword offsets inside a block
MXQ_HD int mxq_c2(int g, int r) { return MXQ_OFF_C2 + g * 16 + r; }
MXQ_HD int mxq_c4(int h, int r) { return MXQ_OFF_C4 + h * 16 + r; }
MXQ_HD int mxq_z2(int g, int r) { return MXQ_OFF_Z2 + g * 16 + r; }
MXQ_HD int mxq_sc_u16(int r) { return MXQ_OFF_SC * 2 + r; }   // u16 index
MXQ_HD int mxq_qq(int g) { return MXQ_OFF_QQ + g * 2; }

#define MXQ_LAYOUT_MIXED 0
#define MXQ_LAYOUT_W2G16 1
#define MXQ_LAYOUT_W4ROW 2
#define MXQ_LAYOUT_MIXEDC 3   /* the mixed layout with COMPACT metadata, see below */

// Compact metadata mode of the mixed layout ("format v2", SURVEY.md H2 / BASELINE.md section 3): the same codes,
// scale codes and (qs, qz), but the 2-bit zero-points stored as fp16 instead of fp32.  Block = 120 dwords = 480 B
// (3.75 bit/weight + rowmeta instead of 4.5):
//   C2  [g:3][r:16] u32  @ dword   0     (as in v1)
//   C4  [h:2][r:16] u32  @ dword  48     (as in v1)
//   Z2H [g:3][r:16] f16  @ dword  80     fp16(zero-point), round-to-nearest-even
//   SC  [r:16]      u16  @ dword 104
//   QQ  [g:4][2]    f32  @ dword 112     slot 3 unused
// The integer codes are those of the exact quantiser (computed with the fp32 zero-point), so the unpack stays
// bit-exact on them; the dequantised weight is fp16(scale * (q - float(fp16(zero)))), which costs ~4.4e-4 of the
// 1e-3 GEMM budget.  The 4-bit arm's per-row parameters (rowmeta) stay fp32.
#define MXQC_BLK_DW 120
#define MXQC_BLK_BYTES 480
#define MXQC_OFF_Z2H 80   /* dword offset; the field is f16[3][16] */
#define MXQC_OFF_SC 104   /* dword offset; the field is u16[16] */
#define MXQC_OFF_QQ 112
MXQ_HD int mxqc_z2_u16(int g, int r) { return MXQC_OFF_Z2H * 2 + g * 16 + r; }   // u16 index
MXQ_HD int mxqc_sc_u16(int r) { return MXQC_OFF_SC * 2 + r; }                    // u16 index
MXQ_HD int mxqc_qq(int g) { return MXQC_OFF_QQ + g * 2; }

MXQ_HD int mxq_layout_blk_dw(int layout) {
    return layout == MXQ_LAYOUT_W4ROW ? 128 : layout == MXQ_LAYOUT_MIXEDC ? MXQC_BLK_DW : 144;
}
MXQ_HD bool mxq_layout_is_mixed(int layout) { return layout == MXQ_LAYOUT_MIXED || layout == MXQ_LAYOUT_MIXEDC; }

#if defined(__HIPCC__) || defined(MXQ_HOST_HALF)
// Field access of the two metadata modes of the mixed layout behind one interface (C2 / C4 sit at the same
// offsets in both): zero-point as float, the row's scale-code word, the (qs, qz) dword offset.
template <bool COMPACT>
struct MxqMixed {
    static constexpr int BLK_DW = COMPACT ? MXQC_BLK_DW : MXQ_BLK_DW;
    static constexpr int BLK_BYTES = BLK_DW * 4;
    static MXQ_HD uint32_t scw(const uint32_t* tile, int r) {
        return ((const uint16_t*)tile)[COMPACT ? mxqc_sc_u16(r) : mxq_sc_u16(r)];
    }
    static MXQ_HD int qq(int g) { return COMPACT ? mxqc_qq(g) : mxq_qq(g); }
#if defined(__HIPCC__)
    static __device__ __forceinline__ float z2(const uint32_t* tile, int g, int r) {
        if constexpr (COMPACT) {
            const uint16_t h = ((const uint16_t*)tile)[mxqc_z2_u16(g, r)];
            return (float)__builtin_bit_cast(_Float16, h);
        } else {
            return __builtin_bit_cast(float, tile[mxq_z2(g, r)]);
        }
    }
#endif
};
#endif
MXQ_HD int mxq_w2_c2(int g, int r) { return g * 16 + r; }
MXQ_HD int mxq_w2_z2(int g, int r) { return 64 + g * 16 + r; }
MXQ_HD int mxq_w4_c4(int q, int h, int r) { return (q * 2 + h) * 16 + r; }
