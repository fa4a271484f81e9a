// MXQ packed weight format v1 ("T16x256") -- shared by every kernel and by the host.
//
// The reference has no packer; this layout is the build's own,
// derived from the lossless parameterisation of MXQGPT.fasterquant
// (reference mxq_quant/lib/mxqgpt.py:404-443, lib/quantizer.py:94-121):
//   per 64 input channels: 3 groups x 16 two-bit codes + 16 four-bit codes,
//   per (row, 2-bit group): 4-bit scale code + fp32 zero-point,
//   per (16-row block, 2-bit group): fp32 (qs, qz) of the second-order scale quantiser,
//   per row: 4-bit-arm scale code + fp32 zero-point; per 16-row block: (qs4, qz4).
//
// Weight W[N, K], N % 16 == 0, K % 64 == 0.  NC = K/64 chunks, NC4 = ceil(NC/4).
// qweight = [N/16][NC4] tiles; one tile covers 16 rows x 4 chunks (256 input channels)
// and is 568 dwords (2272 B, 16-B aligned):
//
//   C2 [g:3][cc:4][r:16] u32   16 two-bit codes of group g of chunk cc, row r
//   C4 [h:2][cc:4][r:16] u32    8 four-bit codes (half h) of chunk cc, row r
//   Z2 [g:3][cc:4][r:16] f32   zero-point of that group
//   SC [cc:4][r:16]      u16   scale codes: g0 | g1 << 4 | g2 << 8
//   QQ [cc:4][g:3][2]    f32   (qs, qz) of the 16-row block for that group
//
// Every field is [.. cc][r]-minor so that 64 lanes (r = lane & 15, cc = lane >> 4)
// read 256 contiguous bytes per load (decode GEMV), and 16 consecutive rows of one
// chunk are one 64-B segment (prefill GEMM, lane <-> row).
//
// Bit order inside a code word is "byte-spread" so that a whole-word mask yields one
// code per byte (feeds v_perm_b32 / v_cvt_f32_ubyteN directly):
//   2-bit word: code of element k (0..15) sits at bit 8*(k&3) + 2*(k>>2)
//   4-bit word: code of element k (0..7)  sits at bit 8*(k&3) + 4*(k>>2)
//
// rowmeta = [N] float4 {zero4, (float)scale_code4, qs4, qz4} (qs4/qz4 replicated over
// the 16 rows of a block).
//
// Rows/chunks beyond the matrix (chunk padding up to NC4*4) are all-zero tiles, which
// dequantise to exactly 0.
#pragma once
#include <stdint.h>

#define MXQ_FORMAT_VERSION 1
#define MXQ_TILE_ROWS 16
#define MXQ_TILE_CHUNKS 4
#define MXQ_CHUNK 64
#define MXQ_TILE_DW 568
#define MXQ_OFF_C2 0
#define MXQ_OFF_C4 192
#define MXQ_OFF_Z2 320
#define MXQ_OFF_SC 512 /* dword offset; the field is u16[64] */
#define MXQ_OFF_QQ 544

#if defined(__HIPCC__)
#define MXQ_HD __host__ __device__ __forceinline__
#else
#define MXQ_HD inline
#endif

MXQ_HD int mxq_nc4(int K) { return (K / MXQ_CHUNK + MXQ_TILE_CHUNKS - 1) / MXQ_TILE_CHUNKS; }
MXQ_HD int64_t mxq_tile_index(int row, int chunk, int K) {
    return (int64_t)(row / MXQ_TILE_ROWS) * mxq_nc4(K) + chunk / MXQ_TILE_CHUNKS;
}
MXQ_HD int mxq_bit2(int k) { return 8 * (k & 3) + 2 * (k >> 2); }   // k in 0..15
MXQ_HD int mxq_bit4(int k) { return 8 * (k & 3) + 4 * (k >> 2); }   // k in 0..7

// dword offsets inside a tile
MXQ_HD int mxq_c2(int g, int cc, int r) { return MXQ_OFF_C2 + (g * 4 + cc) * 16 + r; }
MXQ_HD int mxq_c4(int h, int cc, int r) { return MXQ_OFF_C4 + (h * 4 + cc) * 16 + r; }
MXQ_HD int mxq_z2(int g, int cc, int r) { return MXQ_OFF_Z2 + (g * 4 + cc) * 16 + r; }
MXQ_HD int mxq_sc_u16(int cc, int r) { return MXQ_OFF_SC * 2 + cc * 16 + r; }   // u16 index
MXQ_HD int mxq_qq(int cc, int g) { return MXQ_OFF_QQ + (cc * 3 + g) * 2; }
