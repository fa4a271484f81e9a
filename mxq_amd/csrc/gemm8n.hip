// The fused prefill kernel with a 128-token x 64-channel tile (4 MFMA waves of 32 x 64 + 4 quarter-row dequant waves): gemm8.hip
// compiled a fourth time.  Half the fp32 partial-tile bytes per workgroup of the 128 x 128 build: for launches of few tiles.
// Entry points: mxq_launch_gemm8n_f16 / _layout_f16 / _slices_f16, mxq_gemm8n_workspace_bytes (mxq_kernels.h).
#define MXQ_G8_BM 128
#define MXQ_G8_BN 64
#include "gemm8.hip"
