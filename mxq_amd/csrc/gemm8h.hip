// The fused prefill kernel with a 128-token tile (4 MFMA waves + 4 dequant waves): gemm8.hip compiled a second time.
// Entry points: mxq_launch_gemm8h_f16 / _layout_f16, mxq_gemm8h_workspace_bytes (mxq_kernels.h).
#define MXQ_G8_BM 128
#include "gemm8.hip"
