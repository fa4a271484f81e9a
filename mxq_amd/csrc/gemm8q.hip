// The fused prefill kernel with a 64-token tile (2 MFMA waves + 8 dequant waves): gemm8.hip compiled a third time.
// Entry points: mxq_launch_gemm8q_f16 / _layout_f16 / _slices_f16, mxq_gemm8q_workspace_bytes (mxq_kernels.h).
#define MXQ_G8_BM 64
#include "gemm8.hip"
