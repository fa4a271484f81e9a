// gemm8a.hip with a 128-token tile (4 MFMA waves + 8 dequant waves), for launches of 65-256 tokens.
// Entry points: mxq_launch_gemm8ah_f16, mxq_gemm8ah_workspace_bytes (mxq_kernels.h).
#define MXQ_G8_AWQ 1
#define MXQ_G8_BM 128
#include "gemm8.hip"
