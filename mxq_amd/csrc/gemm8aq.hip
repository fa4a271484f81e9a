// gemm8a.hip with a 64-token tile (2 MFMA waves + 8 dequant waves), for launches of few tokens: K slices + combine launch.
// Entry points: mxq_launch_gemm8aq_f16, mxq_gemm8aq_workspace_bytes (mxq_kernels.h).
#define MXQ_G8_AWQ 1
#define MXQ_G8_BM 64
#include "gemm8.hip"
