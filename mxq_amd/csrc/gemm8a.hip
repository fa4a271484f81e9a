// The fused prefill kernel's skeleton (gemm8.hip: MFMA waves + LDS-DMA x ring + dequant waves + persistent tiles + stream-K tail)
// with dequant waves for the OPERANDS of the reference's gemm_forward_cuda (gemm_cuda.h:3-4, gemm_cuda_gen.cu:28-478,
// dequantize.cuh:15-78): 256-token tile.  Entry points: mxq_launch_gemm8a_f16, mxq_gemm8a_workspace_bytes (mxq_kernels.h).
#define MXQ_G8_AWQ 1
#include "gemm8.hip"
