// The C-ABI block this experiment added to include/mxq_hip.h (removed with it)
#pragma once
#include <stddef.h>
/* A CHAIN of up to MXQ_CHAIN_MAX_OPS one-token fused GEMVs in ONE launch (decode): op i + 1 reads op i's output vector --
 * e.g. o_proj (+ residual) -> RMSNorm + gate|up -> SwiGLU + down (+ residual) -> RMSNorm + the next layer's q|k|v, the
 * chain of Linears of LLM-QAT/models/modeling_llama_quant.py:262-291 / 323-360.  Each op has mxq_gemv_fused_f16's
 * arguments and its results bit for bit; what the chain removes is the launch boundary between dependent GEMVs (~5 us
 * each): a later op's workgroups fetch their weights while the earlier op is still running and wait for its output on
 * device counters (csrc/gemv_chain.hip: why this cannot dead-lock).  ops: HOST array; sync_ws: device memory of
 * mxq_gemv_chain_ws_bytes() bytes that the caller ZEROES BEFORE EVERY LAUNCH on the same stream (hipMemsetAsync, or
 * any kernel of its own that runs before: the counters only count up inside a launch); its LAST int becomes non-zero if
 * a wait ever ran out of its poll budget (a quarter of a second; the results of that launch are then invalid).
 * MXQ_E_SHAPE: n out of range, N % 16 / K % 64, or K beyond the kernel's LDS staging (K <= 24576). */
#define MXQ_CHAIN_MAX_OPS 4
typedef struct {
    const void* x;         /* fp16 [K] ([2K] = (gate, up) for prologue 2); op i > 0: normally op i-1's y */
    const void* qweight;   /* packed weight [N, K] (exact metadata, or compact if the call says so) */
    const void* rowmeta;
    void* y;               /* fp16 [N] */
    const void* norm_w;    /* prologue 1: fp16 [K] */
    const void* residual;  /* nullable fp16 [N] */
    int N, K, prologue;
    float eps;
} mxq_chain_op_t;
size_t mxq_gemv_chain_ws_bytes(void);
int mxq_gemv_chain_f16(const mxq_chain_op_t* ops, int n_ops, int compact, void* sync_ws, void* stream);

