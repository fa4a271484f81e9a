#!/usr/bin/env python3
"""BASELINE config 5: Llama-2-7B W2A16 vs W4A16 vs mixed-2/4 sweep at seq=4096 batch=8
(M = 32768 tokens), 1 x MI355X.  For each arm and each of the three Linear shapes: kernel time
(HIP events), TFLOP/s, fraction of the 2.5 PF fp16 MFMA peak, packed bits/weight; the per-layer
and 32-layer totals; and PyTorch's fp16 GEMM (hipBLASLt) on the dequantised weight as reference.  Every
quantised arm is checked at the full size (<= 1e-3 max-norm and Frobenius against the fp32 product on the
kernel-dequantised weight, 4096 sampled token rows) before its time is reported.
    python tools/sweep_config5.py [--m 32768] [--iters 3]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import packing  # noqa: E402

SHAPES = [("q/k/v/o", 4096, 4096, 4), ("gate/up", 11008, 4096, 2), ("down", 4096, 11008, 1)]
PEAK = 2500.0


def timed(fn, iters):
    for _ in range(4):      # warm-up: the first launches of a process run at a different clock / cold caches
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=32768)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    M = args.m
    report = {"config": f"W2A16 / W4A16 / mixed-2/4 sweep, M = {M} tokens (batch 8 x seq 4096)", "arms": {}}
    for arm in ("mixed", "w2g16", "w4row", "fp16-hipblaslt"):
        layer_t = layer_f = 0.0
        rows = []
        for name, N, K, mult in SHAPES:
            g = torch.Generator(device=dev).manual_seed(N * 7 + K)
            W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
            x = torch.randn(M, K, generator=g, device=dev).half()
            out = torch.empty(M, N, device=dev, dtype=torch.float16)
            if arm == "fp16-hipblaslt":
                t = timed(lambda: torch.matmul(x, W.t(), out=out), args.iters)
                bpw = 16.0
            else:
                p = packing.quantize_pack(W) if arm == "mixed" else packing.quantize_pack_uniform(W, arm)
                t = timed(lambda: packing.linear_layout(x, p, out=out), args.iters)
                bpw = p.bits_per_weight()
                # parity at the full size: fp32 product on the kernel-dequantised weight (bit-exact against the
                # oracle in tests/), 4096 sampled token rows; both error norms of SURVEY.md 8c
                wd = packing.dequant(p) if arm == "mixed" else packing.expand_uniform(p, codes=False)[0]
                rows_ = torch.randint(0, M, (4096,), generator=g, device=dev)
                ref = x[rows_].float() @ wd.float().t()
                got = out[rows_].float()
                err_max = ((got - ref).abs().max() / ref.abs().max()).item()
                err_fro = ((got - ref).norm() / ref.norm()).item()
                assert err_max <= 1e-3 and err_fro <= 1e-3, (arm, name, err_max, err_fro)
                rows_err = {"max_rel": err_max, "fro_rel": err_fro}
                del wd, ref, got
            fl = 2.0 * M * N * K
            rows.append({"linear": name, "N": N, "K": K, "ms": round(t * 1e3, 3), "TFLOPs": round(fl / t / 1e12, 1),
                         "mfma_frac": round(fl / t / 1e12 / PEAK, 3), "bits_per_weight": round(bpw, 3)})
            if arm != "fp16-hipblaslt":
                rows[-1].update({k: float(f"{v:.3g}") for k, v in rows_err.items()})
            layer_t += mult * t
            layer_f += mult * fl
            del W, x, out
        report["arms"][arm] = {"per_linear": rows, "layer_ms": round(layer_t * 1e3, 3),
                               "model_32_layers_ms": round(32 * layer_t * 1e3, 2),
                               "TFLOPs": round(layer_f / layer_t / 1e12, 1),
                               "tokens_per_s": round(M / (32 * layer_t), 1)}
        print(arm, json.dumps(report["arms"][arm]), flush=True)
    print(json.dumps(report))


if __name__ == "__main__":
    main()
