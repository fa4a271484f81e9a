#!/usr/bin/env python3
"""One consolidated round report over the five BASELINE.json configs (run on the GPU box):

    python tools/roofline_report.py > gpurun_out/report.json

config 1  CPU dequant + F.linear (the bench's cpu_baseline leg)        -> TFLOP/s on the host cores
config 2  all 224 Linears at 2048 tokens (bench.py)                     -> TFLOP/s, fraction of the 2.5 PF MFMA peak
config 3  greedy decode, batch 1 (tools/decode_bench.py, one GPU)       -> tokens/s, packed GB/s vs 8 TB/s
config 4  MXAsymQuantizer fwd + STE bwd on one decoder block, bf16      -> us, GB/s vs 8 TB/s
config 5  W2 / W4 / mixed sweep at 32768 tokens (tools/sweep_config5.py)-> TFLOP/s per arm
Each leg is its own subprocess (fresh clocks / caches); the numbers are what the tools print."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args):
    out = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        raise SystemExit(f"{args}: rc={out.returncode}\n{out.stderr[-2000:]}")
    return out.stdout


def last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError("no JSON line in output")


def main():
    rep = {"peaks": {"fp16_mfma_TFLOPs": 2500.0, "hbm_GBps_spec": 8000.0, "hbm_GBps_achievable": 6300.0}}
    b = last_json(run(["bench.py", "--gpus", "1", "--steps", "10", "--warmup", "3"]))
    rep["config1_cpu_dequant_linear"] = b["cpu_baseline"]
    rep["config2_prefill_all_linears"] = {k: b[k] for k in ("value", "unit", "ms_per_step", "tokens_per_s", "roofline")}
    rep["config3_decode_1gpu"] = {}
    for mode, extra in (("exact_metadata", []), ("compact_metadata", ["--compact"])):     # state the mode
        d = last_json(run(["tools/decode_bench.py", "--tokens", "64"] + extra))
        rep["config3_decode_1gpu"][mode] = {"tokens_per_s": d["tokens_per_s"], "ms_per_token": d["ms_per_token"],
                                            "packed_weight_GBps": d["weight_stream_GBps"],
                                            "hbm_frac_of_8TBps": round(d["weight_stream_GBps"] / 8000.0, 3),
                                            "floor_ms_per_token_at_6.3TBps": round(d["packed_weight_GB_per_token"] / 6.3, 3)}
    kb = run(["tools/kernels_bench.py"])
    blk = [l for l in kb.splitlines() if "one decoder block" in l]
    rep["config4_fakequant_block_bf16"] = {"line": blk[0].strip() if blk else None}
    s = last_json(run(["tools/sweep_config5.py"]))
    rep["config5_sweep_32768_tokens"] = {a: {"TFLOPs": v["TFLOPs"], "mfma_frac": round(v["TFLOPs"] / 2500.0, 3),
                                            "tokens_per_s": v["tokens_per_s"]} for a, v in s["arms"].items()}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
