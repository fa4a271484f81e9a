#!/usr/bin/env python3
"""Same-process A/B of the headline step with and without the stream-K split of gate/up's tail at 2048 tokens.

At 2048 tokens only gate/up (688 tiles on 256 CUs: tail 176) has a tail to split; 4096^2 and down are exactly one round of
256 tiles.  path "gemm" = the product dispatch (tail split since dispatch threshold 20, commit 108ba09), path "whole" = the
same kernel without a workspace = whole tiles only = what threshold 24 ran.  Alternating rounds of `--steps` steps each."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mxq_amd import llama_shapes as LS, packing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    layers = bench.build_layers(range(LS.N_LAYERS), dev)
    gx = torch.Generator(device=dev).manual_seed(7)
    x_h = torch.randn(bench.SEQ, LS.HIDDEN, generator=gx, device=dev).half()
    x_i = torch.randn(bench.SEQ, LS.INTERMEDIATE, generator=gx, device=dev).half()
    y = {N: torch.empty(bench.SEQ, N, device=dev, dtype=torch.float16) for N in (LS.HIDDEN, LS.INTERMEDIATE)}

    def step(path):
        for lin in layers:
            for _n, p in lin:
                packing.linear(x_i if p.K == LS.INTERMEDIATE else x_h, p, out=y[p.N], path=path)
    res = {"gemm": [], "whole": []}
    for path in res:
        for _ in range(3):
            step(path)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for path in (("gemm", "whole") if r % 2 == 0 else ("whole", "gemm")):
            res[path].append(bench._events_ms(lambda: step(path), args.steps))
    out = {}
    for path, ts in res.items():
        ts = sorted(ts)
        out[path] = {"ms_per_step_median": round(ts[len(ts) // 2], 4), "min": round(ts[0], 4), "max": round(ts[-1], 4),
                     "TFLOPs": round(LS.linear_flops(bench.SEQ) / ts[len(ts) // 2] / 1e9, 1)}
    out["split_gain_pct"] = round((out["whole"]["ms_per_step_median"] / out["gemm"]["ms_per_step_median"] - 1) * 100, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
