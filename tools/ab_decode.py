#!/usr/bin/env python3
"""Same-process A/B of whole-model greedy decode (one GPU, hipGraph token loop): two DecodeStage variants built side by side
and timed alternately, so that box-to-box differences (+-2 %) do not hide a 1 % step.
    python tools/ab_decode.py [--tokens 64] [--rounds 4] [--compact]
Variants: DecodeStage(staging=...) "consumer" (round-4 launches), "swiglu", "noarena" (swiglu with one allocation per weight) (mxq_amd/llama_decode.py)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import llama_shapes as LS  # noqa: E402
from mxq_amd.llama_decode import DecodeStage  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--compact", action="store_true")
    ap.add_argument("--variants", default="consumer,swiglu")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    stages, bufs = {}, {}
    names = args.variants.split(",")
    for name in names:
        st = (DecodeStage(range(LS.N_LAYERS), dev, max_ctx=args.ctx, compact=args.compact, arena=False) if name == "noarena"
              else DecodeStage(range(LS.N_LAYERS), dev, max_ctx=args.ctx, compact=args.compact, staging=name))
        bufs[name] = torch.zeros(1, dtype=torch.int64, device=dev)
        st.capture_token_loop(bufs[name])
        stages[name] = st
    res = {k: [] for k in stages}
    toks = {}
    for r in range(args.rounds + 1):
        for name in (names if r % 2 == 0 else names[::-1]):
            st = stages[name]
            st.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = st.decode_tokens(bufs[name], 1, args.tokens)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            toks[name] = out
            if r:                                   # round 0 = warm-up
                res[name].append(args.tokens / dt)
    out = {k: {"tokens_per_s": [round(v, 1) for v in vs], "median": round(sorted(vs)[len(vs) // 2], 1)} for k, vs in res.items()}
    out["tokens_identical"] = all(toks[n] == toks[names[0]] for n in names)
    out["gain_pct_vs_first"] = {n: round((out[n]["median"] / out[names[0]]["median"] - 1) * 100, 2) for n in names[1:]}
    out["mode"] = "compact" if args.compact else "exact"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
