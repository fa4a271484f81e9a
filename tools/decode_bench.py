#!/usr/bin/env python3
"""BASELINE config 3: Llama-2-7B W2/4A16 greedy decode, batch 1, layers pipeline-sharded.

    python tools/decode_bench.py [--tokens 64] [--ctx 512] [--layers 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 tools/decode_bench.py

Prints one JSON line on rank 0: tokens/s, ms/token, packed weight bytes streamed per token and
the implied HBM GB/s.  A batch-1 pipeline is sequential, so G GPUs are expected to be ~flat.

Rehearsal on a box with fewer GPUs than ranks (as bench.py): MXQ_BENCH_BACKEND=gloo makes the ranks share
devices (rank r uses device r % device_count) and the hidden state / token hop through host memory; the
numbers of such a run mean nothing, the control flow and the tokens do.  --verify: rank 0 also decodes the
same tokens in ONE process (all layers, token-loop graph) and the run fails unless the pipeline's ids are
identical -- the multi-rank path (LayerPipeline.decode + DecodeStage.step_graph + embed_token / head) checked
against the single-process path end to end."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import llama_shapes as LS  # noqa: E402
from mxq_amd.llama_decode import decode_pipeline_figure  # noqa: E402
from mxq_amd.pipeline import LayerPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=512)
    ap.add_argument("--layers", type=int, default=LS.N_LAYERS)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--verify", action="store_true", help="world > 1: rank 0 re-decodes in one process and requires "
                                                          "identical token ids")
    ap.add_argument("--compact", action="store_true", help="compact metadata mode (fp16 zero-points, 3.75 bit/weight); "
                                                           "default: exact metadata (4.5 bit/weight)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MXQ_BENCH_BACKEND", "nccl")
    local = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        from mxq_amd.pipeline import init_group
        init_group(backend, rank, world, dev)        # 120-s group timeout: a stuck hop fails instead of blocking 10 minutes
    pipe = LayerPipeline(rank, world)
    fig = decode_pipeline_figure(pipe, dev, tokens=args.tokens, ctx=args.ctx, layers=args.layers, compact=args.compact,
                                 verify=args.verify, graph=not args.no_graph, dist=dist if world > 1 else None,
                                 backend=backend)
    if rank == 0:
        if args.verify and world > 1 and not fig["tokens_equal_single_process"]:
            raise SystemExit(f"pipeline decode at world {world} differs from the single-process decode at token "
                             f"{fig['first_mismatch']}")
        fig.pop("token_ids")
        fig["tokens_equal_single_process"] = args.tokens if fig["tokens_equal_single_process"] else None
        print(json.dumps(fig), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    from mxq_amd.pipeline import run_guarded
    run_guarded(main)
