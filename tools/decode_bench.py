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
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import llama_shapes as LS  # noqa: E402
from mxq_amd.llama_decode import DecodeStage  # noqa: E402
from mxq_amd.pipeline import LayerPipeline, layer_range  # noqa: E402


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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    pipe = LayerPipeline(rank, world)
    stage = DecodeStage(layer_range(rank, world, args.layers), dev, max_ctx=args.ctx, first=pipe.is_first,
                        last=pipe.is_last, compact=args.compact)
    if not args.no_graph:
        stage.capture()
    hbuf = torch.zeros(1, LS.HIDDEN, device=dev, dtype=torch.float16)
    tbuf = torch.zeros(1, dtype=torch.int64, device=dev)

    def stage_fn(h, step):
        if args.no_graph:
            out = stage.step(h)
            stage.advance()
            return out
        return stage.step_graph(h)

    single = world == 1 and not args.no_graph
    if single:
        stage.capture_token_loop(tbuf)     # token -> token in one graph (embedding, layers, head, argmax)

    def run(n):
        stage.reset()
        if single:
            return stage.decode_tokens(tbuf, 1, n)
        return pipe.decode(1, n, stage.embed_token if pipe.is_first else None, stage_fn,
                           stage.head if pipe.is_last else None, hbuf, tbuf)

    run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks = run(args.tokens)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bytes_tok = torch.tensor([stage.packed_bytes()], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(bytes_tok)
    verified = None
    if args.verify and world > 1 and rank == 0:
        # the same decode in ONE process: every layer, embedding and head on this device, token-loop graph
        ref = DecodeStage(range(args.layers), dev, max_ctx=args.ctx, first=True, last=True, compact=args.compact)
        rtok = torch.zeros(1, dtype=torch.int64, device=dev)
        ref.capture_token_loop(rtok)
        ref.reset()
        want = ref.decode_tokens(rtok, 1, args.tokens)
        if want != toks:
            bad = next(i for i, (a, b) in enumerate(zip(want, toks)) if a != b)
            raise SystemExit(f"pipeline decode at world {world} differs from the single-process decode at token {bad}: "
                             f"{toks[bad]} != {want[bad]}")
        verified = len(want)
        del ref
    if rank == 0:
        print(json.dumps({"config": "Llama-2-7B W2/4A16 greedy decode, batch 1", "n_gpus": world,
                          "layers": args.layers, "tokens": args.tokens, "ctx": args.ctx,
                          "tokens_per_s": round(args.tokens / dt, 1), "ms_per_token": round(dt / args.tokens * 1e3, 3),
                          "packed_weight_GB_per_token": round(bytes_tok.item() / 1e9, 3),
                          "weight_stream_GBps": round(bytes_tok.item() / (dt / args.tokens) / 1e9, 1),
                          "metadata_mode": "compact (fp16 zero-points)" if args.compact else "exact (fp32 zero-points)",
                          "hipgraph": not args.no_graph, "first_tokens": toks[:8], "backend": backend if world > 1 else None,
                          "tokens_equal_single_process": verified}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
