#!/usr/bin/env python3
"""CPU rehearsal of bench.py's multi-rank control flow at any world size (gloo, no GPU): the SAME harness objects --
mxq_amd.pipeline.Watchdog / init_group / run_guarded / LayerPipeline.run_microbatches(stats=...) / gather_reports /
rank_census / LayerPipeline.decode / hop_round_trip_us -- around a toy stage (a few fp32 matmuls) instead of the HIP
kernels.  It measures nothing about the product; it proves that N ranks rendezvous, stream micro-batches, account
for their waits, decode, report and tear down -- and that a rank that dies or hangs ends the JOB non-zero, quickly, with
the rank named (tests/test_pipeline_gloo.py drives it at world 8, with and without an injected fault).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        tools/pipeline_rehearsal.py --steps 2
    MXQ_BENCH_FAULT=3:exit | 3:hang   rank 3 dies (exit 17) / stops responding right after the rendezvous

This is NOT a fallback of anything: bench.py itself refuses to run without a GPU."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mxq_amd import pipeline  # noqa: E402

N_LAYERS, HID, TOK, VOCAB = 32, 64, 16, 97


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.set_num_threads(1)
    dog = pipeline.Watchdog(rank, world, hard_deadline_s=float(os.environ.get("MXQ_BENCH_DEADLINE_S", 240)))
    dog.phase("rendezvous", 60)
    import torch.distributed as dist
    pipeline.init_group("gloo", rank, world)
    fault = os.environ.get("MXQ_BENCH_FAULT", "")
    if fault and int(fault.split(":")[0]) == rank:
        if fault.endswith(":exit"):
            pipeline.rank_log("MXQ_BENCH_FAULT: this rank dies now (exit 17)", rank, world)
            os._exit(17)
        pipeline.rank_log("MXQ_BENCH_FAULT: this rank hangs now", rank, world)
        dog.phase("injected hang", float(os.environ.get("MXQ_BENCH_FAULT_DEADLINE_S", 10)))
        time.sleep(3600)
    dog.phase("build", 30)
    g = torch.Generator().manual_seed(0)
    Ws = [torch.randn(HID, HID, generator=g) / HID ** 0.5 for _ in range(N_LAYERS)]
    emb, head = torch.randn(VOCAB, HID, generator=g), torch.randn(VOCAB, HID, generator=g)
    mine = pipeline.layer_range(rank, world, N_LAYERS)
    ybuf = torch.empty(TOK, HID)

    def stage_fn(h, step=None):
        for i in mine:
            h = torch.tanh(h @ Ws[i].t()) + h
        return ybuf[: h.shape[0]].copy_(h) if h.shape[0] == TOK else h

    pipe = pipeline.LayerPipeline(rank, world)
    xs = [torch.full((TOK, HID), float(b + 1)) / 7 for b in range(world)]
    dog.phase("warmup", 60)
    for _ in range(args.warmup):
        pipe.run_microbatches(stage_fn, xs, torch.empty(TOK, HID), collect=False)
    dist.barrier()
    dog.phase("timed steps", 60)
    stats = pipeline.PipelineStats(None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.run_microbatches(stage_fn, xs, torch.empty(TOK, HID), collect=False, stats=stats)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    dog.phase("reports", 60)
    rep = {"rank": rank, "layers": [mine[0], mine[-1]], "host_ms_per_step": round(elapsed / args.steps * 1e3, 3)}
    rep.update(stats.summary(args.steps))
    per_rank = pipeline.gather_reports(rep)
    census = pipeline.rank_census()
    dog.phase("decode", 60)
    gen = pipe.decode(3, 6, lambda t: emb.index_select(0, t), stage_fn, lambda h: (h @ head.t()).argmax(-1),
                      torch.empty(1, HID), torch.zeros(1, dtype=torch.int64))
    rtt = pipe.hop_round_trip_us(torch.zeros(1, HID), iters=5)
    hops = pipeline.gather_reports({"rank": rank, "hop_round_trip_us_to_next_rank": rtt})
    if rank == 0:
        h = emb[3:4]
        tok, want = 3, []
        for _ in range(6):
            h = emb[tok:tok + 1]
            for i in range(N_LAYERS):
                h = torch.tanh(h @ Ws[i].t()) + h
            tok = int((h @ head.t()).argmax())
            want.append(tok)
        print(json.dumps({"rehearsal": True, "n_ranks": world, "steps": args.steps, "ranks_seen": census["ranks_seen"],
                          "per_rank": per_rank, "hops": hops, "decode_tokens_equal_single_process": gen == want,
                          "phases_s": dog.history}), flush=True)
    dog.phase("teardown", 30)
    dist.barrier()
    dist.destroy_process_group()
    dog.done()


if __name__ == "__main__":
    pipeline.run_guarded(main)
