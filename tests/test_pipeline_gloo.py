"""The layer-pipeline schedule (mxq_amd/pipeline.py) on CPU with gloo, world_size 2, 3, 4 and 8:
micro-batch streaming and greedy decode must reproduce the single-process result exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mxq_amd.pipeline import LayerPipeline, layer_range, rank_census

N_LAYERS, HID, VOCAB = 8, 32, 50


def _weights():
    g = torch.Generator().manual_seed(0)
    Ws = [torch.randn(HID, HID, generator=g) / HID ** 0.5 for _ in range(N_LAYERS)]
    emb = torch.randn(VOCAB, HID, generator=g)
    head = torch.randn(VOCAB, HID, generator=g)
    return Ws, emb, head


def _stage(Ws, layers):
    def fn(h, step=None):
        for i in layers:
            h = torch.tanh(h @ Ws[i].t()) + h
        return h
    return fn


def _reference(n_tokens, first):
    Ws, emb, head = _weights()
    fn = _stage(Ws, range(N_LAYERS))
    xs = [torch.full((4, HID), float(b + 1)) / 7 for b in range(5)]
    outs = [fn(x) for x in xs]
    tok, gen = first, []
    for _ in range(n_tokens):
        h = fn(emb[tok:tok + 1])
        tok = int((h @ head.t()).argmax())
        gen.append(tok)
    return outs, gen


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Ws, emb, head = _weights()
        pipe = LayerPipeline()
        assert (pipe.rank, pipe.world) == (rank, world)
        fn = _stage(Ws, layer_range(rank, world, N_LAYERS))
        xs = [torch.full((4, HID), float(b + 1)) / 7 for b in range(5)]
        outs = pipe.run_microbatches(fn, xs, torch.empty(4, HID))
        # bench.py's N > 1 schedule: a reused output buffer hops on (the send ring must decouple it), nothing is collected
        ybuf = torch.empty(4, HID)
        assert pipe.run_microbatches(lambda h: ybuf.copy_(h * 2), xs * 3, torch.empty(4, HID), collect=False) == []
        gen = pipe.decode(3, 6, lambda t: emb.index_select(0, t), fn, lambda h: (h @ head.t()).argmax(-1),
                          torch.empty(1, HID), torch.zeros(1, dtype=torch.int64))
        # bench.py's proof of participation: every rank counted once, every rank's identity on every rank
        cen = rank_census()
        assert cen["ranks_seen"] == world and [r["rank"] for r in cen["ranks"]] == list(range(world))
        assert len({r["pid"] for r in cen["ranks"]}) == world and cen["distinct_devices"] == 0    # CPU run: no device
        # by value (numpy), not as shared-memory tensors: a shared tensor must be rebuilt while its producer still runs
        q.put((rank, [o.numpy().copy() for o in outs], gen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_pipeline_matches_single_process(world):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, outs, gen = q.get(timeout=300)
        res[r] = (outs, gen)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_outs, ref_gen = _reference(6, 3)
    outs, gen = res[world - 1]
    assert len(outs) == 5 and all(torch.equal(torch.from_numpy(a), b) for a, b in zip(outs, ref_outs))
    assert res[0][1] == ref_gen and res[world - 1][1] == ref_gen      # first and last stage track the tokens
    assert all(res[r][1] == [] for r in range(1, world - 1))          # the stages in between never see them
    assert all(len(res[r][0]) == 0 for r in range(world - 1))


def _worker_subgroup(rank, world, port, q):
    """The pipeline lives in a SUB-GROUP (global ranks 1, 2 of a 3-rank job): group ranks differ from global ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grp = dist.new_group([1, 2])
        out = None
        if rank in (1, 2):
            Ws, emb, head = _weights()
            pipe = LayerPipeline(group=grp)
            assert (pipe.rank, pipe.world) == (rank - 1, 2)
            fn = _stage(Ws, layer_range(pipe.rank, 2, N_LAYERS))
            xs = [torch.full((4, HID), float(b + 1)) / 7 for b in range(5)]
            outs = pipe.run_microbatches(fn, xs, torch.empty(4, HID))
            gen = pipe.decode(3, 6, lambda t: emb.index_select(0, t), fn, lambda h: (h @ head.t()).argmax(-1),
                              torch.empty(1, HID), torch.zeros(1, dtype=torch.int64))
            out = ([o.numpy().copy() for o in outs], gen)
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipeline_in_a_subgroup_addresses_its_own_members():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_subgroup, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_outs, ref_gen = _reference(6, 3)
    assert res[0] is None
    outs, gen = res[2]
    assert len(outs) == 5 and all(torch.equal(torch.from_numpy(a), b) for a, b in zip(outs, ref_outs))
    assert gen == ref_gen and res[1][1] == ref_gen and res[1][0] == []


def test_layer_range_partition():
    for world in (1, 2, 3, 4, 8):
        got = [i for r in range(world) for i in layer_range(r, world, 32)]
        assert got == list(range(32))
        sizes = [len(layer_range(r, world, 32)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


# -- bench.py's multi-rank harness at world 8 (tools/pipeline_rehearsal.py: the same Watchdog / init_group / run_guarded /
#    PipelineStats / gather_reports / hop_round_trip_us objects bench.py uses, around a toy CPU stage) -------------------
import json
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _rehearse(world, extra_env=None, timeout=240):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", **(extra_env or {}))
    t0 = time.time()
    for attempt in range(2):      # (a job in which no rank ever logged a line -- the launcher's own rendezvous failed, e.g. the port
        #                            picked a moment ago was taken -- is started once more on another port)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "tools", "pipeline_rehearsal.py"), "--steps", "2"]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        if r.returncode == 0 or "[rank " in r.stderr:
            break
    return r, time.time() - t0


def test_harness_world8_every_rank_accounts_for_its_waits():
    r, _ = _rehearse(8)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_ranks"] == 8 and d["ranks_seen"] == 8 and d["decode_tokens_equal_single_process"] is True
    assert [x["rank"] for x in d["per_rank"]] == list(range(8))
    for x in d["per_rank"]:
        first, last = x["rank"] == 0, x["rank"] == 7
        assert x["microbatches_per_step"] == 8
        assert x["hops_sent_per_step"] == (0 if last else 8) and x["hops_received_per_step"] == (0 if first else 8)
        assert x["bytes_sent_per_step"] == (0 if last else 8 * 16 * 64 * 4)
        assert x["host_compute_ms_per_step"] > 0 and (first or x["host_recv_wait_ms_per_step"] > 0)
    assert [h["hop_round_trip_us_to_next_rank"] is not None for h in d["hops"]] == [True] * 7 + [False]
    assert [p[0] for p in d["phases_s"]][:4] == ["start", "rendezvous", "build", "warmup"]


@pytest.mark.parametrize("fault,code", [("3:exit", 17), ("5:hang", 86)])
def test_harness_world8_a_dead_or_stuck_rank_fails_the_job_fast_and_is_named(fault, code):
    """a deliberately killed rank makes the job exit != 0 well within 180 s with the rank named; a rank
    that stops responding is ended by its own watchdog (stack dump, exit 86), never by the driver's silent limit."""
    r, took = _rehearse(8, {"MXQ_BENCH_FAULT": fault, "MXQ_GROUP_TIMEOUT_S": "20", "MXQ_BENCH_FAULT_DEADLINE_S": "5"})
    who = fault.split(":")[0]
    assert r.returncode != 0 and took < 120, (r.returncode, took)
    assert f"[rank {who}/8" in r.stderr and "MXQ_BENCH_FAULT" in r.stderr
    assert f"exitcode: {code}) local_rank: {who}" in r.stderr                 # the launcher names the rank that failed first
    if code == 86:
        assert "WATCHDOG: phase 'injected hang' exceeded its deadline" in r.stderr and "File " in r.stderr   # with its stacks
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]        # no JSON line from a failed job


def test_run_guarded_turns_an_exception_into_a_tagged_nonzero_exit():
    code = ("import sys; sys.path.insert(0, %r)\nfrom mxq_amd.pipeline import run_guarded\n"
            "def main():\n    raise ValueError('boom')\nrun_guarded(main)\nprint('not reached')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, RANK="2", WORLD_SIZE="4"))
    assert r.returncode == 1 and "[rank 2/4" in r.stderr and "ValueError: boom" in r.stderr and "not reached" not in r.stdout
