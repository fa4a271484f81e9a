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
