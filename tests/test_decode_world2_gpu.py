"""BASELINE configs[2] end to end on ONE GPU box: the layer pipeline's greedy decode (LayerPipeline.decode +
DecodeStage.step_graph + embed_token / head) at world 2 over gloo with both ranks sharing cuda:0, against the
single-process decode of the same model -- token ids must be identical (tools/decode_bench.py --verify).  The launcher
(torch.distributed.run) never touches the GPU; two worker processes do (well inside the box's process guard)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> str:
    """A port nobody listens on right now (bind to 0, read it back): a leftover or parallel run on the box must not fail
    the rendezvous with "address in use"."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _torchrun(nproc, script_args, env, timeout):
    """`python -m torch.distributed.run` of one of this repo's multi-rank programs on 127.0.0.1 with a fresh port.  A job in which
    NOTHING of ours ran -- no rank-tagged line on stderr, no JSON on stdout: the launcher's own rendezvous failed before any
    worker reached its first line, e.g. the port picked a moment ago was taken -- is started once more on another port
    (nothing has touched the GPU in such a job); any other outcome is returned as it is."""
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", _free_port()] + script_args
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        ours = "[rank " in r.stderr or any(l.startswith("{") for l in r.stdout.splitlines())
        if r.returncode == 0 or ours:
            break
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("compact", [False, True])
def test_decode_pipeline_world2_tokens_equal_single_process(compact):
    env = dict(os.environ, MXQ_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _torchrun(2, [os.path.join(ROOT, "tools", "decode_bench.py"), "--tokens", "12", "--layers", "4", "--ctx", "64", "--verify"]
                  + (["--compact"] if compact else []), env, 600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["tokens_equal_single_process"] == 12 and d["backend"] == "gloo"


@pytest.mark.gpu
def test_bench_world2_proves_its_ranks_and_carries_the_decode_figure():
    """bench.py --gpus 2 (gloo rehearsal: both ranks share cuda:0): the JSON line must show WHO took part -- ranks_seen
    from an all-reduce of ones, every rank's pid / device identity -- and carry the bounded configs[2] side figure
    (32 greedy-decode tokens through the same layer pipeline, rank 0 re-decoding them in one process).  On the driver's
    8-GPU SCALE run the same fields read ranks_seen = distinct_devices = N over RCCL."""
    env = dict(os.environ, MXQ_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env, 900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["distinct_devices"] == 1                                   # a rehearsal: one GPU under both ranks
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and len({x["pid"] for x in d["ranks"]}) == 2
    assert all(x["device_id"] for x in d["ranks"])
    fig = d["decode_pipeline"]
    assert fig["n_gpus"] == 2 and fig["tokens"] == 32 and fig["tokens_equal_single_process"] is True
    assert fig["tokens_per_s"] > 0 and d["value"] > 0
    _check_per_rank(d, 2)


def _check_per_rank(d, world):
    """Every rank's own account of the timed region rides on rank 0's line: its GEMMs' device time, the
    time its stream waited for the hop in / for a send slot, bytes hopped, stream-K status, shader clock; and per stage of the
    decode pipeline the time per token and the hidden row's round trip to the next rank."""
    assert d["config"]["entry"].startswith("packing.linear(path='auto')")
    assert [x["rank"] for x in d["per_rank"]] == list(range(world))
    hop = 2048 * 4096 * 2
    for x in d["per_rank"]:
        first, last = x["rank"] == 0, x["rank"] == world - 1
        assert x["workspace_status"] == "ok" and x["microbatches_per_step"] == world
        assert x["bytes_sent_per_step"] == (0 if last else world * hop) and x["bytes_received_per_step"] == (0 if first else world * hop)
        assert x["stream_compute_ms_per_step"] > 0 and x["stage_TFLOPs"] > 0 and x["launches_per_step"] == 224
        assert first or x["stream_recv_wait_ms_per_step"] >= 0
        assert x["sclk_MHz"] is None or 500 < x["sclk_MHz"]["median"] < 3000
    assert [p[0] for p in d["phases_s"]][:5] == ["start", "rendezvous", "build weights", "warmup", "timed steps"]
    pr = d["decode_pipeline"]["per_rank"]
    assert [x["rank"] for x in pr] == list(range(world)) and all(x["stage_us_per_token"] > 0 for x in pr)
    assert all((x["hop_round_trip_us_to_next_rank"] is not None) == (x["rank"] < world - 1) for x in pr)
    assert d["decode_pipeline"]["sum_of_stages_ms"] > 0


@pytest.mark.gpu
def test_bench_world4_rehearsal_on_one_gpu():
    """bench.py --gpus 4 under the gloo rehearsal backend (4 ranks share cuda:0; with this test process that is 5 of the 6
    processes a box lets on its card -- the driver's N = 8 run cannot be rehearsed on one GPU; the world-8 CONTROL FLOW is
    rehearsed on CPU by tests/test_pipeline_gloo.py through the same harness objects): 8 layers per rank, 4 sequences in
    flight, every rank's report on the line."""
    env = dict(os.environ, MXQ_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _torchrun(4, [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1"], env, 600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 4 and d["backend"] == "gloo" and d["distinct_devices"] == 1
    assert d["decode_pipeline"]["tokens_equal_single_process"] is True
    _check_per_rank(d, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("fault,code", [("1:exit", 17), ("1:hang", 86)])
def test_bench_world2_a_dead_or_stuck_rank_fails_the_job_fast_and_is_named(fault, code):
    import time
    env = dict(os.environ, MXQ_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               MXQ_BENCH_FAULT=fault, MXQ_GROUP_TIMEOUT_S="30", MXQ_BENCH_FAULT_DEADLINE_S="10")
    t0 = time.time()
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env, 400)
    took = time.time() - t0
    assert r.returncode != 0 and took < 180, (r.returncode, took)
    assert "[rank 1/2" in r.stderr and f"exitcode: {code}) local_rank: 1" in r.stderr, r.stderr[-6000:]
    if code == 86:
        assert "WATCHDOG: phase 'injected hang' exceeded its deadline" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
