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


@pytest.mark.gpu
@pytest.mark.parametrize("compact", [False, True])
def test_decode_pipeline_world2_tokens_equal_single_process(compact):
    env = dict(os.environ, MXQ_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "tools", "decode_bench.py"), "--tokens", "12",
           "--layers", "4", "--ctx", "64", "--verify"] + (["--compact"] if compact else [])
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
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
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["distinct_devices"] == 1                                   # a rehearsal: one GPU under both ranks
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and len({x["pid"] for x in d["ranks"]}) == 2
    assert all(x["device_id"] for x in d["ranks"])
    fig = d["decode_pipeline"]
    assert fig["n_gpus"] == 2 and fig["tokens"] == 32 and fig["tokens_equal_single_process"] is True
    assert fig["tokens_per_s"] > 0 and d["value"] > 0
