"""Whole-layer pipeline sharding over the GPUs of one node.

The reference's only multi-GPU inference mode is accelerate's ``device_map="auto"``
(mxq_quant/main.py:23, lib/prune.py:371-378): decoder layers are placed on successive GPUs
and the hidden state hops GPU -> GPU with ``.to(dev)`` inside ONE process.  The native
counterpart: one process per GPU, rank ``r`` of ``G`` owns layers ``[L*r/G, L*(r+1)/G)``, and the
only exchange is the ``[tokens, hidden]`` fp16 activation moving to the next stage with
``torch.distributed`` point-to-point ``isend``/``irecv`` (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests) plus, for greedy decode, the 8-byte next-token id going from the last
stage back to the first.  No all-reduce / all-gather / broadcast is needed, so the per-link ring bound
of xGMI never enters.

Schedule of ``run_microbatches`` (stage r, micro-batch b):

    wait  irecv(b)
    h = stage_fn(recv slot b % 2)
    wait  isend(b-2) (frees send ring slot b % 2); copy h into it
    ONE group { isend(b), irecv(b+1) into recv ring slot (b+1) % 2 }          <- both fly under b+1's compute

RCCL runs point-to-point transfers on its own stream, ordered against the compute stream by events at
post / wait time, so "flies under" is real overlap on the GPU; the blocking ``send`` / ``recv`` of round 1
made the compute stream wait for every hop.  The send of b and the receive of b+1 are posted as one group
(``dist.batch_isend_irecv``): issued one after the other on a communicator's single stream, the later one
would wait for the earlier one's peer (round 5; a host-memory backend -- gloo moving device tensors --
keeps the receive-first order, it has no such stream).  The copy into the send ring (16.8 MB at 2048 tokens, ~4 us)
decouples the transfer from whatever buffer the stage reuses for its next output.

Nothing here touches the HIP library: the stage computation is a callable, so the schedule is
unit-tested on CPU with gloo (tests/test_pipeline_gloo.py) and used unchanged with RCCL.
"""
from __future__ import annotations

import contextlib
import faulthandler
import os
import sys
import threading
import time
import traceback
from datetime import timedelta
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

GROUP_TIMEOUT_S = 120.0      # process-group timeout of a bench / rehearsal run (torch's default is 10 min: longer than the
                             # driver's own limit, so a stuck rendezvous or hop used to end as "killed at limit", stderr empty)


class _StagedRecv:
    """irecv of a device tensor over a backend that only moves host memory (gloo): the transfer lands in a host
    buffer and ``wait`` copies it onto the device (stream-ordered), so the caller sees the same contract as RCCL's."""

    def __init__(self, work, host, dst):
        self.work, self.host, self.dst = work, host, dst

    def wait(self):
        self.work.wait()
        self.dst.copy_(self.host)


# -- failure containment for multi-rank runs (bench.py --gpus N, tools/decode_bench.py, tools/pipeline_rehearsal.py) -------
def rank_log(msg: str, rank: Optional[int] = None, world: Optional[int] = None) -> None:
    """One rank-tagged line on stderr (stdout stays the ONE JSON line of rank 0)."""
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    print(f"[rank {rank}/{world} pid {os.getpid()} t={time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


class Watchdog:
    """Per-phase deadlines for one rank of a multi-rank run.  ``phase(name, seconds)`` arms the deadline of the phase that
    starts now; when it passes, the rank prints a tagged line naming the phase, dumps the Python stack of every thread
    (faulthandler: where it is stuck -- a rendezvous, a hop's wait, a barrier) and leaves with exit code 86, so the launcher
    (torch.distributed.run) reports THAT rank and tears the others down.  ``hard_deadline_s`` bounds the whole run the same
    way (below the driver's own limit, whose kill leaves no trace).  The process is never replaced: it exits."""

    EXIT_CODE = 86

    def __init__(self, rank: int, world: int, hard_deadline_s: Optional[float] = None):
        self.rank, self.world = rank, world
        self._lock = threading.Lock()
        self._name, self._deadline = "start", None
        self._hard = time.monotonic() + hard_deadline_s if hard_deadline_s else None
        self._stop = threading.Event()
        self.history: List[tuple] = []                     # (phase, seconds it took)
        self._t_phase = time.monotonic()
        faulthandler.enable(file=sys.stderr, all_threads=True)          # a fatal signal prints the stacks too
        self._thread = threading.Thread(target=self._run, name="mxq-watchdog", daemon=True)
        self._thread.start()

    def phase(self, name: str, seconds: float) -> None:
        now = time.monotonic()
        with self._lock:
            self.history.append((self._name, round(now - self._t_phase, 3)))
            self._name, self._deadline, self._t_phase = name, now + seconds, now
        if self.rank == 0 or os.environ.get("MXQ_BENCH_VERBOSE"):
            rank_log(f"phase {name} (deadline {seconds:.0f} s)", self.rank, self.world)

    def done(self) -> None:
        self.phase("done", 1e9)
        self._stop.set()

    def _run(self):
        while not self._stop.wait(0.25):
            now = time.monotonic()
            with self._lock:
                name, deadline = self._name, self._deadline
            late = deadline is not None and now > deadline
            if late or (self._hard is not None and now > self._hard):
                why = (f"phase '{name}' exceeded its deadline" if late else f"run exceeded its hard deadline in phase '{name}'")
                rank_log(f"WATCHDOG: {why}; phases so far {self.history}; stacks of all threads follow; exiting {self.EXIT_CODE}",
                         self.rank, self.world)
                try:
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                    sys.stderr.flush()
                finally:
                    os._exit(self.EXIT_CODE)


def init_group(backend: str, rank: int, world: int, dev: Optional[torch.device] = None,
               timeout_s: Optional[float] = None) -> None:
    """``init_process_group`` with a timeout that fits inside a driver run (GROUP_TIMEOUT_S, or MXQ_GROUP_TIMEOUT_S): a
    collective or point-to-point operation whose peer never shows up fails after that long -- gloo raises in the waiting
    rank, RCCL's watchdog thread aborts the process -- instead of blocking for torch's default 10 minutes."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if timeout_s is None:
        timeout_s = float(os.environ.get("MXQ_GROUP_TIMEOUT_S", GROUP_TIMEOUT_S))
    kw = dict(rank=rank, world_size=world, timeout=timedelta(seconds=timeout_s))
    if backend == "nccl":                      # nccl == RCCL on ROCm; device_id binds the communicator to this rank's GPU
        dist.init_process_group("nccl", device_id=dev, **kw)
    else:
        dist.init_process_group(backend, **kw)


def run_guarded(main: Callable[[], Optional[int]]) -> None:
    """Run ``main`` so that ANY failure of this rank ends the process with a non-zero exit code and a rank-tagged traceback
    on stderr -- at once, without the interpreter's orderly shutdown: a rank that raises while its peers sit in a
    collective would otherwise hang in the process group's destructor and the job would end as a silent timeout."""
    try:
        code = main()
    except SystemExit as e:
        code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        if code and not isinstance(e.code, int):
            rank_log(f"FAILED: {e.code}")
    except BaseException:          # noqa: BLE001 -- report, then leave
        rank_log("FAILED with an exception:\n" + traceback.format_exc())
        code = 1
    sys.stdout.flush()
    sys.stderr.flush()
    if code:
        os._exit(int(code))        # peers may be blocked on this rank: leave NOW, the launcher tears them down


class PipelineStats:
    """Per-rank timeline of ``run_microbatches``: for every micro-batch the time this stage spent (a) waiting for the
    incoming hidden state, (b) in its own layers, (c) waiting for a send-ring slot.  On a GPU the three are spans between
    HIP events on the compute stream (a ``wait`` on an RCCL work object does not block the host: it makes the STREAM wait, and
    only an event pair around it sees how long) next to host-clock spans; on CPU tensors only the host clock.  One bench
    line then explains its own scaling curve: compute is flat in N, the waits are what the hops cost."""

    KINDS = ("recv_wait", "compute", "send_wait")

    def __init__(self, device: Optional[torch.device] = None):
        self.cuda = device is not None and device.type == "cuda"
        self._spans: Dict[str, list] = {k: [] for k in self.KINDS}
        self.host_ms: Dict[str, float] = {k: 0.0 for k in self.KINDS}
        self.bytes_sent = self.bytes_received = self.microbatches = self.hops_sent = self.hops_received = 0

    @contextlib.contextmanager
    def span(self, kind: str):
        ev = None
        if self.cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.host_ms[kind] += (time.perf_counter() - t0) * 1e3
            if ev is not None:
                ev[1].record()
                self._spans[kind].append(ev)

    def summary(self, steps: int = 1) -> dict:
        """Per-step sums in ms (synchronises the device once to read the events)."""
        out = {"microbatches_per_step": self.microbatches / max(1, steps),
               "bytes_sent_per_step": self.bytes_sent / max(1, steps), "bytes_received_per_step": self.bytes_received / max(1, steps),
               "hops_sent_per_step": self.hops_sent / max(1, steps), "hops_received_per_step": self.hops_received / max(1, steps)}
        if self.cuda:
            torch.cuda.synchronize()
        for k in self.KINDS:
            out[f"host_{k}_ms_per_step"] = round(self.host_ms[k] / max(1, steps), 4)
            if self.cuda:
                out[f"stream_{k}_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in self._spans[k]) / max(1, steps), 4)
        return out


def gather_reports(report: dict, group=None) -> List[dict]:
    """Every rank's report on every rank (object all-gather: host side, any backend)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return [report]
    out: List[Optional[dict]] = [None] * world
    dist.all_gather_object(out, report, group=group)
    return out


_NULL = contextlib.nullcontext()


def layer_range(rank: int, world: int, n_layers: int) -> range:
    """Layers owned by ``rank``: contiguous, sizes differ by at most one."""
    return range(n_layers * rank // world, n_layers * (rank + 1) // world)


def rank_census(dev=None, group=None) -> dict:
    """Who took part: every rank contributes a one (all-reduce, on the device for RCCL) and its identity -- rank, pid,
    host, device index and the device's UUID / PCI bus id -- gathered onto every rank.  ``ranks_seen == world`` and
    ``distinct_devices == world`` in a bench line prove that N processes ran on N different GPUs (bench.py --gpus N;
    a gloo rehearsal on one GPU shows ``distinct_devices`` 1)."""
    import os
    import socket
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    me = {"rank": rank, "pid": os.getpid(), "host": socket.gethostname(), "device_index": None, "device": None,
          "device_id": None}
    if dev is not None and torch.cuda.is_available():
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        ident = getattr(props, "uuid", None)
        if ident is None or not str(ident).strip("0-"):
            ident = ":".join(str(getattr(props, k, "?")) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        me.update(device_index=idx, device=props.name, device_id=f"{me['host']}/{ident}")
    if world == 1:
        return {"ranks_seen": 1, "distinct_devices": 1 if me["device_id"] else 0, "ranks": [me]}
    staged = dev is not None and dev.type == "cuda" and "nccl" not in str(dist.get_backend(group))
    one = torch.ones(1, dtype=torch.int32, device="cpu" if (dev is None or staged) else dev)
    dist.all_reduce(one, group=group)
    ranks = [None] * world
    dist.all_gather_object(ranks, me, group=group)
    ids = {r["device_id"] for r in ranks if r and r["device_id"]}
    return {"ranks_seen": int(one.item()), "distinct_devices": len(ids), "ranks": ranks}


class LayerPipeline:
    """Point-to-point activation pipeline between consecutive ranks of ``group``.

    ``rank`` / ``world`` are GROUP-relative; ``torch.distributed``'s ``src`` / ``dst`` are global ranks, so peers
    are translated with ``dist.get_global_rank`` (a sub-group of a larger job addresses its own members)."""

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None):
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if world > 1 else 0
        self.rank, self.world = rank, world
        self._device_capable = None       # does the group's backend move device memory (RCCL)?  decided on first use

    def _peer(self, group_rank: int) -> int:
        """Global rank of member ``group_rank`` of this pipeline's group."""
        if self.group is None:
            return group_rank
        return dist.get_global_rank(self.group, group_rank)

    # -- transport --------------------------------------------------------------------------------
    # RCCL ("nccl") takes device tensors and orders the transfer against the current stream.  gloo moves HOST memory:
    # handed a device tensor it reads / writes the raw pointer from the CPU with no stream ordering at all (on this
    # platform device memory is host-visible, so nothing fails -- the peer just receives stale bytes: found by the
    # world-2 decode rehearsal of round 3, profiles/r03_decode_world2_rehearsal.log).  Device tensors over gloo are
    # therefore staged through host memory here: .cpu() synchronises with the producing stream, copy_() back is
    # stream-ordered.  CPU tensors (the gloo unit tests) and RCCL go straight through.
    def _staged(self, t: torch.Tensor) -> bool:
        # by capability, not by name equality: a group created without an explicit backend reports a composite
        # ("cpu:gloo,cuda:nccl" / "undefined") and still moves device tensors with RCCL
        if not t.is_cuda:
            return False
        if self._device_capable is None:
            try:
                cfg = str(dist.get_backend_config(self.group))
            except Exception:
                cfg = str(dist.get_backend(self.group))
            self._device_capable = "nccl" in cfg
        return not self._device_capable

    def _send(self, t: torch.Tensor, dst: int) -> None:
        dist.send(t.cpu() if self._staged(t) else t, dst=dst, group=self.group)

    def _recv(self, t: torch.Tensor, src: int) -> None:
        if self._staged(t):
            host = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(host, src=src, group=self.group)
            t.copy_(host)
        else:
            dist.recv(t, src=src, group=self.group)

    def _isend(self, t: torch.Tensor, dst: int):
        return dist.isend(t.cpu() if self._staged(t) else t, dst=dst, group=self.group)

    def _irecv(self, t: torch.Tensor, src: int):
        if self._staged(t):
            host = torch.empty(t.shape, dtype=t.dtype)
            return _StagedRecv(dist.irecv(host, src=src, group=self.group), host, t)
        return dist.irecv(t, src=src, group=self.group)

    @property
    def is_first(self) -> bool:
        return self.rank == 0

    @property
    def is_last(self) -> bool:
        return self.rank == self.world - 1

    # -- one hop (blocking forms, kept for simple callers) ------------------------------------------
    def recv_hidden(self, buf: torch.Tensor) -> torch.Tensor:
        """Receive the previous stage's output into ``buf`` (no-op on the first stage)."""
        if self.world > 1 and not self.is_first:
            self._recv(buf, self._peer(self.rank - 1))
        return buf

    def send_hidden(self, h: torch.Tensor) -> None:
        """Send this stage's output to the next stage (no-op on the last stage)."""
        if self.world > 1 and not self.is_last:
            self._send(h.contiguous(), self._peer(self.rank + 1))

    def hop_round_trip_us(self, buf: torch.Tensor, iters: int = 20, sync: Optional[Callable[[], None]] = None) -> Optional[float]:
        """Round-trip time in microseconds of ``buf`` across THIS rank's downstream boundary (rank -> rank + 1 -> rank),
        measured on the upstream rank over ``iters`` dependent round trips between two synchronisations; the boundaries
        are measured one after the other with a barrier in between, so no two of them share a link or a host thread.
        Every rank of the group must call it; the last rank (no downstream neighbour) and a world of 1 return None.
        Half of it is the per-hop latency a batch-1 decode token pays at each stage boundary."""
        if self.world == 1:
            return None
        sync = sync or (torch.cuda.synchronize if buf.is_cuda else (lambda: None))
        mine = None
        for b in range(self.world - 1):
            if self.rank == b:
                nxt = self._peer(b + 1)
                for timed in (False, True):                 # one untimed pass: communicator / connection set-up
                    sync()
                    t0 = time.perf_counter()
                    for _ in range(iters if timed else 2):
                        self._send(buf, nxt)
                        self._recv(buf, nxt)
                    sync()
                    mine = (time.perf_counter() - t0) / iters * 1e6
            elif self.rank == b + 1:
                prv = self._peer(b)
                for n in (2, iters):
                    for _ in range(n):
                        self._recv(buf, prv)
                        self._send(buf, prv)
                sync()
            dist.barrier(group=self.group)
        return mine

    # -- prefill-style streaming of micro-batches -------------------------------------------------
    def run_microbatches(self, stage_fn: Callable[[torch.Tensor], torch.Tensor], inputs: List[torch.Tensor],
                         recv_buf: torch.Tensor, collect: bool = True,
                         stats: Optional[PipelineStats] = None) -> List[torch.Tensor]:
        """Stream ``inputs`` (used by the first stage; later stages only need their count and shape) through the
        pipeline with the overlapped schedule of the module docstring.  ``recv_buf`` gives shape / dtype / device
        of the hidden state (it is slot 0 of the receive ring).  ``stage_fn(h)`` returns this stage's OUTPUT, which
        is what hops on: stages are data-dependent.  Returns the last stage's outputs (empty list elsewhere, or
        when ``collect`` is False: a throughput run that does not keep them).  ``stats``: a PipelineStats that receives
        this rank's timeline (waiting for the hop in / own layers / waiting for a send slot) and the bytes it moved."""
        n = len(inputs)
        span = stats.span if stats is not None else (lambda kind: _NULL)
        outs: List[torch.Tensor] = []
        if self.world == 1:
            for x in inputs:
                h = stage_fn(x)
                if collect:
                    outs.append(h)
            return outs
        rbuf = [recv_buf, torch.empty_like(recv_buf)] if not self.is_first else None
        sbuf = [None, None]                     # send ring, allocated from the first output's shape
        rwork = [None, None]
        swork = [None, None]
        src = self._peer(self.rank - 1) if not self.is_first else None
        dst = self._peer(self.rank + 1) if not self.is_last else None
        staged = self._staged(recv_buf)
        if not self.is_first and n:
            rwork[0] = self._irecv(rbuf[0], src)
        for b in range(n):
            more = not self.is_first and b + 1 < n          # slot (b+1) % 2 last held micro-batch b-1, consumed by stage_fn(b-1)
            if more and staged:
                # host-memory backend: the receive lands in a host buffer, so it can be posted before b is computed
                rwork[(b + 1) % 2] = self._irecv(rbuf[(b + 1) % 2], src)
            if not self.is_first:
                with span("recv_wait"):
                    rwork[b % 2].wait()
                x = rbuf[b % 2]
            else:
                x = inputs[b]
            with span("compute"):
                h = stage_fn(x)
            s = b % 2
            if not self.is_last:
                if swork[s] is not None:
                    with span("send_wait"):
                        swork[s].wait()         # isend(b-2) done: its ring slot is free
            if stats is not None:
                nb = h.numel() * h.element_size()
                stats.microbatches += 1
                if not self.is_last:
                    stats.bytes_sent += nb
                    stats.hops_sent += 1
                if not self.is_first:
                    stats.bytes_received += nb
                    stats.hops_received += 1
            if staged:
                if not self.is_last:            # .cpu() IS the copy out of the stage's buffer
                    swork[s] = self._isend(h, dst)
            else:
                # RCCL (and CPU tensors over gloo): the send of b and the receive of b+1 are ONE group
                # (dist.batch_isend_irecv = ncclGroupStart / End): torch's NCCL backend enqueues the point-to-point ops of
                # a communicator on one stream in issue order, so a receive of b+1 posted BEFORE the send of b would hold
                # that send back until the upstream stage has finished b+1 -- the downstream stage would idle for a whole
                # stage time per micro-batch.  Grouped, neither orders the other; both fly under stage_fn(b+1).
                ops = []
                if not self.is_last:
                    if sbuf[s] is None:
                        sbuf[s] = torch.empty_like(h, memory_format=torch.contiguous_format)
                    sbuf[s].copy_(h)
                    ops.append(dist.P2POp(dist.isend, sbuf[s], dst, group=self.group))
                if more:
                    ops.append(dist.P2POp(dist.irecv, rbuf[(b + 1) % 2], src, group=self.group))
                if ops:
                    works = dist.batch_isend_irecv(ops)
                    # (a coalesced batch returns ONE work for the group, otherwise one per op in order)
                    if not self.is_last:
                        swork[s] = works[0]
                    if more:
                        rwork[(b + 1) % 2] = works[-1]
            if self.is_last and collect:
                outs.append(h.clone() if (rbuf is not None and any(h is r for r in rbuf)) else h)
        with span("send_wait"):
            for w in swork:
                if w is not None:
                    w.wait()
        return outs

    # -- greedy decode ------------------------------------------------------------------------------
    def decode(self, first_token: int, n_tokens: int, embed_fn: Callable[[torch.Tensor], torch.Tensor],
               stage_fn: Callable[[torch.Tensor, int], torch.Tensor], head_fn: Callable[[torch.Tensor], torch.Tensor],
               hidden_buf: torch.Tensor, token_buf: torch.Tensor) -> List[int]:
        """Batch-1 greedy decode of ``n_tokens`` tokens.

        first stage: ``embed_fn(token [1] int64) -> hidden``; every stage: ``stage_fn(hidden, step)``;
        last stage: ``head_fn(hidden) -> next token [1] int64``, sent point-to-point to the first stage -- the only
        rank that needs it (no collective on the critical path).  A batch-1 pipeline is sequential by nature
        (each token needs the previous one), so G GPUs give memory capacity, not speed-up.
        Returns the generated ids on the first and on the last stage, an empty list on the stages in between."""
        token_buf.fill_(int(first_token))
        keeps = self.is_first or self.is_last
        generated = torch.zeros(n_tokens if keeps else 0, dtype=token_buf.dtype, device=token_buf.device)
        for step in range(n_tokens):
            if self.is_first:
                h = embed_fn(token_buf)
            else:
                h = self.recv_hidden(hidden_buf)
            h = stage_fn(h, step)
            if self.is_last:
                token_buf.copy_(head_fn(h).reshape(-1)[:1])
                if self.world > 1:
                    self._send(token_buf, self._peer(0))
            else:
                self.send_hidden(h)
                if self.is_first:
                    self._recv(token_buf, self._peer(self.world - 1))
            if keeps:
                generated[step:step + 1].copy_(token_buf)     # stays on the device: no host sync per token
        return generated.tolist()
