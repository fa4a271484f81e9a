"""Whole-layer pipeline sharding over the GPUs of one node.

The reference's only multi-GPU inference mode is accelerate's ``device_map="auto"``
(mxq_quant/main.py:23, lib/prune.py:371-378): decoder layers are placed on successive GPUs
and the hidden state hops GPU -> GPU with ``.to(dev)`` inside ONE process.  The native
counterpart: one process per GPU, rank ``r`` of ``G`` owns layers ``[L*r/G, L*(r+1)/G)``, and the
only exchange is the ``[tokens, hidden]`` fp16 activation moving to the next stage with
``torch.distributed`` point-to-point ``send``/``recv`` (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests) plus, for greedy decode, the 8-byte next-token id going from the last
stage back to the first.  No all-reduce / all-gather is needed, so the per-link ring bound of
xGMI never enters.

Nothing here touches the HIP library: the stage computation is a callable, so the schedule is
unit-tested on CPU with gloo (tests/test_pipeline_gloo.py) and used unchanged with RCCL.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def layer_range(rank: int, world: int, n_layers: int) -> range:
    """Layers owned by ``rank``: contiguous, sizes differ by at most one."""
    return range(n_layers * rank // world, n_layers * (rank + 1) // world)


class LayerPipeline:
    """Point-to-point activation pipeline between consecutive ranks of ``group``."""

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None):
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if world > 1 else 0
        self.rank, self.world = rank, world

    @property
    def is_first(self) -> bool:
        return self.rank == 0

    @property
    def is_last(self) -> bool:
        return self.rank == self.world - 1

    # -- one hop ------------------------------------------------------------------------------
    def recv_hidden(self, buf: torch.Tensor) -> torch.Tensor:
        """Receive the previous stage's output into ``buf`` (no-op on the first stage)."""
        if self.world > 1 and not self.is_first:
            dist.recv(buf, src=self.rank - 1, group=self.group)
        return buf

    def send_hidden(self, h: torch.Tensor) -> None:
        """Send this stage's output to the next stage (no-op on the last stage)."""
        if self.world > 1 and not self.is_last:
            dist.send(h.contiguous(), dst=self.rank + 1, group=self.group)

    # -- prefill-style streaming of micro-batches -------------------------------------------------
    def run_microbatches(self, stage_fn: Callable[[torch.Tensor], torch.Tensor], inputs: List[torch.Tensor],
                         recv_buf: torch.Tensor, collect: bool = True) -> List[torch.Tensor]:
        """Stream ``inputs`` (used by the first stage; later stages only need their count and
        shape) through the pipeline.  Stage ``r`` works on micro-batch ``b`` while stage ``r+1``
        works on ``b-1``.  Returns the last stage's outputs (empty list elsewhere, or when
        ``collect`` is False: a throughput run that does not keep them)."""
        outs = []
        for x in inputs:
            h = x if self.is_first else self.recv_hidden(recv_buf)
            h = stage_fn(h)
            if not self.is_last:
                self.send_hidden(h)
            elif collect:
                outs.append(h.clone() if h is recv_buf else h)
        return outs

    # -- greedy decode ------------------------------------------------------------------------------
    def decode(self, first_token: int, n_tokens: int, embed_fn: Callable[[torch.Tensor], torch.Tensor],
               stage_fn: Callable[[torch.Tensor, int], torch.Tensor], head_fn: Callable[[torch.Tensor], torch.Tensor],
               hidden_buf: torch.Tensor, token_buf: torch.Tensor) -> List[int]:
        """Batch-1 greedy decode of ``n_tokens`` tokens.

        first stage: ``embed_fn(token [1] int64) -> hidden``; every stage: ``stage_fn(hidden, step)``;
        last stage: ``head_fn(hidden) -> next token [1] int64``, sent back to the first stage.
        A batch-1 pipeline is sequential by nature (each token needs the previous one), so G
        GPUs give memory capacity, not speed-up.  Returns the generated ids (same on every rank)."""
        token_buf.fill_(int(first_token))
        generated = torch.zeros(n_tokens, dtype=token_buf.dtype, device=token_buf.device)
        for step in range(n_tokens):
            if self.is_first:
                h = embed_fn(token_buf)
            else:
                h = self.recv_hidden(hidden_buf)
            h = stage_fn(h, step)
            if self.is_last:
                token_buf.copy_(head_fn(h).reshape(-1)[:1])
            else:
                self.send_hidden(h)
            # next-token id: last stage -> everyone (8 bytes); a broadcast keeps every rank's
            # bookkeeping identical and is latency-equivalent to the single send to rank 0
            if self.world > 1:
                dist.broadcast(token_buf, src=self.world - 1, group=self.group)
            generated[step:step + 1].copy_(token_buf)     # stays on the device: no host sync per token
        return generated.tolist()
