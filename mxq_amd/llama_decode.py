"""Llama-2-7B-shaped greedy-decode stage on packed MXQ weights (BASELINE config 3).

Only the quantised Linears are this repo's kernels (``packing.linear`` -> mxq_gemv_f16 for one
token); attention over the KV cache, RMSNorm and RoPE are plain PyTorch-ROCm ops -- plumbing
around the hot path, not part of it.  q/k/v and gate/up share their input,
so their packed weights are concatenated along the output dimension (format v1 blocks are
independent per 16 rows): 4 GEMV launches per layer instead of 7.

Weights are synthetic (no checkpoint offline): ``randn * 0.02`` seeded per (layer, linear) as in
bench.py.  Everything in ``step`` is stream-ordered with device-resident position / token
tensors, so a whole stage step can be captured in a hipGraph (``capture()``).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from . import _lib
from . import llama_shapes as LS
from . import packing


_concat_packed = packing.concat_packed


class ContextWindow:
    """Host-side mirror of the device-resident decode position.  The position tensor is advanced on the device
    (graph replay), where nothing can raise; every host entry point that advances it goes through ``take`` first,
    so a decode past the KV cache fails here instead of overrunning the cache (csrc/decode_ops.hip poisons its
    output for such a position, but never writes outside the cache)."""

    def __init__(self, max_ctx: int):
        if max_ctx <= 0:
            raise ValueError("max_ctx must be positive")
        self.max_ctx, self.pos = int(max_ctx), 0

    def take(self, n: int = 1) -> int:
        """Reserve the next ``n`` positions; returns the first.  Raises if they do not fit the cache."""
        if n < 0:
            raise ValueError("cannot take a negative number of positions")
        if self.pos + n > self.max_ctx:
            raise RuntimeError(f"decode position {self.pos} + {n} exceeds the KV cache (max_ctx = {self.max_ctx}); "
                               "reset() or build the stage with a larger max_ctx")
        first = self.pos
        self.pos += n
        return first

    def reset(self):
        self.pos = 0


class DecodeStage:
    SPLIT_KEYS = 128          # csrc/decode_ops.hip SPLIT_MIN: contexts up to this many keys are one workgroup per head

    def __init__(self, layers, dev, max_ctx: int = 512, hidden: int = LS.HIDDEN, inter: int = LS.INTERMEDIATE,
                 heads: int = 32, first: bool = True, last: bool = True, vocab: int = 32000, fused: bool = True,
                 compact: bool = False, staging: str = "swiglu", arena: bool = True):
        self.layers, self.dev, self.max_ctx = list(layers), dev, max_ctx
        self.hidden, self.inter, self.heads, self.hd = hidden, inter, heads, hidden // heads
        self.first, self.last, self.compact = first, last, compact
        # fused: RMSNorm / SwiGLU / residual folded into the GEMV launches and one RoPE + cache-append
        # + attention kernel per layer (5 launches per layer); otherwise plain torch ops around 4 GEMVs
        self.fused = fused and self.hd == 128
        # staging (round 5): "swiglu" = gate|up applies the SwiGLU in its final reduction and writes the activation in the form
        # down's workgroups stage it in (they copy 22 KB instead of staging 44 KB with 11 k exponentials each); "consumer" =
        # the round-4 launches.  Bit-identical; +3.6 % (tools/ab_decode.py, profiles/r05_decode_staging_ab.txt, which also
        # records the two further producer-side steps that lost)
        if staging not in ("consumer", "swiglu"):
            raise ValueError("staging must be 'consumer' or 'swiglu'")
        self.swiglu_in_producer = staging == "swiglu" and self.fused and (2 * inter) % 32 == 0 and hidden % 256 == 0
        self.launches_per_layer = 5 if self.fused else None     # q|k|v, attention, o, gate|up, down (else: torch ops around 4 GEMVs)
        self.w = []
        for li in self.layers:
            def mk(idx, N, K):
                g = torch.Generator(device=dev).manual_seed(1000 * li + idx)
                W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
                return packing.quantize_pack(W, compact_meta=compact)     # compact: fp16 zero-points, 3.75 bit/weight
            qkv = _concat_packed([mk(0, hidden, hidden), mk(1, hidden, hidden), mk(2, hidden, hidden)])
            o = mk(3, hidden, hidden)
            gu = _concat_packed([mk(4, inter, hidden), mk(5, inter, hidden)])
            down = mk(6, hidden, inter)
            self.w.append((qkv, o, gu, down))
        if arena:            # one allocation for the stage's packed weights, in streaming order (+0.35 %, tools/ab_decode.py)
            self._to_arena()
        n = len(self.layers)
        self.k_cache = torch.zeros(n, heads, max_ctx, self.hd, device=dev, dtype=torch.float16)
        self.v_cache = torch.zeros_like(self.k_cache)
        self.norm_w = torch.ones(hidden, device=dev, dtype=torch.float16)
        inv = 1.0 / (10000 ** (torch.arange(0, self.hd, 2, device=dev).float() / self.hd))
        ang = torch.arange(max_ctx, device=dev).float()[:, None] * inv[None, :]
        self.cos, self.sin = ang.cos(), ang.sin()                     # [max_ctx, hd/2]
        self.pos = torch.zeros(1, dtype=torch.int64, device=dev)      # device-resident position
        self.window = ContextWindow(max_ctx)                          # its host-side mirror (bounds check)
        self.ctx_ids = torch.arange(max_ctx, device=dev)
        # embedding and lm_head are seeded SEPARATELY: drawn one after the other from one generator, a stage that is
        # last but not first got the embedding's draw as its lm_head, i.e. a pipeline whose tokens could never match
        # the single-process run (found by the world-2 rehearsal of round 3, tools/decode_bench.py --verify)
        if first:
            g = torch.Generator(device=dev).manual_seed(99)
            self.embed = (torch.randn(vocab, hidden, generator=g, device=dev) * 0.02).half()
        if last:
            g = torch.Generator(device=dev).manual_seed(98)
            self.lm_head = (torch.randn(vocab, hidden, generator=g, device=dev) * 0.02).half()   # plain fp16 Linear
        self.rope_row = torch.zeros(2 * (self.hd // 2), dtype=torch.float32, device=dev)   # cos | sin of the current position
        # contexts beyond 128 keys: a head's keys split over up to 8 workgroups (include/mxq_hip.h: mxq_attn_decode_split_f16)
        # -- chosen on the HOST, which mirrors the device position (ContextWindow): up to SPLIT_KEYS keys the one-workgroup
        # launch (32 workgroups; the split launch's 256 cost 0.36 us per layer there), beyond it the split one; a captured
        # stage holds one graph of each and replays the one the position calls for
        self.attn_splits = (16 if max_ctx >= 1024 else 8) if (self.fused and max_ctx > self.SPLIT_KEYS) else 1
        self._long_ctx = False
        self._pin_long = None        # capture (and its eager pre-run) pins the attention launch; None: by the host position
        self._attn_ws = (torch.zeros(_lib.load().mxq_attn_split_workspace_bytes(heads, self.attn_splits), dtype=torch.uint8, device=dev)
                         if self.attn_splits > 1 else None)
        self._graph = None
        self._head_ws = None
        self._head_tok = None
        self._h_in = torch.zeros(1, hidden, device=dev, dtype=torch.float16)
        self._h_out = torch.zeros(1, hidden, device=dev, dtype=torch.float16)

    def _to_arena(self):
        """All packed weights of the stage in ONE allocation, in the order a token streams them."""
        total = sum((p.qweight.numel() * 4 + 255) // 256 * 256 + (p.rowmeta.numel() * 4 + 255) // 256 * 256 for ws in self.w for p in ws)
        self._arena = torch.empty(total, dtype=torch.uint8, device=self.dev)
        off, new = 0, []
        for ws in self.w:
            row = []
            for p in ws:
                nq, nr = p.qweight.numel() * 4, p.rowmeta.numel() * 4
                q = self._arena[off:off + nq].view(torch.int32)
                q.copy_(p.qweight)
                off += (nq + 255) // 256 * 256
                r = self._arena[off:off + nr].view(torch.float32).view(p.rowmeta.shape)
                r.copy_(p.rowmeta)
                off += (nr + 255) // 256 * 256
                row.append(packing.PackedMXQ(q, r, p.N, p.K, p.compact))
            new.append(tuple(row))
        self.w = new

    def packed_bytes(self) -> int:
        return sum(p.nbytes() for ws in self.w for p in ws)

    def _rms(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).half() * self.norm_w

    def _rope(self, t):                      # t [heads, hd], rotate-half convention
        c = self.cos.index_select(0, self.pos)[0]
        s = self.sin.index_select(0, self.pos)[0]
        t1, t2 = t[:, : self.hd // 2].float(), t[:, self.hd // 2:].float()
        return torch.cat([t1 * c - t2 * s, t2 * c + t1 * s], dim=-1).half()

    def embed_token(self, tok):
        return self.embed.index_select(0, tok)

    def head(self, h, out: Optional[torch.Tensor] = None):
        """Final RMSNorm + lm_head + greedy argmax -> token id tensor [1] (int64).  Fused stages of Llama width run it
        as one native launch pair (csrc/decode_ops.hip, mxq_lmhead_argmax_f16); otherwise plain torch ops.
        ``out`` (int64 [1], device): where the native pair writes the id -- the token loop passes its token buffer and
        saves a copy launch per token."""
        if self.fused and self.hidden == 4096 and h.shape[0] == 1:
            if self._head_ws is None:
                self._head_ws = torch.empty(2 * 1024, dtype=torch.float32, device=self.dev)
                self._head_tok = torch.zeros(1, dtype=torch.int64, device=self.dev)
            tok = out if out is not None and out.dtype == torch.int64 and out.numel() == 1 and out.is_contiguous() else self._head_tok
            hc = h.contiguous()
            lib = _lib.load()
            _lib.check(lib.mxq_lmhead_argmax_f16(hc.data_ptr(), self.norm_w.data_ptr(), 1e-5, self.lm_head.data_ptr(),
                                                 self.lm_head.shape[0], self.hidden, self._head_ws.data_ptr(), 1024,
                                                 tok.data_ptr(),
                                                 torch.cuda.current_stream(self.dev).cuda_stream), "mxq_lmhead_argmax_f16")
            return tok
        logits = torch.nn.functional.linear(self._rms(h), self.lm_head)
        return logits.argmax(dim=-1)

    def _rope_row(self):
        """cos / sin of the current (device-resident) position, gathered ONCE per step for all of this stage's layers."""
        _lib.check(_lib.load().mxq_rope_row_f32(self.pos.data_ptr(), self.cos.data_ptr(), self.sin.data_ptr(),
                                                self.rope_row.data_ptr(), self.hd // 2, self.max_ctx,
                                                torch.cuda.current_stream(self.dev).cuda_stream), "mxq_rope_row_f32")

    def _attn(self, qkv, i):
        out = torch.empty((1, self.hidden), dtype=torch.float16, device=self.dev)
        lib = _lib.load()
        st = torch.cuda.current_stream(self.dev).cuda_stream
        if self.attn_splits > 1 and self._long_ctx:
            _lib.check(lib.mxq_attn_decode_split_f16(qkv.data_ptr(), self.k_cache[i].data_ptr(), self.v_cache[i].data_ptr(),
                                                     self.pos.data_ptr(), self.rope_row.data_ptr(), out.data_ptr(), self.heads,
                                                     self.hd, self.max_ctx, self.attn_splits, self._attn_ws.data_ptr(), st),
                       "mxq_attn_decode_split_f16")
        else:
            _lib.check(lib.mxq_attn_decode_row_f16(qkv.data_ptr(), self.k_cache[i].data_ptr(), self.v_cache[i].data_ptr(),
                                                   self.pos.data_ptr(), self.rope_row.data_ptr(), out.data_ptr(), self.heads,
                                                   self.hd, self.max_ctx, st), "mxq_attn_decode_row_f16")
        return out

    def step(self, h, rope_done: bool = False):
        """One token through this stage's layers.  h [1, hidden] fp16.  ``rope_done``: the caller has already gathered the
        position's rotary row (the token loop does it together with the embedding look-up)."""
        if self.window.pos >= self.max_ctx:
            raise RuntimeError(f"decode position {self.window.pos} is outside the KV cache (max_ctx = {self.max_ctx})")
        if self.fused:
            if self._pin_long is not None:
                self._long_ctx = self._pin_long                             # capture and its eager pre-run: pinned
            elif not torch.cuda.is_current_stream_capturing():
                self._long_ctx = self.window.pos + 1 > self.SPLIT_KEYS      # eager step: the host knows the position
            if not rope_done:
                self._rope_row()
            for i, (qkv, o, gu, down) in enumerate(self.w):
                y = packing.linear_fused(h, qkv, 1, self.norm_w)                  # RMSNorm -> q|k|v
                a = self._attn(y, i)                                              # RoPE + cache + attention
                h = packing.linear_fused(a, o, 0, residual=h)                     # o_proj + skip
                if self.swiglu_in_producer:
                    act, act_sum = packing.linear_swiglu(h, gu, self.norm_w)          # RMSNorm -> gate|up -> SwiGLU
                    h = packing.linear_staged(act, act_sum, down, residual=h)         # down_proj + skip
                else:
                    g = packing.linear_fused(h, gu, 1, self.norm_w)               # RMSNorm -> gate|up
                    h = packing.linear_fused(g, down, 2, residual=h)              # SwiGLU -> down_proj + skip
            return h
        scale = 1.0 / math.sqrt(self.hd)
        mask = (self.ctx_ids <= self.pos)[None, None, :]              # [1, 1, max_ctx]
        for i, (qkv, o, gu, down) in enumerate(self.w):
            x = self._rms(h)
            y = packing.linear(x, qkv)                                # GEMV kernel (1 token)
            q, k, v = (y[0, j * self.hidden:(j + 1) * self.hidden].view(self.heads, self.hd) for j in range(3))
            q, k = self._rope(q), self._rope(k)
            self.k_cache[i].index_copy_(1, self.pos, k[:, None, :])
            self.v_cache[i].index_copy_(1, self.pos, v[:, None, :])
            att = torch.baddbmm(torch.zeros(1, device=self.dev, dtype=torch.float16), q[:, None, :],
                                self.k_cache[i].transpose(1, 2), alpha=scale)          # [heads, 1, ctx]
            att = att.float().masked_fill(~mask, float("-inf")).softmax(-1).half()
            a = torch.bmm(att, self.v_cache[i]).reshape(1, self.hidden)
            h = h + packing.linear(a, o)
            x = self._rms(h)
            gu_y = packing.linear(x, gu)
            act = torch.nn.functional.silu(gu_y[:, : self.inter].float()).half() * gu_y[:, self.inter:]
            h = h + packing.linear(act, down)
        return h

    def advance(self):
        self.window.take(1)
        self.pos += 1

    # -- hipGraph capture of one stage step --------------------------------------------------------
    def capture(self):
        for _ in range(2):                                            # warm-up outside capture
            self._h_out.copy_(self.step(self._h_in))
        self.pos.zero_()
        torch.cuda.synchronize()

        def body():
            self._h_out.copy_(self.step(self._h_in))
            self.pos += 1
        self._graph, self._graph_long = self._capture_short_and_long(body)
        self.pos.zero_()
        self.window.reset()
        return self

    def _capture_short_and_long(self, body):
        """One graph of ``body`` with the one-workgroup attention launch and -- when the cache is longer than SPLIT_KEYS --
        one with the split launch; ``_pick`` replays the one the host-side position calls for."""
        graphs = []
        try:
            for long_ctx in ((False, True) if self.attn_splits > 1 else (False,)):
                self._pin_long = long_ctx          # step() takes the pinned launch, eager or captured
                if long_ctx:                       # eager pre-run of the SPLIT launch: its first launch (and, for caches whose
                    body()                         # LDS need exceeds 64 KiB, its hipFuncSetAttribute) happens outside capture
                    self.pos.zero_()
                    torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    body()
                graphs.append(g)
        finally:
            self._pin_long = None
            self._long_ctx = False
        return graphs[0], (graphs[1] if len(graphs) > 1 else None)

    def _pick(self, short, long_, position: int):
        """The graph for a token at ``position`` (it attends to position + 1 keys)."""
        return long_ if (long_ is not None and position + 1 > self.SPLIT_KEYS) else short

    def step_graph(self, h):
        p = self.window.take(1)                # the replay advances the device position by one
        self._h_in.copy_(h)
        self._pick(self._graph, self._graph_long, p).replay()
        return self._h_out

    def capture_token_loop(self, token_buf: torch.Tensor):
        """Single-stage decode (this stage is first AND last): capture token -> embedding -> layers -> final
        norm -> lm_head -> argmax -> token as ONE graph that updates ``token_buf`` [1] int64 in place, so a
        decoded token costs one graph launch instead of ~20 small host-launched kernels around the layer graph."""
        if getattr(self, "embed", None) is None or getattr(self, "lm_head", None) is None:
            raise ValueError("capture_token_loop needs the first and the last stage in one process")

        # Llama-width fused stages: the token's two table look-ups (embedding row, rotary row) are ONE launch, and the head's
        # second launch also appends the id to a device-side list and advances the position: 4 + 32 x 5 launches per token,
        # nothing between two graph replays
        native = (self.fused and self.hidden == 4096 and token_buf.dtype == torch.int64 and token_buf.numel() == 1
                  and token_buf.is_contiguous())
        self._generated = torch.zeros(self.max_ctx, dtype=torch.int64, device=self.dev) if native else None
        lib = _lib.load()

        def one():
            if not native:
                t = self.head(self.step(self.embed_token(token_buf)), out=token_buf)
                if t is not token_buf:
                    token_buf.copy_(t.reshape(-1)[:1])
                self.pos += 1
                return
            st = torch.cuda.current_stream(self.dev).cuda_stream
            _lib.check(lib.mxq_embed_rope_row(token_buf.data_ptr(), self.embed.data_ptr(), self.embed.shape[0], self.hidden,
                                              self._h_in.data_ptr(), self.pos.data_ptr(), self.cos.data_ptr(), self.sin.data_ptr(),
                                              self.rope_row.data_ptr(), self.hd // 2, self.max_ctx, st), "mxq_embed_rope_row")
            h = self.step(self._h_in, rope_done=True).contiguous()
            if self._head_ws is None:
                self._head_ws = torch.empty(2 * 1024, dtype=torch.float32, device=self.dev)
            _lib.check(lib.mxq_lmhead_argmax_advance_f16(h.data_ptr(), self.norm_w.data_ptr(), 1e-5, self.lm_head.data_ptr(),
                                                         self.lm_head.shape[0], self.hidden, self._head_ws.data_ptr(), 1024,
                                                         token_buf.data_ptr(), self.pos.data_ptr(), self._generated.data_ptr(),
                                                         self.max_ctx, st), "mxq_lmhead_argmax_advance_f16")
        for _ in range(2):
            one()
        self.pos.zero_()
        torch.cuda.synchronize()
        self._tgraph, self._tgraph_long = self._capture_short_and_long(one)
        self.pos.zero_()
        self.window.reset()
        return self

    def decode_tokens(self, token_buf: torch.Tensor, first_token: int, n_tokens: int):
        """Greedy decode with the graph of ``capture_token_loop``; returns the generated ids."""
        p0 = self.window.take(n_tokens)        # every replay advances the device position by one
        token_buf.fill_(int(first_token))
        if getattr(self, "_generated", None) is not None:      # the graph itself appends the ids at generated[position]
            for i in range(n_tokens):
                self._pick(self._tgraph, self._tgraph_long, p0 + i).replay()
            return self._generated[p0:p0 + n_tokens].tolist()
        out = torch.zeros(n_tokens, dtype=token_buf.dtype, device=token_buf.device)
        for i in range(n_tokens):
            self._pick(self._tgraph, self._tgraph_long, p0 + i).replay()
            out[i:i + 1].copy_(token_buf)
        return out.tolist()

    def reset(self):
        self.window.reset()
        self.pos.zero_()
        self.k_cache.zero_()
        self.v_cache.zero_()

    def seek(self, position: int, seed: int = 5):
        """Benchmark helper: pretend ``position`` tokens have been decoded -- the KV cache rows below it filled with seeded
        random keys / values, the device and host positions set -- so that the next tokens run at that context length."""
        if not 0 <= position < self.max_ctx:
            raise ValueError("position outside the KV cache")
        g = torch.Generator(device=self.dev).manual_seed(seed)
        self.k_cache.zero_()
        self.v_cache.zero_()
        self.k_cache[:, :, :position].copy_(torch.randn(self.k_cache[:, :, :position].shape, generator=g, device=self.dev).half())
        self.v_cache[:, :, :position].copy_(torch.randn(self.v_cache[:, :, :position].shape, generator=g, device=self.dev).half())
        self.pos.fill_(position)
        self.window.reset()
        self.window.take(position)


def decode_pipeline_figure(pipe, dev, tokens: int = 32, ctx: int = 64, layers: int = LS.N_LAYERS, compact: bool = False,
                           verify: bool = False, graph: bool = True, warm: int = 8, dist=None, backend: str = "nccl",
                           start: int = 0) -> dict:
    """BASELINE configs[2] as one measurement: greedy decode of ``tokens`` tokens at batch 1 with the decoder layers
    sharded over ``pipe``'s ranks (rank r owns ``layer_range(r, world, layers)``), the hidden state and the next-token
    id moving point to point (pipeline.LayerPipeline.decode).  Every rank calls it; every rank returns the same kind of
    dict, rank 0's carries the token ids.  ``verify`` (world > 1): rank 0 also decodes the same tokens in ONE process
    (all layers, token-loop graph) and reports whether the ids are identical (``tokens_equal_single_process``).
    Used by tools/decode_bench.py and by bench.py's ``decode_pipeline`` side figure (reference analogue: accelerate's
    layer placement, mxq_quant/main.py:23, lib/prune.py:371-378)."""
    import time
    from .pipeline import layer_range
    world, rank = pipe.world, pipe.rank
    stage = DecodeStage(layer_range(rank, world, layers), dev, max_ctx=ctx, first=pipe.is_first, last=pipe.is_last,
                        compact=compact)
    if graph:
        stage.capture()
    hbuf = torch.zeros(1, stage.hidden, device=dev, dtype=torch.float16)
    tbuf = torch.zeros(1, dtype=torch.int64, device=dev)

    def stage_fn(h, step):
        if not graph:
            out = stage.step(h)
            stage.advance()
            return out
        return stage.step_graph(h)

    single = world == 1 and graph
    if single:
        stage.capture_token_loop(tbuf)     # token -> token in one graph (embedding, layers, head, argmax)

    def run(n):
        stage.reset()
        if start:                              # decode at a long context: the cache below `start` holds seeded random rows
            stage.seek(start)
        if single:
            return stage.decode_tokens(tbuf, 1, n)
        return pipe.decode(1, n, stage.embed_token if pipe.is_first else None, stage_fn,
                           stage.head if pipe.is_last else None, hbuf, tbuf)

    if start + max(warm, tokens) > ctx:
        raise ValueError("ctx must hold the warm-up run and the timed run (each starts at position `start`)")
    run(warm)
    torch.cuda.synchronize(dev)
    if dist is not None and world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    toks = run(tokens)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    stage_launches = stage.launches_per_layer
    nbytes = torch.tensor([float(stage.packed_bytes())], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    if dist is not None and world > 1:
        dist.all_reduce(nbytes)
    # world > 1: what one token's trip is made of, per rank -- this stage's layers alone (graph replays between two HIP
    # events, nothing hopping) and the round trip of the [1, hidden] row across the boundary to the next rank.  A batch-1
    # pipeline is sequential: ms_per_token ~ sum of the stages + one hop per boundary + the id's way back.
    per_rank = None
    if dist is not None and world > 1:
        from .pipeline import gather_reports
        stage.reset()
        n_rep = min(16, ctx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_rep):
            stage_fn(hbuf, 0)
        e1.record()
        torch.cuda.synchronize(dev)
        stage_us = e0.elapsed_time(e1) / n_rep * 1e3
        rtt = pipe.hop_round_trip_us(hbuf, iters=20, sync=lambda: torch.cuda.synchronize(dev))
        per_rank = gather_reports({"rank": rank, "layers": len(stage.layers), "stage_us_per_token": round(stage_us, 2),
                                   "hop_round_trip_us_to_next_rank": None if rtt is None else round(rtt, 2)}, pipe.group)
    equal = None
    if verify and world > 1 and rank == 0:
        del stage
        ref = DecodeStage(range(layers), dev, max_ctx=ctx, first=True, last=True, compact=compact)
        rtok = torch.zeros(1, dtype=torch.int64, device=dev)
        ref.capture_token_loop(rtok)
        ref.reset()
        want = ref.decode_tokens(rtok, 1, tokens)
        equal = want == toks
        bad = next((i for i, (a, b) in enumerate(zip(want, toks)) if a != b), None)
        del ref
    else:
        bad = None
    return {"config": "BASELINE configs[2]: Llama-2-7B W2/4A16 greedy decode, batch 1, whole layers sharded over the ranks",
            "n_gpus": world, "layers": layers, "tokens": tokens, "ctx": ctx, "start_position": start,
            "tokens_per_s": round(tokens / dt, 1), "ms_per_token": round(dt / tokens * 1e3, 3),
            "packed_weight_GB_per_token": round(nbytes.item() / 1e9, 3),
            "weight_stream_GBps": round(nbytes.item() / (dt / tokens) / 1e9, 1),
            "metadata_mode": "compact (fp16 zero-points)" if compact else "exact (fp32 zero-points)",
            "hipgraph": bool(graph), "launches_per_layer": stage_launches, "first_tokens": toks[:8], "token_ids": toks,
            "backend": backend if world > 1 else None,
            "per_rank": per_rank,
            "sum_of_stages_ms": None if per_rank is None else round(sum(r["stage_us_per_token"] for r in per_rank) / 1e3, 3),
            "sum_of_hops_ms": None if per_rank is None else round(
                sum((r["hop_round_trip_us_to_next_rank"] or 0.0) / 2 for r in per_rank) / 1e3
                + max((r["hop_round_trip_us_to_next_rank"] or 0.0) / 2 for r in per_rank) / 1e3, 3),
            "tokens_equal_single_process": equal, "first_mismatch": bad}
