"""Packed MXQ weights (format v1, csrc/mxq_format.h) and the ops on them.

Every function here is a thin host wrapper: validate like the reference's launchers
should have (SURVEY.md 8b "Errors"), allocate outputs with torch, pass raw device pointers
and torch's current HIP stream to libmxq_hip.so.  No arithmetic happens in Python and
there is no CPU path: tensors must live on a ROCm device.

Reference semantics implemented by the kernels:
  quantize_pack  MXQGPT.fasterquant       mxq_quant/lib/mxqgpt.py:387-448
  unpack         Quantizer.quantize codes mxq_quant/lib/quantizer.py:14-16
  dequant        Quantizer.dequantize     mxq_quant/lib/quantizer.py:19-20 (+ fp16 cast, mxqgpt.py:448)
  linear         nn.Linear on that weight mxq_quant/main.py:85 / gemv_mxq_cuda.cu:39-208
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib

_TORCH2CODE = {torch.float32: _lib.DTYPE_F32, torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16}

PARAM_KEYS = ("codes2", "sc2", "zero2", "qs2", "qz2", "codes4", "sc4", "zero4", "qs4", "qz4")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOCTX = _NoCtx()


def _on_device(device: torch.device):
    """Context that makes `device` current for a launch -- nothing at all when it already is (the usual case: one process
    per GPU; torch.cuda.device() costs ~4 us of host time per call, a third of a small-token Linear's)."""
    idx = device.index
    return _NOCTX if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(device)


_WORKSPACES: Dict[tuple, torch.Tensor] = {}
_WS_EAGER: Dict[tuple, bool] = {}    # key -> the counter head has been zeroed (eagerly, or by a memset recorded in its graph)
_WS_HEAD = 65536      # bytes of K-step counters at the head of a workspace (include/mxq_hip.h)
_WS_MAILBOX_OFF = 65504   # bytes: 64-bit address of the workspace's host mailbox (csrc/gemm8.hip SK_MAILBOX_OFF)
# A stream-K wait that expires (include/mxq_hip.h) writes the affected tile as NaN and flags the workspace.  The flag must reach
# the CALLER without anybody polling: every workspace owns 4 ints of pinned host memory ("mailbox") whose address sits in its
# head; the kernel stores its status there with system-scope stores, and every later hand-out of that workspace -- i.e. the next
# packing.linear / QuantLinear.forward call on it -- reads the mailbox (plain host memory, no synchronisation) and raises.  A
# starved launch therefore cannot feed NaN activations into a model for more than the launches already queued behind it.
# MXQ_CHECK_WORKSPACE=1 checks synchronously after EVERY call instead (debugging).
_MAILBOX: Dict[tuple, tuple] = {}     # key -> (pinned int32[4] tensor, its numpy view)
_CHECK_EVERY_CALL = __import__("os").environ.get("MXQ_CHECK_WORKSPACE", "") not in ("", "0")
MIDM_MAX_TOKENS = 256     # capi.hip: the most tokens mxq_linear_f16_ws may hand to the mid-M split-K kernel (no counters)


_CAPTURE_KEYS: list = []          # capture workspaces in order of creation
# A capture workspace lives in its graph's private memory pool; this cache holds a reference so that the memory cannot be
# handed to a later capture that shares the pool while the first graph still replays on it.  The price is that
# the bytes stay pinned after their graph is gone, so the cache is bounded -- generously: 256 graphs of <= 67 MB.
_MAX_CAPTURE_WS = 256


def _capture_id(stream_handle: int):
    """(capturing?, id of the capture sequence) of a stream, asked through libmxq_hip.so's own HIP runtime
    (include/mxq_hip.h: mxq_stream_capture_id -- never a second runtime opened by name)."""
    import ctypes
    active, cid = ctypes.c_int(0), ctypes.c_ulonglong(0)
    _lib.check(_lib.load().mxq_stream_capture_id(ctypes.c_void_p(stream_handle), ctypes.byref(active), ctypes.byref(cid)),
               "mxq_stream_capture_id")
    return bool(active.value), int(cid.value)


def gemm_workspace(device: torch.device, counters: bool = True, stream: Optional[int] = None,
                   need: Optional[int] = None) -> Optional[torch.Tensor]:
    """Scratch buffer of the stream-K prefill GEMM and of the mid-M kernel's partial tiles (include/mxq_hip.h:
    mxq_linear_f16_ws).  Its counter head must be zero when a stream-K launch starts; the kernels leave it zeroed.

    Eager launches: one buffer of the maximum size per (device, stream), created and zeroed on first use -- launches on
    one stream run in order and may share it, launches on different streams may not.

    Captured launches: buffers per CAPTURED GRAPH (keyed by the capture sequence's id), so two graphs replayed at the same
    time on two streams never share partial sums or counters.  ``need`` (bytes this launch can use,
    mxq_linear_workspace_need; None = the maximum) sizes them: a graph whose launches never leave the skinny kernel gets
    none at all (``need`` 0 returns None), one that stays within the small-tile builds a 34-MB buffer instead of 67 MB
    (two size classes per graph at most).  A buffer born under capture is never zeroed for real until a replay runs: its
    first counters hand-out records ONE memset of the 64-KiB counter head into the graph.

    ``counters=False``: the caller's kernel uses the buffer beyond the head only (the mid-M kernel's partial tiles):
    no memset is recorded for it."""
    if need is not None and need <= 0:
        return None
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    if stream is None:                      # (callers on the hot path pass the handle they already hold)
        stream = torch.cuda.current_stream(device).cuda_stream
    lib = _lib.load()
    full = lib.mxq_gemm_workspace_bytes()
    # (torch's own query looks at the CURRENT device's stream; only when it says "capturing" -- or the device is another
    #  one -- is the stream itself asked, through the library)
    capturing, cid = False, 0
    if dev_index != torch.cuda.current_device() or torch.cuda.is_current_stream_capturing():
        capturing, cid = _capture_id(stream)
    if capturing:
        nbytes = full if need is None or 2 * need > full else (full + _WS_HEAD) // 2     # two size classes
        nbytes = max(nbytes, need or 0)
        key = (dev_index, "capture", cid, nbytes)
    else:
        nbytes, key = full, (dev_index, stream)
    ws = _WORKSPACES.get(key)
    if ws is None and capturing:             # a bigger buffer of the same graph serves a smaller need
        ws = _WORKSPACES.get((dev_index, "capture", cid, full))
        if ws is not None:
            key = (dev_index, "capture", cid, full)
    if ws is None:
        ws = _WORKSPACES[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if not capturing:
            # (eager workspaces only: pinning host memory is not a capturable operation, and a captured graph's launches
            #  are replayed without passing through here -- its workspace is watched by workspace_status() alone)
            box = torch.zeros(4, dtype=torch.int32).pin_memory()
            _MAILBOX[key] = (box, box.numpy())
        if capturing:
            _WS_EAGER[key] = False                    # ... until a counters hand-out has recorded the memset
            _CAPTURE_KEYS.append(key)
            while len(_CAPTURE_KEYS) > _MAX_CAPTURE_WS:
                old = _CAPTURE_KEYS.pop(0)
                _WORKSPACES.pop(old, None)
                _WS_EAGER.pop(old, None)
                _MAILBOX.pop(old, None)
        else:
            _WS_EAGER[key] = True
            _zero_head(ws, key)
    else:
        box = _MAILBOX.get(key)
        if box is not None and box[1][0] != 0:        # a launch that ran on this workspace since the last hand-out gave up a wait
            _raise_expired(key, ws)
    if capturing and counters and not _WS_EAGER.get(key, False):
        _zero_head(ws, key)                            # recorded once per graph and buffer (memset + the mailbox's address)
        _WS_EAGER[key] = True
    return ws


def _zero_head(ws: torch.Tensor, key) -> None:
    """Zero a workspace's 64-KiB head and write the address of its host mailbox into it (stream-ordered; under capture both
    operations are recorded into the graph)."""
    ws[:_WS_HEAD].zero_()
    box = _MAILBOX.get(key)
    if box is not None:
        ws[_WS_MAILBOX_OFF:_WS_MAILBOX_OFF + 8].view(torch.int64).fill_(box[0].data_ptr())


def _raise_expired(key, ws: torch.Tensor):
    """The mailbox of ``ws`` is set: report, clear it, make the workspace usable again (synchronises), raise."""
    box, view = _MAILBOX[key]
    st = [int(v) for v in view]
    dev_index = key[0]
    torch.cuda.synchronize(dev_index)
    view[:] = 0
    with torch.cuda.device(dev_index):
        _zero_head(ws, key)
        torch.cuda.synchronize(dev_index)
    raise RuntimeError(f"stream-K wait expired on workspace {key}: [code, tail tile, workgroup, count seen] = {st}; the affected "
                       "output tiles of that launch are NaN (results computed since then on this stream are suspect); the "
                       "workspace has been reset")


_HOIST: Dict[tuple, torch.Tensor] = {}
HOIST_MIN_TOKENS = 4096     # mxq_hoist_min_tokens(): "auto" hoists the dequant out of the token loop from this many tokens on


def hoist_scratch(device: torch.device, nbytes: int) -> torch.Tensor:
    """Transient fp16-weight scratch of the hoisted-dequant mode (include/mxq_hip.h: mxq_linear_f16_hoisted), one per
    (device, stream), grown on demand.  Nothing is cached in it between calls: every call rewrites it first."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _HOIST.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _HOIST[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf


def _layout_code(p) -> int:
    if isinstance(p, PackedMXQ):
        return 3 if p.compact else 0
    return LAYOUTS[p.layout]


def linear_hoisted(x: torch.Tensor, p, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Quantised Linear with the dequant hoisted out of the token loop (dequant kernel once into a scratch buffer,
    then the MFMA kernel on fp16 tiles): bit-identical to the fused GEMM, faster from HOIST_MIN_TOKENS (4096) tokens per
    launch (profiles/r03_dense256.txt)."""
    _need_gpu(x, p.qweight)
    if x.dtype != torch.float16 or x.shape[-1] != p.K:
        raise ValueError("activations must be fp16 [..., in_features]")
    x2 = x.reshape(-1, p.K).contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, p.N), dtype=torch.float16, device=x.device)
    if M == 0:
        return out.reshape(*x.shape[:-1], p.N)
    lib = _lib.load()
    scratch = hoist_scratch(x2.device, lib.mxq_hoist_scratch_bytes(p.N, p.K))
    with _on_device(x.device):
        _lib.check(lib.mxq_linear_f16_hoisted(x2.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(),
                                              M, p.N, p.K, _layout_code(p), scratch.data_ptr(), scratch.numel(),
                                              _stream(x2)), "mxq_linear_f16_hoisted")
    return out.reshape(*x.shape[:-1], p.N)


def workspace_status(device: Optional[torch.device] = None, reset: bool = True) -> None:
    """Raise RuntimeError if a stream-K wait gave up on any cached workspace (all devices, or one) since the last check
    (include/mxq_hip.h: mxq_workspace_status -- the affected tiles were written as NaN).  Synchronises the workspace's
    stream; call it wherever the caller synchronises anyway (end of a prefill, a test, a bench's timed region).
    ``reset``: re-zero the counter head of a flagged workspace so that it can be used again."""
    import ctypes
    lib = _lib.load()
    bad = []
    for key, ws in list(_WORKSPACES.items()):
        dev_index = key[0]
        if device is not None and (device.index if device.index is not None else torch.cuda.current_device()) != dev_index:
            continue
        st = (ctypes.c_int * 4)()
        with torch.cuda.device(dev_index):
            torch.cuda.synchronize(dev_index)          # (capture workspaces have no eager stream of their own)
            _lib.check(lib.mxq_workspace_status(ws.data_ptr(), ws.numel(), ctypes.cast(st, ctypes.c_void_p), None),
                       "mxq_workspace_status")
            if st[0] != 0:
                bad.append((key, list(st)))
                if reset:
                    box = _MAILBOX.get(key)
                    if box is not None:
                        box[1][:] = 0
                    _zero_head(ws, key)
                    torch.cuda.synchronize(dev_index)
    if bad:
        raise RuntimeError("stream-K wait expired (workspace key, [code, tail tile, workgroup, count seen]): "
                           f"{bad}; the affected output tiles are NaN" + ("; counter heads re-zeroed" if reset else ""))


def reset_gemm_workspace(device: Optional[torch.device] = None):
    """Re-zero the counter heads of the cached stream-K workspaces (all devices, or one).  Needed only after a
    GEMM launch was aborted mid-kernel (device reset, killed process sharing the buffer): the kernels themselves
    always leave the counters zeroed.  Synchronises the device(s) involved."""
    for key, ws in list(_WORKSPACES.items()):
        dev_index = key[0]
        if device is None or (device.index if device.index is not None else torch.cuda.current_device()) == dev_index:
            torch.cuda.synchronize(dev_index)
            box = _MAILBOX.get(key)
            if box is not None:
                box[1][:] = 0
            with torch.cuda.device(dev_index):
                _zero_head(ws, key)
            torch.cuda.synchronize(dev_index)


def _need_gpu(*ts: torch.Tensor):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("mxq_amd ops run on the GPU only (no CPU fallback): got a CPU tensor")


def check_shape(N: int, K: int):
    if N <= 0 or K <= 0 or N % 16 != 0 or K % 64 != 0:
        raise ValueError(f"MXQ layout needs out_features % 16 == 0 and in_features % 64 == 0, got [{N}, {K}]")


def qweight_bytes(N: int, K: int) -> int:
    check_shape(N, K)
    return (N // 16) * (K // 64) * 576


@dataclass
class PackedMXQ:
    """qweight: int32 [N/16 * K/64 * 144] (576-B blocks, exact metadata) or [N/16 * K/64 * 120] (480-B blocks,
    ``compact``: fp16 zero-points, csrc/mxq_format.h MXQ_LAYOUT_MIXEDC); rowmeta: float32 [N, 4]."""
    qweight: torch.Tensor
    rowmeta: torch.Tensor
    N: int
    K: int
    compact: bool = False

    @property
    def device(self):
        return self.qweight.device

    def nbytes(self) -> int:
        return self.qweight.numel() * 4 + self.rowmeta.numel() * 4

    def bits_per_weight(self) -> float:
        return 8.0 * self.nbytes() / (self.N * self.K)


def concat_packed(ps) -> PackedMXQ:
    """Stack packed weights along the output dimension (e.g. q | k | v, or gate | up): format v1 is
    row-block-major, so this is a plain concatenation of the block arrays and of the row metadata, and one
    launch then computes all the stacked Linears of a shared input."""
    ps = list(ps)
    if not ps or any(p.K != ps[0].K or p.compact != ps[0].compact for p in ps):
        raise ValueError("concat_packed needs packed weights with the same in_features and metadata mode")
    return PackedMXQ(torch.cat([p.qweight for p in ps]), torch.cat([p.rowmeta for p in ps]),
                     sum(p.N for p in ps), ps[0].K, ps[0].compact)


def compact(p: PackedMXQ) -> PackedMXQ:
    """Exact -> compact metadata (format "v2": the 2-bit zero-points as fp16; 3.75 instead of 4.5 bit/weight).  The
    integer codes, scale codes and (qs, qz) are unchanged, so ``unpack`` stays bit-exact on them; the dequantised
    weight moves by at most the fp16 rounding of a zero-point times its scale, the GEMM result by ~4e-4 relative
    (inside the 1e-3 budget: tests).  ``rowmeta`` is shared with ``p``."""
    if p.compact:
        return p
    _need_gpu(p.qweight)
    lib = _lib.load()
    nbytes = lib.mxq_qweight_bytes_layout(p.N, p.K, 3)
    q = torch.empty(nbytes // 4, dtype=torch.int32, device=p.device)
    with _on_device(p.device):
        _lib.check(lib.mxq_compact(p.qweight.data_ptr(), q.data_ptr(), p.N, p.K, _stream(q)), "mxq_compact")
    return PackedMXQ(q, p.rowmeta, p.N, p.K, True)


def _alloc(N: int, K: int, device) -> PackedMXQ:
    lib = _lib.load()
    nbytes = lib.mxq_qweight_bytes(N, K)
    assert nbytes == qweight_bytes(N, K)
    return PackedMXQ(torch.empty(nbytes // 4, dtype=torch.int32, device=device),
                     torch.empty((N, 4), dtype=torch.float32, device=device), N, K)


def quantize_pack(W: torch.Tensor, dead: Optional[torch.Tensor] = None, compact_meta: bool = False) -> PackedMXQ:
    """MXQ-quantise W [N, K] (fp16 / bf16 / fp32) on device and return the packed form (``compact_meta``: in the
    compact metadata mode, see ``compact``)."""
    _need_gpu(W, dead)
    if W.dim() != 2:
        raise ValueError("weight must be 2-D [out_features, in_features]")
    if W.dtype not in _TORCH2CODE:
        raise ValueError(f"unsupported weight dtype {W.dtype}")
    N, K = W.shape
    check_shape(N, K)
    W = W.contiguous()
    dead_p = None
    if dead is not None:
        if dead.numel() != K:
            raise ValueError("dead-column mask must have in_features entries")
        dead = dead.to(device=W.device, dtype=torch.uint8).contiguous()
        dead_p = dead.data_ptr()
    lib = _lib.load()
    p = _alloc(N, K, W.device)
    with _on_device(W.device):
        _lib.check(lib.mxq_quantize_pack(W.data_ptr(), _TORCH2CODE[W.dtype], dead_p, p.qweight.data_ptr(),
                                         p.rowmeta.data_ptr(), N, K, _stream(W)), "mxq_quantize_pack")
    return compact(p) if compact_meta else p


def _param_shapes(N: int, K: int) -> Dict[str, tuple]:
    G = 3 * K // 64
    return dict(codes2=((N, 3 * K // 4), torch.uint8), sc2=((N, G), torch.uint8), zero2=((N, G), torch.float32),
                qs2=((N // 16, G), torch.float32), qz2=((N // 16, G), torch.float32),
                codes4=((N, K // 4), torch.uint8), sc4=((N,), torch.uint8), zero4=((N,), torch.float32),
                qs4=((N // 16,), torch.float32), qz4=((N // 16,), torch.float32))


def pack_codes(params: Dict[str, torch.Tensor], N: int, K: int) -> PackedMXQ:
    """Pack integer codes + parameters (the output of Quantizer.quantize / find_params)."""
    check_shape(N, K)
    shapes = _param_shapes(N, K)
    ts = []
    for k in PARAM_KEYS:
        t = params[k]
        _need_gpu(t)
        shape, dt = shapes[k]
        if tuple(t.shape) != shape or t.dtype != dt:
            raise ValueError(f"{k}: expected {shape} {dt}, got {tuple(t.shape)} {t.dtype}")
        ts.append(t.contiguous())
    lib = _lib.load()
    p = _alloc(N, K, ts[0].device)
    with _on_device(p.device):
        _lib.check(lib.mxq_pack_codes(*[t.data_ptr() for t in ts], p.qweight.data_ptr(), p.rowmeta.data_ptr(), N, K,
                                      _stream(p.qweight)), "mxq_pack_codes")
    return p


def unpack(p: PackedMXQ) -> Dict[str, torch.Tensor]:
    """Integer unpack: codes and parameters exactly as packed."""
    _need_gpu(p.qweight)
    out = {k: torch.empty(shape, dtype=dt, device=p.device) for k, (shape, dt) in _param_shapes(p.N, p.K).items()}
    lib = _lib.load()
    fn = lib.mxq_unpack_compact if p.compact else lib.mxq_unpack
    with _on_device(p.device):
        _lib.check(fn(p.qweight.data_ptr(), p.rowmeta.data_ptr(), *[out[k].data_ptr() for k in PARAM_KEYS],
                      p.N, p.K, _stream(p.qweight)), "mxq_unpack")
    return out


def dequant(p: PackedMXQ) -> torch.Tensor:
    """Dense fp16 [N, K] fake-quant weight (bit-identical to the reference's write-back)."""
    _need_gpu(p.qweight)
    out = torch.empty((p.N, p.K), dtype=torch.float16, device=p.device)
    lib = _lib.load()
    fn = lib.mxq_dequant_f16_compact if p.compact else lib.mxq_dequant_f16
    with _on_device(p.device):
        _lib.check(fn(p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), p.N, p.K, _stream(out)),
                   "mxq_dequant_f16")
    return out


DENSE_VARIANTS = {"auto": 0, "dense128": 1, "dense256": 2}   # include/mxq_hip.h: mxq_dense_f16 variants


def linear_dense(x: torch.Tensor, w16: torch.Tensor, out: Optional[torch.Tensor] = None, variant: str = "auto") -> torch.Tensor:
    """y = x @ w16.T on a dense fp16 [N, K] weight (e.g. ``dequant(p)``): the reference's ``nn.Linear`` on the fake-quant
    weight (mxq_quant/main.py:85) and the second half of the hoisted mode; fp32 accumulation, fp16 result.
    variant: "auto", "dense128" (256 x 128-tile kernel), "dense256" (256 x 256-tile kernel; K % 128 == 0)."""
    _need_gpu(x, w16)
    if x.dtype != torch.float16 or w16.dtype != torch.float16 or w16.dim() != 2 or x.shape[-1] != w16.shape[1]:
        raise ValueError("expected fp16 x [..., K] and fp16 w16 [N, K]")
    if not w16.is_contiguous():
        raise ValueError("w16 must be contiguous")
    if variant not in DENSE_VARIANTS:
        raise ValueError(f"unknown variant {variant!r}")
    N, K = w16.shape
    check_shape(N, K)
    x2 = x.reshape(-1, K).contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    elif out.shape != (M, N) or out.dtype != torch.float16 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float16 [tokens, out_features] tensor")
    if M:
        with _on_device(x.device):
            _lib.check(_lib.load().mxq_dense_f16(x2.data_ptr(), w16.data_ptr(), out.data_ptr(), M, N, K,
                                                 DENSE_VARIANTS[variant], _stream(x2)), f"mxq_dense_f16[{variant}]")
    return out.reshape(*x.shape[:-1], N)


GEMM_PATHS = {"gemm": 0, "fused": 0, "gemm1": 1, "gemm8": 8, "gemm9": 9, "midm": 10, "gemm8h": 12, "gemm8h_split": 13, "gemm8h_slices": 14, "gemm8q_split": 16, "gemm8q_slices": 17, "gemm8n_split": 20, "gemm8n_slices": 21}   # include/mxq_hip.h: enum mxq_gemm_variant


def linear(x: torch.Tensor, p: PackedMXQ, out: Optional[torch.Tensor] = None, path: str = "auto") -> torch.Tensor:
    """y = x @ dequant(p).T for x [..., K] fp16 -> [..., N] fp16 (fp32 accumulation).

    path: "auto" (the library's own dispatch, mxq_linear_f16_auto: GEMV kernel for <= 4 tokens, skinny MFMA kernel up to 40 --
    20 for weights beyond 24 M elements --, then by tile count the fused kernel's 64- / 128-token builds or the mid-M split-K
    kernel, the 256-token fused kernel beyond, the hoisted mode from 4096 tokens: profiles/r04_dispatch_map.txt), "gemm",
    "gemv", "midm",
    "skinny" (1..64 tokens), "hoist" (dequant hoisted out of the token loop; "auto" / "gemm" take it from
    HOIST_MIN_TOKENS tokens on), "fused" (never hoist), "whole" (fused kernel, whole tiles only), or an explicit GEMM
    kernel: "gemm1" (128x128 tile), "gemm8" (256x128 tile, wave-specialised, persistent, stream-K tail), "gemm9"
    (gemm8 splitting its tail whenever that is structurally possible: tests), "gemm8h" / "gemm8h_split" / "gemm8h_slices"
    (its 128-token build: tail split where it pays / always / K slices + combine launch), "gemm8q_split" / "gemm8q_slices"
    (the 64-token build)."""
    _need_gpu(x, p.qweight)
    if x.dtype != torch.float16:
        raise ValueError(f"activations must be float16 (W2/4 x A16), got {x.dtype}")
    if x.shape[-1] != p.K:
        raise ValueError(f"in_features mismatch: x has {x.shape[-1]}, weight has {p.K}")
    if x.device != p.device:
        raise ValueError("x and the packed weight live on different devices")
    flat = x.dim() == 2 and x.is_contiguous()          # (the usual call: no view objects made on the way in or out)
    x2 = x if flat else x.reshape(-1, p.K).contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, p.N), dtype=torch.float16, device=x.device)
    elif out.shape != (M, p.N) or out.dtype != torch.float16 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float16 [tokens, out_features] tensor")
    if M == 0:
        return out.reshape(*x.shape[:-1], p.N)
    if path == "hoist":
        return linear_hoisted(x, p, out=out)
    lib = _lib.load()
    args = (x2.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, p.N, p.K)
    if path == "auto" and M > 4:
        # ONE C call for the whole dispatch (include/mxq_hip.h: mxq_linear_f16_auto): skinny / mid-M split-K / fused
        # prefill kernel with its stream-K tail / hoisted-dequant mode, chosen inside the library by token count
        hoists = M >= HOIST_MIN_TOKENS
        st = _stream(x2)
        lay = _layout_code(p)
        with _on_device(x.device):
            # (a workspace only where the dispatch can use one, and -- under capture -- only as big as this launch needs)
            ws = gemm_workspace(x2.device, stream=st, need=lib.mxq_linear_workspace_need(M, p.N, p.K, lay, 1))
            scratch = hoist_scratch(x2.device, lib.mxq_hoist_scratch_bytes(p.N, p.K)) if hoists else None
            _lib.check(lib.mxq_linear_f16_auto(*args, lay, ws.data_ptr() if ws is not None else None,
                                               ws.numel() if ws is not None else 0,
                                               scratch.data_ptr() if scratch is not None else None,
                                               scratch.numel() if scratch is not None else 0, st),
                       "mxq_linear_f16_auto")
            if _CHECK_EVERY_CALL and ws is not None:
                workspace_status(x2.device)
        return out if flat else out.reshape(*x.shape[:-1], p.N)
    if path == "gemm" and M >= HOIST_MIN_TOKENS:
        return linear_hoisted(x, p, out=out)
    if path == "skinny":
        with _on_device(x.device):
            _lib.check(lib.mxq_skinny_f16(*args, 3 if p.compact else 0, _stream(x2)), "mxq_skinny_f16")
        return out.reshape(*x.shape[:-1], p.N)
    if path == "whole":   # the fused prefill kernel on whole tiles only (no workspace: no stream-K split, so a tile's sums
        #                   do not depend on the launch's token count: tests compare launches of different sizes bit for bit)
        with _on_device(x.device):
            _lib.check(lib.mxq_gemm_f16_layout(*args, _layout_code(p), _stream(x2)), "mxq_gemm_f16_layout[whole]")
        return out.reshape(*x.shape[:-1], p.N)
    if p.compact:      # compact metadata: the layout entry points (same kernels, other field offsets)
        if path not in ("auto", "gemm", "gemv", "gemm8", "fused"):
            raise ValueError(f"path {path!r} is not available for compact metadata")
        with _on_device(x.device):
            fn = lib.mxq_gemv_f16_layout if path == "gemv" or (path == "auto" and M <= 4) else lib.mxq_gemm_f16_layout
            _lib.check(fn(*args, 3, _stream(x2)), f"mxq_linear_f16[{path}, compact]")
        return out.reshape(*x.shape[:-1], p.N)
    with _on_device(x.device):
        if path == "gemv":
            rc = lib.mxq_gemv_f16(*args, _stream(x2))
        elif path == "auto" and M <= 4:                          # GEMV: no workspace involved
            rc = lib.mxq_linear_f16(*args, _stream(x2))
        else:
            midm = path == "midm"
            ws = (gemm_workspace(x2.device, counters=not midm)
                  if path in ("auto", "gemm", "fused", "gemm8", "gemm9", "midm", "gemm8h", "gemm8h_split", "gemm8h_slices", "gemm8q_split", "gemm8q_slices", "gemm8n_split", "gemm8n_slices") else None)
            wsp, wsn = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
            if path == "auto":
                rc = lib.mxq_linear_f16_ws(*args, wsp, wsn, _stream(x2))
            elif path in GEMM_PATHS:
                rc = lib.mxq_gemm_f16_ws(*args, GEMM_PATHS[path], wsp, wsn, _stream(x2))
            else:
                raise ValueError(f"unknown path {path!r}")
        _lib.check(rc, f"mxq_linear_f16[{path}]")
        if _CHECK_EVERY_CALL and path not in ("gemv",) and M > 4:
            workspace_status(x2.device)
    return out.reshape(*x.shape[:-1], p.N)


def linear_fused(x: torch.Tensor, p: PackedMXQ, prologue: int = 0, norm_w: Optional[torch.Tensor] = None,
                 eps: float = 1e-5, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One-token quantised Linear with the neighbouring decoder-layer ops fused in (decode harness):
    prologue 0 none, 1 RMSNorm(x) * norm_w, 2 SwiGLU gate (x is [1, 2K] = gate | up);
    ``residual`` [1, N] is added to the result.  fp16 in / out, GEMV kernel."""
    _need_gpu(x, p.qweight, norm_w, residual)
    k_in = 2 * p.K if prologue == 2 else p.K
    if x.dtype != torch.float16 or x.numel() != k_in:
        raise ValueError(f"expected one fp16 token with {k_in} features")
    if prologue == 1 and (norm_w is None or norm_w.dtype != torch.float16 or norm_w.numel() != p.K):
        raise ValueError("RMSNorm prologue needs an fp16 weight of in_features elements")
    if residual is not None and (residual.dtype != torch.float16 or residual.numel() != p.N):
        raise ValueError("residual must be fp16 [1, out_features]")
    x = x.contiguous()
    out = torch.empty((1, p.N), dtype=torch.float16, device=x.device)
    lib = _lib.load()
    fused = lib.mxq_gemv_fused_f16_compact if p.compact else lib.mxq_gemv_fused_f16
    with _on_device(x.device):
        _lib.check(fused(x.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(),
                         p.N, p.K, int(prologue), norm_w.data_ptr() if norm_w is not None else None,
                         float(eps), residual.contiguous().data_ptr() if residual is not None else None,
                         _stream(x)), "mxq_gemv_fused_f16")
    return out


def linear_swiglu(x: torch.Tensor, gate_up: PackedMXQ, norm_w: torch.Tensor, eps: float = 1e-5):
    """One token through RMSNorm -> gate | up -> SwiGLU in ONE launch (include/mxq_hip.h: mxq_gemv_swiglu_f16): ``gate_up`` is
    gate stacked on up ([2 I, K], ``concat_packed``).  Returns (act fp16 [1, I] in the kernels' staged order, act_sum f32
    [I / 16]) -- the input of ``linear_staged``; bit for bit what ``linear_fused(.., 1, norm_w)`` followed by the SwiGLU
    staging of ``linear_fused(.., 2)`` computes."""
    _need_gpu(x, gate_up.qweight, norm_w)
    if x.dtype != torch.float16 or x.numel() != gate_up.K or norm_w.dtype != torch.float16 or norm_w.numel() != gate_up.K:
        raise ValueError("expected one fp16 token and an fp16 RMSNorm weight of in_features elements")
    if gate_up.N % 32 != 0 or gate_up.K % 256 != 0:
        raise ValueError("the fused SwiGLU launch needs out_features % 32 == 0 (gate | up) and in_features % 256 == 0")
    inter = gate_up.N // 2
    x = x.contiguous()
    act = torch.empty((1, inter), dtype=torch.float16, device=x.device)
    act_sum = torch.empty(inter // 16, dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        _lib.check(_lib.load().mxq_gemv_swiglu_f16(x.data_ptr(), gate_up.qweight.data_ptr(), gate_up.rowmeta.data_ptr(),
                                                   act.data_ptr(), act_sum.data_ptr(), gate_up.N, gate_up.K, norm_w.data_ptr(),
                                                   float(eps), int(gate_up.compact), _stream(x)), "mxq_gemv_swiglu_f16")
    return act, act_sum


def linear_staged(act: torch.Tensor, act_sum: torch.Tensor, p: PackedMXQ, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = residual + W . act for a staged row from ``linear_swiglu`` (mxq_gemv_staged_f16)."""
    _need_gpu(act, act_sum, p.qweight, residual)
    if act.dtype != torch.float16 or act.numel() != p.K or act_sum.dtype != torch.float32 or act_sum.numel() != p.K // 16:
        raise ValueError("expected a staged fp16 row of in_features elements and its in_features / 16 fp32 group sums")
    if residual is not None and (residual.dtype != torch.float16 or residual.numel() != p.N):
        raise ValueError("residual must be fp16 [1, out_features]")
    out = torch.empty((1, p.N), dtype=torch.float16, device=act.device)
    with _on_device(act.device):
        _lib.check(_lib.load().mxq_gemv_staged_f16(act.data_ptr(), act_sum.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(),
                                                   out.data_ptr(), p.N, p.K,
                                                   residual.contiguous().data_ptr() if residual is not None else None,
                                                   int(p.compact), _stream(act)), "mxq_gemv_staged_f16")
    return out


# ------------------------------------------------------------------------------------------------
# uniform layouts of the BASELINE config-5 sweep (W2A16 / W4A16 arms next to the mixed layout)
# ------------------------------------------------------------------------------------------------
LAYOUTS = {"mixed": 0, "w2g16": 1, "w4row": 2}


@dataclass
class PackedUniform:
    """W2G16 / W4ROW packed weight (csrc/mxq_format.h); same [N/16][K/64] block grid as PackedMXQ."""
    qweight: torch.Tensor
    rowmeta: torch.Tensor
    N: int
    K: int
    layout: str

    def nbytes(self) -> int:
        return self.qweight.numel() * 4 + self.rowmeta.numel() * 4

    def bits_per_weight(self) -> float:
        return 8.0 * self.nbytes() / (self.N * self.K)


def quantize_pack_uniform(W: torch.Tensor, layout: str) -> PackedUniform:
    """Quantise W [N, K] with ``Quantizer(bits=2, group 16)`` ("w2g16") or ``Quantizer(bits=4, per row)``
    ("w4row"), both with the 4-bit second-order scale coding (reference lib/quantizer.py:61-147)."""
    _need_gpu(W)
    if layout not in ("w2g16", "w4row"):
        raise ValueError("layout must be 'w2g16' or 'w4row'")
    if W.dim() != 2 or W.dtype not in _TORCH2CODE:
        raise ValueError("weight must be a 2-D fp16 / bf16 / fp32 tensor")
    N, K = W.shape
    check_shape(N, K)
    W = W.contiguous()
    lib = _lib.load()
    nbytes = lib.mxq_qweight_bytes_layout(N, K, LAYOUTS[layout])
    p = PackedUniform(torch.empty(nbytes // 4, dtype=torch.int32, device=W.device),
                      torch.empty((N, 4), dtype=torch.float32, device=W.device), N, K, layout)
    with _on_device(W.device):
        _lib.check(lib.mxq_quantize_pack_layout(W.data_ptr(), _TORCH2CODE[W.dtype], p.qweight.data_ptr(),
                                                p.rowmeta.data_ptr(), N, K, LAYOUTS[layout], _stream(W)),
                   "mxq_quantize_pack_layout")
    return p


def expand_uniform(p: PackedUniform, codes: bool = True):
    """(fp16 [N, K] dequantised weight, dict of integer codes / parameters or None)."""
    N, K, dev = p.N, p.K, p.qweight.device
    w16 = torch.empty((N, K), dtype=torch.float16, device=dev)
    out = None
    ptrs = [None] * 5
    if codes:
        G = K // 16 if p.layout == "w2g16" else 1
        out = dict(codes=torch.empty((N, K), dtype=torch.uint8, device=dev),
                   sc=torch.empty((N, G), dtype=torch.uint8, device=dev),
                   zero=torch.empty((N, G), dtype=torch.float32, device=dev),
                   qs=torch.empty((N // 16, G), dtype=torch.float32, device=dev),
                   qz=torch.empty((N // 16, G), dtype=torch.float32, device=dev))
        ptrs = [out[k].data_ptr() for k in ("codes", "sc", "zero", "qs", "qz")]
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.mxq_expand_layout(p.qweight.data_ptr(), p.rowmeta.data_ptr(), w16.data_ptr(), *ptrs, N, K,
                                         LAYOUTS[p.layout], _stream(w16)), "mxq_expand_layout")
    return w16, out


def linear_layout(x: torch.Tensor, p, out: Optional[torch.Tensor] = None, path: str = "gemm") -> torch.Tensor:
    """Quantised Linear on a PackedMXQ (mixed) or PackedUniform weight: path "gemm" = MFMA dequant-GEMM at any
    token count, "gemv" = the streaming GEMV (<= 4 tokens), "skinny" = the skinny MFMA kernel (<= 64 tokens), "auto" = the
    library's dispatch (mxq_linear_f16_auto: GEMV up to 4 tokens, skinny kernel up to 48, fused GEMM, hoisted mode from
    HOIST_MIN_TOKENS)."""
    layout = LAYOUTS[getattr(p, "layout", "mixed")]
    if layout == 0:
        return linear(x, p, out=out, path=path)        # the mixed layout's default (fastest) kernels
    _need_gpu(x, p.qweight)
    if x.dtype != torch.float16 or x.shape[-1] != p.K:
        raise ValueError("activations must be fp16 [..., in_features]")
    x2 = x.reshape(-1, p.K).contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, p.N), dtype=torch.float16, device=x.device)
    if path == "whole":
        path = "fused"                                 # (uniform layouts: "fused" already runs without a workspace)
    if path not in ("gemm", "gemv", "auto", "hoist", "fused", "skinny"):
        raise ValueError(f"unknown path {path!r}")
    if M == 0:
        return out.reshape(*x.shape[:-1], p.N)
    if path == "hoist" or (path == "gemm" and M >= HOIST_MIN_TOKENS):
        return linear_hoisted(x, p, out=out)
    lib = _lib.load()
    args = (x2.data_ptr(), p.qweight.data_ptr(), p.rowmeta.data_ptr(), out.data_ptr(), M, p.N, p.K, layout)
    with _on_device(x.device):
        if path == "auto" and M > 4:        # the library's own dispatch: skinny kernel / fused prefill GEMM / hoisted mode
            hoists = M >= HOIST_MIN_TOKENS
            ws = gemm_workspace(x2.device, need=lib.mxq_linear_workspace_need(M, p.N, p.K, layout, 1))
            scratch = hoist_scratch(x2.device, lib.mxq_hoist_scratch_bytes(p.N, p.K)) if hoists else None
            _lib.check(lib.mxq_linear_f16_auto(*args, ws.data_ptr() if ws is not None else None,
                                               ws.numel() if ws is not None else 0,
                                               scratch.data_ptr() if scratch is not None else None,
                                               scratch.numel() if scratch is not None else 0, _stream(x2)),
                       "mxq_linear_f16_auto")
        elif path == "skinny":
            _lib.check(lib.mxq_skinny_f16(*args, _stream(x2)), "mxq_skinny_f16")
        else:
            fn, what = ((lib.mxq_gemv_f16_layout, "mxq_gemv_f16_layout") if path in ("gemv", "auto")
                        else (lib.mxq_gemm_f16_layout, "mxq_gemm_f16_layout"))
            _lib.check(fn(*args, _stream(x2)), what)
    return out.reshape(*x.shape[:-1], p.N)
