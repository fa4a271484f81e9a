#!/usr/bin/env python3
"""Headline benchmark: quantized-Linear TFLOP/s (+ tokens/s) of the MXQ W2/4 x A16 dequant-GEMM
on Llama-2-7B-shaped synthetic weights (BASELINE.json: metric; workload = configs[1]:
"Llama-2-7B all-Linear W2/4A16, batch=1 seq=2048, 1 x MI355X HIP dequant-GEMM kernel").

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of all 224 quantized Linears (32 layers x q,k,v,o,gate,up,down) over
one sequence of 2048 tokens = 26.53 TFLOP, inputs and packed weights resident in HBM; every launch goes
through the product entry (packing.linear(path="auto") -> mxq_linear_f16_auto, what QuantLinear.forward
calls; --path picks one schedule for A/B runs and says so in config.entry).
N > 1: whole decoder layers are sharded over the ranks (rank r owns layers [32r/N, 32(r+1)/N),
SURVEY.md 8e) and N sequences flow through the layer pipeline per step, the [2048, 4096] fp16
hidden state hopping rank -> rank+1 by RCCL send/recv: per-GPU work is fixed ("weak").  N > 1 runs are
contained (mxq_amd/pipeline.py: Watchdog, init_group, run_guarded): 120-s process-group timeout, a deadline per
phase and a hard one below the driver's limit -- a stuck rank prints its stacks and exits 86, a failed rank exits
non-zero at once with a rank-tagged traceback -- and rank 0's line carries every rank's own account of the timed
region (`per_rank`: its GEMMs' device time, the time its stream waited for the hop in / for a send slot, bytes
hopped, stream-K status, shader clock) and, in `decode_pipeline.per_rank`, each stage's time per token and the
round trip of the hidden row to the next rank.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (mxq_gemm8_f16_kernel,
MFMA-bound): algorithmic 2*M*N*K flops of the launches of one step / their device time measured
with HIP events on the launch stream; `roofline.traffic` = HBM-side bytes per launch from rocprofv3
PMC child passes made in this run (live_traffic).  The other BASELINE configs ride along as side
figures with their own roofline fractions: `decode_1gpu` (configs[2]), `fakequant_block` (configs[3]),
`config5` (configs[4]); `--figure NAME` runs one of them alone.  `cpu_baseline` times the CPU restatement of the
reference's dequant + F.linear (oracle/cpu_linear.py) on a bounded sample, rank 0 / N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
# (NCCL_DEBUG is left alone: RCCL writes its log to STDOUT, which must stay the one JSON line; torch's own error text and the
#  watchdog's stack dumps go to stderr)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mxq_amd import llama_shapes as LS  # noqa: E402
from mxq_amd import packing  # noqa: E402
from mxq_amd import pipeline  # noqa: E402
from mxq_amd.pipeline import LayerPipeline, rank_census  # noqa: E402

SEQ = 2048
PEAK_F16_TFLOPS = 2500.0     # MI355X dense fp16 MFMA (MI355X_MICROARCH.md, chip-level parameters)


def build_layers(layers, dev):
    """Synthetic MXQ-quantised weights: randn * 0.02 -> fp16 -> fused HIP quantise-and-pack.
    Seeds as SURVEY.md 8d: 1000 * layer + linear index."""
    out = []
    for li in layers:
        lin = []
        for idx, (name, N, K) in enumerate(LS.LAYER_LINEARS):
            g = torch.Generator(device=dev).manual_seed(1000 * li + idx)
            W = (torch.randn(N, K, generator=g, device=dev, dtype=torch.float32) * 0.02).half()
            lin.append((name, packing.quantize_pack(W)))
            del W
        out.append(lin)
    return out


def host_cores():
    """Cores this process may really use: its affinity mask, cut by the cgroup CPU quota where one is set (the 1-GPU
    box gives the job a share of a 256-CPU host: the mask says 256, the quota what it gets), capped at 16."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // period)
        except (OSError, ValueError):
            pass
    return cores, quota, max(1, min(16, cores, quota or cores))


def cpu_baseline(dev, budget_s=12.0):
    """Config 1 of BASELINE.json on the host cores: one [4096, 4096] MXQ Linear, batch 1 x
    seq 128 -> dequant (fp32 scale*(q-zero), fp16 cast) + F.linear per call; and, next to it, the GPU's own time
    for the same call on the same inputs ("the CPU number next to the GPU kernel ... with
    the speed-up"), measured with HIP events after the headline region."""
    from oracle import cpu_linear
    cores, quota, threads = host_cores()
    torch.set_num_threads(threads)
    N = K = 4096
    M = 128
    g = torch.Generator(device=dev).manual_seed(0)
    W = (torch.randn(N, K, generator=g, device=dev) * 0.02).half()
    p = packing.quantize_pack(W)
    params = {k: v.cpu() for k, v in packing.unpack(p).items()}
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(7)).half()
    xd = x.to(dev)
    y_gpu = packing.linear(xd, p).float().cpu()               # the product dispatch for 128 tokens
    for _ in range(2):
        y = cpu_linear.dequant_linear(params, N, K, x)
    rel = ((y - y_gpu).abs().max() / y.abs().max()).item()
    # GPU: 200 stream-ordered calls between two HIP events on the launch stream (weights re-read every call; 9.4 MB
    # stay in the Infinity Cache between calls, as they would between the tokens of a real prefill chunk)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(20):
        packing.linear(xd, p, out=out)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        packing.linear(xd, p, out=out)
    ev1.record()
    torch.cuda.synchronize()
    gpu_ms = ev0.elapsed_time(ev1) / reps
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < 200:
        t0 = time.perf_counter()
        cpu_linear.dequant_linear(params, N, K, x)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    flop = 2.0 * M * N * K
    return {"value": round(flop / med / 1e12, 5), "unit": "TFLOP/s", "cores": torch.get_num_threads(),
            "affinity_cpus": cores, "cgroup_cpu_quota": quota,
            "kind": "port", "tokens_per_s": round(M / med, 1), "ms_per_call": round(med * 1e3, 3),
            "iters": len(times), "host_cpus": os.cpu_count(), "gpu_vs_cpu_max_rel_err": rel,
            "gpu_ms_per_call": round(gpu_ms, 5), "gpu_TFLOPs": round(flop / (gpu_ms * 1e-3) / 1e12, 2),
            "gpu_speedup": round(med * 1e3 / gpu_ms, 1),
            "gpu_kernel": "fused kernel's 128 x 64-tile build in slices mode (64 tiles x 4 K-slices + combine launch: the "
                          "mxq_linear_f16_auto dispatch at 128 tokens), stream-ordered launches, HIP events",
            "sample": "config 1: one 4096x4096 MXQ Linear, M=128 tokens, dequant(fp32)+F.linear per call "
                      "(4.295 GFLOP), median of the calls that fit ~12 s; the GPU runs the same call on the same inputs"}


def live_traffic(budget_s=150.0):
    """HBM-side bytes per launch of the headline kernel, MEASURED in this run (the figure used to be an
    offline constant): rocprofv3 child processes -- `--kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE
    passes, as MI355X_MICROARCH.md's HBM section prescribes -- around tools/gemm_prof.py at the three Linear shapes of the
    step; bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (the gfx950 correction: FETCH_SIZE reports half the bytes of wide
    reads), weighted by the shapes' launches per step.  Returns (avg bytes per launch, detail) or None when rocprofv3 is
    not there, a pass fails or the budget runs out (the caller then falls back to the committed constant)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    # already under a profiler (rocprofv3 -- python3 bench.py): no nested profiler runs
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    shapes = [(LS.HIDDEN, LS.HIDDEN, 128), (LS.INTERMEDIATE, LS.HIDDEN, 64), (LS.HIDDEN, LS.INTERMEDIATE, 32)]   # N, K, launches per step
    tmp = tempfile.mkdtemp(prefix="mxq_pmc_", dir="/tmp")
    t_end = time.perf_counter() + budget_s
    detail, tot, tot_n = {}, 0.0, 0
    try:
        for N, K, n in shapes:
            vals = {}
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                left = t_end - time.perf_counter()
                if left < 5:
                    return None
                d = os.path.join(tmp, f"{c}_{N}x{K}")
                # (the program itself follows "--": no shell, no env wrapper -- the profiler's library has initialised the GPU by then)
                r = subprocess.run([exe, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable,
                                    os.path.join(ROOT, "tools", "gemm_prof.py"), "gemm", str(SEQ), str(N), str(K), "6"],
                                   cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL, timeout=left)
                if r.returncode != 0:
                    return None
                rows = []
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    rows += [float(x["Counter_Value"]) for x in csv.DictReader(open(f))
                             if "gemm8" in x["Kernel_Name"] and x["Counter_Name"] == c]
                if not rows:
                    return None
                tail = rows[len(rows) // 2:]            # drop the warm-up half
                vals[c] = sum(tail) / len(tail)
            hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
            alg = (N // 16) * (K // 64) * 576 + N * 16 + 2 * SEQ * K + 2 * SEQ * N
            detail[f"{SEQ}x{N}x{K}"] = {"hbm_bytes": hbm, "algorithmic_bytes": float(alg), "ratio": round(hbm / alg, 2)}
            tot += hbm * n
            tot_n += n
        return tot / tot_n, detail
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


PEAK_HBM_GBPS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming kernel reaches)


def _events_ms(fn, iters):
    """ms per call of `fn` over `iters` back-to-back calls between two HIP events on the launch stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def launch_floor(dev, launches=160, reps=5):
    """Fixed cost of ONE dependent decode launch, in microseconds: a hipGraph of `launches` GEMV launches, each fed by the one
    before it (RMSNorm prologue + residual epilogue as in the decode layer), on a weight of 4096 x 64 -- 147 KB, i.e. next to no
    bytes.  What is left is what every launch of the token loop pays before and after its bytes: the gap to the previous
    kernel, the first loads' round trip, activation staging, the final reduction and store.  A layer of 5 launches cannot be
    shorter than 5 x this, whatever the HBM does."""
    g = torch.Generator(device=dev).manual_seed(2)
    p = packing.quantize_pack((torch.randn(LS.HIDDEN, 64, generator=g, device=dev) * 0.02).half())
    norm_w = torch.ones(64, device=dev, dtype=torch.float16)
    h0 = torch.randn(1, LS.HIDDEN, generator=g, device=dev).half()

    def chain():
        h = h0
        for _ in range(launches):
            h = packing.linear_fused(h[:, :64], p, 1, norm_w, residual=h)
        return h
    chain()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chain()
    gr.replay()
    torch.cuda.synchronize()
    return round(min(_events_ms(gr.replay, 3) for _ in range(reps)) * 1e3 / launches, 3)


def decode_1gpu(dev, tokens=64, ctx=512):
    """Side figure, not `value`: BASELINE configs[2] on ONE GPU -- greedy decode, batch 1, all 32 layers, hipGraph token
    loop (mxq_amd/llama_decode.py), in both metadata modes.  Bytes = the packed weights a token streams (codes + group
    metadata + row metadata of the 224 Linears; lm_head, KV cache and activations not counted): the HBM roofline's
    algorithmic bytes per token.  The dominant kernels are the streaming GEMVs (csrc/gemv.hip).
    """
    from mxq_amd.llama_decode import decode_pipeline_figure
    floor_us = launch_floor(dev)
    out = {"workload": f"BASELINE configs[2] on one GPU: Llama-2-7B W2/4A16 greedy decode, batch 1, {tokens} tokens from position 0 "
                       f"(KV cache of {ctx}), hipGraph token loop; bytes = packed weight bytes per token",
           "peak_GBps": PEAK_HBM_GBPS,
           # the latency floor next to every `frac` below: a dependent launch that moves no bytes
           "fixed_cost_us_per_launch": floor_us, "launches_per_token": 5 * LS.N_LAYERS + 4,
           "latency_floor_ms_per_token": round(floor_us * (5 * LS.N_LAYERS + 4) / 1e3, 4)}
    for mode, compact in (("exact", False), ("compact", True)):
        fig = decode_pipeline_figure(LayerPipeline(0, 1), dev, tokens=tokens, ctx=ctx, compact=compact)
        out[mode] = {"tokens_per_s": fig["tokens_per_s"], "ms_per_token": fig["ms_per_token"],
                     "packed_weight_GB_per_token": fig["packed_weight_GB_per_token"],
                     "GBps": fig["weight_stream_GBps"], "frac": round(fig["weight_stream_GBps"] / PEAK_HBM_GBPS, 4),
                     "launches_per_layer": fig.get("launches_per_layer"), "us_per_layer": round(fig["ms_per_token"] * 1e3 / LS.N_LAYERS, 2),
                     # bytes over the time that is NOT the launches' fixed cost: what the streaming itself achieves
                     "GBps_beyond_the_latency_floor": round(fig["packed_weight_GB_per_token"] /
                                                            max(1e-9, fig["ms_per_token"] - out["latency_floor_ms_per_token"]) * 1e3, 1),
                     "first_tokens": fig["first_tokens"]}
        if not compact:      # the same at a LONG context: 64 tokens from position 1920 of a 2048-row cache (split attention launch)
            fig = decode_pipeline_figure(LayerPipeline(0, 1), dev, tokens=tokens, ctx=2048, start=1920, compact=False)
            out["exact_at_position_1920"] = {"tokens_per_s": fig["tokens_per_s"], "ms_per_token": fig["ms_per_token"], "kv_cache_rows": 2048,
                                             "GBps": fig["weight_stream_GBps"], "frac": round(fig["weight_stream_GBps"] / PEAK_HBM_GBPS, 4)}
        # (no torch.cuda.empty_cache() between figures: weights allocated into memory that was just handed back to the driver
        #  decode 3 % slower -- 700 vs 723 tokens/s on one box, tools/_variants/order_test.py in round 5 -- presumably smaller
        #  physically contiguous fragments behind the same virtual range; the box has 288 GB, nothing needs to be returned)
    return out


def fakequant_block(dev, iters=10):
    """Side figure, not `value`: BASELINE configs[3] -- MXAsymQuantizer forward and STE backward (csrc/fakequant.hip) over
    the 7 weights of one decoder block in bf16, one launch per weight, HIP events around `iters` passes over the block.
    Algorithmic bytes: forward reads + writes every element (4 B), backward reads grad and weight and writes grad (6 B)."""
    from mxq_amd.utils_quant import mx_fake_quant, ste_clip_backward
    g = torch.Generator(device=dev).manual_seed(3)
    ws = [(torch.randn(N, K, generator=g, device=dev) * 0.02).bfloat16() for _n, N, K in LS.LAYER_LINEARS]
    gs = [torch.randn(w.shape, generator=g, device=dev).bfloat16() for w in ws]
    n = sum(w.numel() for w in ws)

    def fwd():
        for w in ws:
            mx_fake_quant(w, 2)

    def bwd():
        for go, w in zip(gs, ws):
            ste_clip_backward(go, w, -2.0, 2.0)
    fwd(); bwd()
    torch.cuda.synchronize()
    f_us, b_us = _events_ms(fwd, iters) * 1e3, _events_ms(bwd, iters) * 1e3
    return {"workload": "BASELINE configs[3]: MXAsymQuantizer fwd + STE bwd over one decoder block's 7 weights "
                        f"({n} elements), bf16, w_bits 2, one launch per weight",
            "fwd_us": round(f_us, 1), "bwd_us": round(b_us, 1),
            "fwd_GBps": round(4.0 * n / f_us / 1e3, 1), "bwd_GBps": round(6.0 * n / b_us / 1e3, 1),
            "fwd_frac": round(4.0 * n / f_us / 1e3 / PEAK_HBM_GBPS, 4), "bwd_frac": round(6.0 * n / b_us / 1e3 / PEAK_HBM_GBPS, 4),
            "peak_GBps": PEAK_HBM_GBPS, "kernels": "mxq_fakequant_fwd_* / mxq_fakequant_bwd_kernel"}


def config5(dev, tokens=32768, iters=3):
    """Side figure, not `value`: BASELINE configs[4] -- uniform W2 (group 16), uniform W4 (per row) and the mixed 2/4 layout
    on one decoder layer's 7 Linears.  Prefill leg: batch 8 x seq 4096 = `tokens` tokens per launch through the product
    dispatch (hoisted-dequant mode at this size: dequant pass + csrc/dense256.hip inside every timed launch) -> TFLOP/s and
    fraction of the fp16 MFMA peak.  Decode leg: ONE token through the same 7 Linears (streaming GEMV), replayed from a
    hipGraph over >= 600 MB of distinct copies of the layer so that every launch streams from HBM -> GB/s of packed bytes and
    fraction of the HBM peak."""
    g = torch.Generator(device=dev).manual_seed(5)
    Ws = [(torch.randn(N, K, generator=g, device=dev) * 0.02).half() for _n, N, K in LS.LAYER_LINEARS]
    xs = {K: torch.randn(tokens, K, generator=g, device=dev).half() for K in (LS.HIDDEN, LS.INTERMEDIATE)}
    ys = {N: torch.empty(tokens, N, device=dev, dtype=torch.float16) for N in (LS.HIDDEN, LS.INTERMEDIATE)}
    x1 = {K: xs[K][:1].contiguous() for K in xs}
    y1 = {N: torch.empty(1, N, device=dev, dtype=torch.float16) for N in ys}
    fl = 2.0 * tokens * LS.PARAMS_PER_LAYER
    out = {"workload": f"BASELINE configs[4]: W2A16 (group 16) vs W4A16 (per row) vs mixed 2/4 on one decoder layer's 7 Linears; "
                       f"prefill leg {tokens} tokens per launch, decode leg 1 token per launch", "arms": {}}
    for arm in ("w2g16", "w4row", "mixed"):
        lin = [packing.quantize_pack(W) if arm == "mixed" else packing.quantize_pack_uniform(W, arm) for W in Ws]

        def layer():
            for p in lin:
                packing.linear_layout(xs[p.K], p, out=ys[p.N], path="auto")
        layer()
        torch.cuda.synchronize()
        ms = _events_ms(layer, iters)
        nbytes = sum(p.nbytes() for p in lin)
        copies = [lin] + [[type(p)(p.qweight.clone(), p.rowmeta.clone(), *((p.N, p.K, p.layout) if arm != "mixed" else (p.N, p.K)))
                           for p in lin] for _ in range(int(600e6 / nbytes))]

        def tokens1():
            for c in copies:
                for p in c:
                    packing.linear_layout(x1[p.K], p, out=y1[p.N], path="auto")
        tokens1()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            tokens1()
        gr.replay()
        torch.cuda.synchronize()
        us = _events_ms(gr.replay, 5) * 1e3 / len(copies)       # per layer (7 launches)
        out["arms"][arm] = {"bits_per_weight": round(8.0 * nbytes / LS.PARAMS_PER_LAYER, 3),
                            "prefill": {"layer_ms": round(ms, 3), "TFLOPs": round(fl / ms / 1e9, 1),
                                        "mfma_frac": round(fl / ms / 1e9 / PEAK_F16_TFLOPS, 4)},
                            "decode_M1": {"layer_us": round(us, 2), "packed_MB": round(nbytes / 1e6, 2),
                                          "GBps": round(nbytes / us / 1e3, 1), "frac": round(nbytes / us / 1e3 / PEAK_HBM_GBPS, 4)}}
        del copies, gr, lin
    return out


def fused_launch_figure(layers, dev, x_h, x_i, y_h, steps=5):
    """Side figure, not `value`: the same 224 Linears with q|k|v and gate|up as ONE launch each (128 launches per step;
    the packed blocks are independent per 16 rows, so the fused weight is the concatenation along the output dimension --
    what the decode stage does, mxq_amd/llama_decode.py).  Same FLOPs, same outputs (concatenated); fewer, longer
    launches: q|k|v is exactly 3 rounds of tiles and gate|up's tail is a smaller share of its launch."""
    fused = [[("qkv", packing.concat_packed([p for _, p in lin[0:3]])), lin[3],
              ("gate_up", packing.concat_packed([p for _, p in lin[4:6]])), lin[6]] for lin in layers]
    ys = {N: torch.empty(SEQ, N, device=dev, dtype=torch.float16) for N in (3 * LS.HIDDEN, 2 * LS.INTERMEDIATE)}

    def step():
        for lin in fused:
            for _name, p in lin:
                x = x_i if p.K == LS.INTERMEDIATE else x_h
                packing.linear(x, p, out=(ys[p.N] if p.N in ys else y_h), path="auto")
    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    fl = LS.linear_flops(SEQ, len(layers))
    return {"workload": "the headline's 224 Linears as 128 launches per step: q|k|v and gate|up fused along the output "
                        "dimension (NOT the headline configuration)",
            "ms_per_step": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1), "launches_per_step": 4 * len(layers)}


class ClockStamps:
    """Shader clock held between two points of a stream: 8 one-wave workgroups (one per XCD) stamp s_memtime and the 100-MHz
    s_memrealtime (include/mxq_hip.h: mxq_clock_stamp); sclk = d(s_memtime) / d(s_memrealtime) x 100 MHz per XCD."""

    def __init__(self, dev):
        from mxq_amd import _lib
        self.lib, self.dev = _lib.load(), dev
        self.a = torch.zeros(32, dtype=torch.int64, device=dev)
        self.b = torch.zeros(32, dtype=torch.int64, device=dev)

    def stamp(self, which):
        from mxq_amd import _lib
        t = self.a if which == 0 else self.b
        _lib.check(self.lib.mxq_clock_stamp(t.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream), "mxq_clock_stamp")

    def mhz(self):
        a, b = self.a.view(8, 4).cpu(), self.b.view(8, 4).cpu()
        by_xcc = {}
        for r0 in a.tolist():
            for r1 in b.tolist():
                if r0[2] == r1[2] and r1[1] > r0[1]:
                    by_xcc[int(r0[2])] = (r1[0] - r0[0]) / (r1[1] - r0[1]) * 100.0
        vals = sorted(by_xcc.values())
        if not vals:
            return None
        return {"median": round(vals[len(vals) // 2], 1), "min": round(vals[0], 1), "max": round(vals[-1], 1), "xcds": len(vals)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="no side figures and no CPU baseline: the process launches nothing but the headline's kernels "
                         "(what tools/prof_round.sh traces, so that the trace's per-kernel average is the headline's)")
    ap.add_argument("--figure", choices=["decode_1gpu", "fakequant_block", "config5"], default=None,
                    help="run ONE side figure only and print it (no headline): what tools/prof_round.sh traces per figure, so "
                         "that a trace's per-kernel average x launches reproduces the figure")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1: replay the step as one hipGraph (measured r04: 1073.3 vs 1073.4 TFLOP/s stream-ordered -- the "
                         "queue never runs dry, so this is not the default)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC constant instead of rocprofv3 child passes in this run (~40 s)")
    ap.add_argument("--no-decode-pipeline", action="store_true",
                    help="N > 1: skip the bounded configs[2] side figure (greedy decode through the layer pipeline)")
    ap.add_argument("--fuse", action="store_true",
                    help="NOT the headline configuration: q|k|v and gate|up as one launch each (5 launches per layer "
                         "instead of 7; same weights, same FLOPs), reported with config.fused_launches = true")
    ap.add_argument("--path", default="auto",
                    help="packing.linear path of the timed launches.  Default 'auto' = the product entry, mxq_linear_f16_auto, with "
                         "the workspace and scratch QuantLinear.forward passes; 'gemm', 'whole', 'gemm8', ... pick one schedule (A/B)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU fallback for the measured path)")
    # (MXQ_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks: ranks then
    # share devices and the hidden state hops through host memory; numbers from such a run mean nothing)
    backend = os.environ.get("MXQ_BENCH_BACKEND", "nccl")
    local_rank = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    # N > 1: every phase has a deadline (a stuck rank prints where it is stuck and exits 86; the launcher names it), the whole
    # run a hard one (450 s) below the driver's 600 s, and the process group a 120-s timeout instead of torch's 10 minutes
    dog = pipeline.Watchdog(rank, world, hard_deadline_s=float(os.environ.get("MXQ_BENCH_DEADLINE_S", 450))) if world > 1 else None

    def phase(name, seconds):
        if dog is not None:
            dog.phase(name, seconds)
    phase("rendezvous", 150)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        pipeline.init_group(backend, rank, world, dev)
        fault = os.environ.get("MXQ_BENCH_FAULT", "")          # tests: "<rank>:exit" | "<rank>:hang" right after the rendezvous
        if fault and int(fault.split(":")[0]) == rank:
            if fault.endswith(":exit"):
                pipeline.rank_log("MXQ_BENCH_FAULT: this rank dies now (exit 17)", rank, world)
                os._exit(17)
            pipeline.rank_log("MXQ_BENCH_FAULT: this rank hangs now", rank, world)
            dog.phase("injected hang", float(os.environ.get("MXQ_BENCH_FAULT_DEADLINE_S", 20)))
            time.sleep(3600)

    if args.figure:
        fig = {"decode_1gpu": decode_1gpu, "fakequant_block": fakequant_block, "config5": config5}[args.figure](dev)
        print(json.dumps({args.figure: fig}), flush=True)
        return
    phase("build weights", 240)
    my_layers = list(LS.layer_range(rank, world))
    layers = build_layers(my_layers, dev)
    if args.fuse:   # LAYER_LINEARS order: q, k, v, o, gate, up, down
        layers = [[("qkv", packing.concat_packed([p for _, p in lin[0:3]])), lin[3],
                   ("gate_up", packing.concat_packed([p for _, p in lin[4:6]])), lin[6]] for lin in layers]
    launches_per_layer = len(layers[0]) if layers else 0
    gx = torch.Generator(device=dev).manual_seed(7)
    x_h = torch.randn(SEQ, LS.HIDDEN, generator=gx, device=dev).half()          # hidden-width input
    x_i = torch.randn(SEQ, LS.INTERMEDIATE, generator=gx, device=dev).half()    # down_proj input
    y_h = torch.empty(SEQ, LS.HIDDEN, device=dev, dtype=torch.float16)
    y_i = torch.empty(SEQ, LS.INTERMEDIATE, device=dev, dtype=torch.float16)
    y_f = {N: torch.empty(SEQ, N, device=dev, dtype=torch.float16) for N in (3 * LS.HIDDEN, 2 * LS.INTERMEDIATE)} if args.fuse else {}
    recv_buf = torch.empty_like(x_h)
    n_micro = world          # sequences in flight per step: per-GPU work stays one full-model pass

    def stage(x_hidden):
        # every launch goes through packing.linear(path="auto") -> mxq_linear_f16_auto: the entry QuantLinear.forward calls,
        # with the per-stream workspace it passes (2048 tokens: the 256-token fused kernel, stream-K tail on gate / up)
        for lin in layers:
            for name, p in lin:
                x = x_i if p.K == LS.INTERMEDIATE else x_hidden
                packing.linear(x, p, out=(y_f[p.N] if p.N in y_f else y_i if p.N == LS.INTERMEDIATE else y_h), path=args.path)

    # N > 1: the schedule is mxq_amd.pipeline.LayerPipeline's (the same object the gloo tests drive on CPU): irecv of
    # micro-batch b+1 posted before b is computed, isend of b's output from a ring slot under b+1's compute
    pipe = LayerPipeline(rank, world) if world > 1 else None
    stats = None

    def stage_fn(h):
        stage(h)
        return y_h        # the stage's OUTPUT hops on (written last by the final layer's down_proj): the next stage's
                          # GEMMs depend on this stage's GEMMs and on the transfer, as in a real layer pipeline

    def step():
        if pipe is None:
            stage(x_h)
        else:
            pipe.run_microbatches(stage_fn, [x_h] * n_micro, recv_buf, collect=False, stats=stats)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    phase("warmup", 150)
    for _ in range(args.warmup):
        step()
    sync()
    # N = 1: the step's 224 launches are captured ONCE in a hipGraph and every timed step is one replay (the launches,
    # their order and their arguments are those of stage(); the ~2 us host-side gap between stream-ordered launches goes).
    # N > 1 keeps stream-ordered launches: the pipeline's RCCL send / recv are not capturable.
    graph = None
    if pipe is None and args.graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            stage(x_h)

        def step():   # noqa: F811
            graph.replay()
        step()
        sync()
    phase("timed steps", 150)
    stats = pipeline.PipelineStats(dev) if pipe is not None else None
    clocks = ClockStamps(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    clocks.stamp(0)
    for _ in range(args.steps):
        step()
    clocks.stamp(1)
    ev1.record()
    sync()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    phase("reports", 90)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # a stream-K wait that gave up in the timed region (tiles would be NaN) must fail the run: N = 1 raises here; N > 1
    # reports per rank first (every rank's verdict travels to rank 0's line), then every rank exits 4
    ws_status = "ok"
    try:
        packing.workspace_status(dev)
    except RuntimeError as e:
        if world == 1:
            raise
        ws_status = str(e)
    sclk = clocks.mhz()
    bits_per_weight = sum(p.nbytes() for lin in layers for _, p in lin) * 8.0 / (LS.PARAMS_PER_LAYER * max(1, len(my_layers)))
    tokens_per_step = SEQ * n_micro
    flops_per_step = LS.linear_flops(SEQ) * n_micro                      # whole job
    flops_rank_step = LS.linear_flops(SEQ, len(my_layers)) * n_micro     # this rank's launches
    launches_rank_step = launches_per_layer * len(my_layers) * n_micro
    ms_per_step = elapsed / args.steps * 1e3
    value = flops_per_step / (elapsed / args.steps) / 1e12
    kern_ms = dev_ms / args.steps / launches_rank_step
    achieved = flops_rank_step / (dev_ms / args.steps * 1e-3) / 1e12

    per_rank = None
    if world > 1:        # every rank's own account of the timed region, on rank 0's line: one record explains its curve
        summ = stats.summary(args.steps)
        comp = summ.get("stream_compute_ms_per_step") or 0.0
        rep = {"rank": rank, "layers": [my_layers[0], my_layers[-1]] if my_layers else [], "launches_per_step": launches_rank_step,
               "device_ms_per_step": round(dev_ms / args.steps, 4), "host_ms_per_step": round(ms_per_step, 4),
               # this rank's GEMMs alone (HIP events around its stage_fn calls): flat in N if the hops hide under compute
               "stage_TFLOPs": round(flops_rank_step / comp / 1e9, 1) if comp > 0 else None,
               "workspace_status": ws_status, "sclk_MHz": sclk}
        rep.update(summ)
        per_rank = pipeline.gather_reports(rep)
        bad = [r for r in per_rank if r["workspace_status"] != "ok"]
        if bad:
            if rank == 0:
                print("bench.py: stream-K wait expired on rank(s) " + ", ".join(f"{r['rank']}: {r['workspace_status']}" for r in bad),
                      file=sys.stderr, flush=True)
            raise SystemExit(4)

    traffic = None       # HBM bytes per launch from rocprofv3 PMC counters (collected offline, see the file)
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_gemm8_traffic.json") for r in (6, 5, 4, 3, 2))
                  if os.path.exists(q)), "")
    if world == 1 and tpath:
        traffic = json.load(open(tpath))["avg_hbm_bytes_per_launch"]

    # N > 1: who took part (all-reduce of ones + every rank's device identity), and a bounded BASELINE configs[2] figure:
    # 32 greedy-decode tokens through the same layer pipeline, rank 0 re-decoding them in one process to compare the ids.
    # After the timed region; every rank runs it.
    census = rank_census(dev) if world > 1 else None
    if census is not None and backend == "nccl" and (census["ranks_seen"] != world or census["distinct_devices"] != world):
        # N ranks on fewer than N GPUs (or a rank missing) is not the N-GPU measurement the line would claim: fail loudly,
        # on every rank, with a non-zero exit (the gloo rehearsal mode shares devices on purpose and is exempt)
        if rank == 0:
            print(f"bench.py: --gpus {world} under RCCL needs {world} ranks on {world} distinct devices, saw ranks_seen="
                  f"{census['ranks_seen']} distinct_devices={census['distinct_devices']}: {census['ranks']}", file=sys.stderr, flush=True)
        raise SystemExit(3)
    decode_fig = None
    if world > 1 and rank == 0:
        # the headline of this run, on stderr, BEFORE the side figure: should the decode pipeline fail on hardware it has never
        # seen, the record still holds what the timed region measured (stdout carries the one complete line at the end)
        pipeline.rank_log("headline so far: " + json.dumps({"value_TFLOPs": round(value, 3), "n_gpus": world, "steps": args.steps,
                                                            "ms_per_step": round(ms_per_step, 4), "per_rank": per_rank}), rank, world)
    if world > 1 and not args.no_decode_pipeline:
        phase("decode pipeline figure", 240)
        from mxq_amd.llama_decode import decode_pipeline_figure
        decode_fig = decode_pipeline_figure(pipe, dev, tokens=32, ctx=64, verify=True, dist=dist, backend=backend)
    # (rank 0 alone re-decodes the tokens in one process inside the decode figure and then prints: the other ranks wait for it in
    #  the teardown barrier, so their deadlines cover rank 0's extra work -- and stay below the 120-s group timeout that bounds
    #  the barrier itself)
    phase("print", 420 if world == 1 else 100)
    if rank == 0:
        bpw = bits_per_weight
        out = {
            "metric": "quantized-Linear TFLOP/s, Llama-2-7B W2/4A16 all-Linear prefill (tokens/s alongside)",
            "value": round(value, 3), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "tokens_per_s": round(tokens_per_step / (elapsed / args.steps), 1),
            "config": {"workload": "BASELINE configs[1]: Llama-2-7B all 224 quantized Linears (32 layers x "
                                   "q,k,v,o,gate,up,down), batch=1 seq=2048 (M=2048), MXQ mixed 2/4-bit weights "
                                   "(48x2b+16x4b per 64), fp16 activations, fp32 accumulate",
                       "tokens_per_step": tokens_per_step, "flop_per_step": flops_per_step,
                       "weight_format": "mxq-v1 exact metadata", "bits_per_weight": round(bpw, 3),
                       "fused_launches": bool(args.fuse),
                       "entry": ("packing.linear(path='auto') -> mxq_linear_f16_auto (the product dispatch QuantLinear.forward calls)"
                                 if args.path == "auto" else f"packing.linear(path={args.path!r}) (A/B: NOT the product dispatch)"),
                       "launch_mode": "hipGraph replay of the step's launches" if graph is not None else "stream-ordered launches",
                       "parallelism": "single GPU" if world == 1 else
                       f"pp{world}: whole layers sharded, {world} sequences in flight, RCCL send/recv of the "
                       f"[2048,4096] fp16 hidden state"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                         "traffic_source": (f"offline constant: rocprofv3 PMC passes of this bench "
                                            f"(2*FETCH_SIZE + WRITE_SIZE per launch), {os.path.relpath(tpath, ROOT)}; "
                                            "not measured in this run") if traffic is not None else None,
                         "kernel": "mxq_gemm8_f16_kernel", "avg_launch_ms": round(kern_ms, 5),
                         "launches_per_step": launches_rank_step,
                         "algorithmic_flop_per_launch": flops_rank_step / launches_rank_step,
                         # the shader clock the chip held over the timed region (s_memtime / s_memrealtime stamps on the launch
                         # stream, per XCD): the peak is quoted at 2400 MHz, so frac_at_held_clock = frac x 2400 / sclk
                         "sclk_MHz_during_timed_region": sclk,
                         "frac_at_held_clock": (round(achieved / (PEAK_F16_TFLOPS * sclk["median"] / 2400.0), 4)
                                                if sclk and sclk["median"] > 0 else None)},
        }
        if world == 1 and not args.headline_only and not args.no_live_traffic:
            lt = live_traffic()
            if lt is not None:
                out["roofline"]["traffic"] = lt[0]
                out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes "
                                                     "(separate) of tools/gemm_prof.py at the step's three shapes, (2*FETCH_SIZE + WRITE_SIZE) KiB "
                                                     "per launch, weighted by launches per step")
                out["roofline"]["traffic_by_shape"] = lt[1]
        if census is not None:
            out["ranks_seen"] = census["ranks_seen"]
            out["distinct_devices"] = census["distinct_devices"]
            out["ranks"] = census["ranks"]
            out["backend"] = backend
            out["per_rank"] = per_rank
            out["phases_s"] = dog.history
        if decode_fig is not None:
            decode_fig.pop("token_ids", None)
            out["decode_pipeline"] = decode_fig
        if world == 1 and not args.fuse and not args.headline_only:
            out["fused_launches_figure"] = fused_launch_figure(layers, dev, x_h, x_i, y_h)
            # the other BASELINE configs, each with its own roofline fraction (side figures; `value` is configs[1])
            out["decode_1gpu"] = decode_1gpu(dev)              # configs[2] on one GPU (HBM-bound)
            out["fakequant_block"] = fakequant_block(dev)      # configs[3] (HBM-bound)
            out["config5"] = config5(dev)                      # configs[4] (MFMA-bound prefill leg, HBM-bound decode leg)
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            out["cpu_baseline"] = cpu_baseline(dev)
        print(json.dumps(out), flush=True)
    if dist is not None:
        phase("teardown", 110)
        dist.barrier()
        dist.destroy_process_group()
    if dog is not None:
        dog.done()


if __name__ == "__main__":
    pipeline.run_guarded(main)     # any failure of a rank: tagged traceback on stderr, immediate non-zero exit
