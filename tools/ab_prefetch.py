#!/usr/bin/env python3
"""Does an Infinity-Cache prefetch on a side stream speed up the decode GEMV chain?  (GPU box)

A decode step is a chain of dependent GEMV launches (q|k|v, o, gate|up, down per layer), each of which spends part of
its time in dequant arithmetic with the HBM idle.  This tool replays such a chain (distinct weights for ``--layers``
layers, so every replay streams from HBM) as one hipGraph, with and without ``mxq_prefetch`` of the weights of the
launch ``D`` positions ahead running on a second stream under the current launch, and prints the time per layer.

    python tools/ab_prefetch.py [--layers 8] [--dist 0,1,2,3] [--wgs 512,1024,2048]

Result (profiles/r02_prefetch_ab.txt): every cross-stream edge of the graph costs ~13 us, the chain gets 2x SLOWER.
A second experiment appended prefetch workgroups to the GEMV launch itself (no extra launch, no cross-queue edge):
also slower, 49.6 vs 41.1 us per layer -- and the GEMV reading Infinity-Cache-resident weights is only ~15 % faster
than from HBM (tools/ab_gemv.py --mb 1 --repeat 40), so there was nothing to win; that variant is not kept in the code.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mxq_amd import _lib, packing  # noqa: E402
from mxq_amd import llama_shapes as LS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--dist", default="0,1,2,3")
    ap.add_argument("--wgs", default="512,1024,2048")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--compact", action="store_true")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    H, I = LS.HIDDEN, LS.INTERMEDIATE
    g = torch.Generator(device=dev).manual_seed(0)

    def mk(N, K):
        return packing.quantize_pack((torch.randn(N, K, generator=g, device=dev) * 0.02).half(), compact_meta=args.compact)
    base = [mk(3 * H, H), mk(H, H), mk(2 * I, H), mk(H, I)]
    chain = []          # (x, packed, out)
    xs = {H: torch.randn(1, H, generator=g, device=dev).half(), I: torch.randn(1, I, generator=g, device=dev).half()}
    for _ in range(args.layers):
        for b in base:
            p = packing.PackedMXQ(b.qweight.clone(), b.rowmeta.clone(), b.N, b.K, b.compact)
            chain.append((xs[b.K], p, torch.empty(1, b.N, device=dev, dtype=torch.float16)))
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    bytes_per_layer = sum(b.nbytes() for b in base)
    side = torch.cuda.Stream(device=dev)

    def prefetch(p, wgs):
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.mxq_prefetch(p.qweight.data_ptr(), p.qweight.numel() * p.qweight.element_size(), wgs,
                                    sink.data_ptr(), st), "mxq_prefetch")

    def build(dist, wgs):
        def run():
            main_s = torch.cuda.current_stream(dev)
            for k, (x, p, out) in enumerate(chain):
                if dist > 0 and k + dist < len(chain):
                    side.wait_stream(main_s)
                    with torch.cuda.stream(side):
                        prefetch(chain[k + dist][1], wgs)
                packing.linear(x, p, out=out, path="gemv")
            if dist > 0:
                main_s.wait_stream(side)
        run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        return gr

    variants = [(0, 0)] + [(d, w) for d in (int(v) for v in args.dist.split(",")) if d > 0
                           for w in (int(v) for v in args.wgs.split(","))]
    graphs = {v: build(*v) for v in variants}
    ts = {v: [] for v in variants}
    for _ in range(args.rounds):
        for v in variants:
            gr = graphs[v]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / args.layers * 1e3)
    report = []
    for v in variants:
        t = sorted(ts[v])
        med = t[len(t) // 2]
        row = {"dist": v[0], "prefetch_wgs": v[1], "us_per_layer": round(med, 2), "us_min": round(t[0], 2),
               "weight_TBps": round(bytes_per_layer / med / 1e6, 3)}
        report.append(row)
        print(json.dumps(row), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"layers": args.layers, "bytes_per_layer": bytes_per_layer, "rows": report}, f, indent=1)


if __name__ == "__main__":
    main()
