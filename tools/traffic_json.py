#!/usr/bin/env python3
"""HBM traffic per launch of the prefill GEMM from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE
runs of tools/gemm_prof.py, see tools/prof_round.sh) -> profiles/<tag>_gemm8_traffic.json, which bench.py reports as
roofline.traffic.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: the gfx950 correction of MI355X_MICROARCH.md, section HBM
(FETCH_SIZE reports half the bytes of a wide coalesced read; WRITE_SIZE is exact for 16-byte stores).

    python tools/traffic_json.py gpurun_out r02 profiles/r02_gemm8_traffic.json
"""
import csv
import glob
import json
import os
import sys

src, tag, dst = sys.argv[1], sys.argv[2], sys.argv[3]
M = 2048
SHAPES = [(4096, 4096, 128), (11008, 4096, 64), (4096, 11008, 32)]      # N, K, launches per bench step


def counter(dirname, name):
    vals = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm8" in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {name} rows for the gemm8 kernel under {dirname}")
    tail = vals[len(vals) // 2:]            # drop the warm-up half
    return sum(tail) / len(tail)


out = {"kernel": "mxq_gemm8_f16_kernel",
       "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_round.sh: "
              "tools/gemm_prof.py gemm 2048 N K 6, mean of the last 3 launches); bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB "
              "with the gfx950 x2 correction of MI355X_MICROARCH.md section HBM",
       "shapes": {}}
tot_hbm = tot_alg = tot_n = 0.0
for N, K, n in SHAPES:
    f = counter(os.path.join(src, f"{tag}_pmc_FETCH_SIZE_{N}x{K}"), "FETCH_SIZE")
    w = counter(os.path.join(src, f"{tag}_pmc_WRITE_SIZE_{N}x{K}"), "WRITE_SIZE")
    hbm = (2 * f + w) * 1024
    alg = (N // 16) * (K // 64) * 576 + N * 16 + 2 * M * K + 2 * M * N
    out["shapes"][f"{M}x{N}x{K}"] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "launches_per_step": n,
                                     "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": float(alg)}
    tot_hbm += hbm * n
    tot_alg += alg * n
    tot_n += n
out["avg_hbm_bytes_per_launch"] = tot_hbm / tot_n
out["avg_algorithmic_bytes_per_launch"] = tot_alg / tot_n
out["note"] = ("FETCH_SIZE counts what the 8 per-XCD L2s request from the fabric (Infinity Cache / HBM): every XCD that "
               "works on a tile row / weight panel fetches its own copy; the bytes actually read from HBM are lower "
               "(the 256 MiB Infinity Cache serves the repeats).")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.startswith("avg")}))
