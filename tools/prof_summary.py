#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` output directory into a
small, committed summary under profiles/: per-kernel stats (names truncated) and, for our own
kernels, per-grid-size duration statistics from the kernel trace.

    python tools/prof_summary.py gpurun_out/prof1 profiles/r01_bench_gemm_v1.txt "bench.py --steps 5 --warmup 2"
"""
import collections
import csv
import glob
import os
import sys


def main(src, dst, cmd):
    # (a directory that collected more than one run holds one file set per process id: take the newest)
    stats = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    trace = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    lines = [f"# rocprofv3 --kernel-trace --stats -- python3 {cmd}", f"# source: {src}", ""]
    if stats:
        lines.append("## kernel stats (rocprofv3 *_kernel_stats.csv; names truncated to 90 chars)")
        lines.append(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  name")
        for r in csv.DictReader(open(stats[0])):
            lines.append(f"{int(r['Calls']):7d} {int(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
                         f"{int(r['MinNs'])/1e3:9.2f} {int(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}  "
                         f"{r['Name'][:90]}")
    if trace:
        lines += ["", "## mxq kernels by launch geometry (from *_kernel_trace.csv)",
                  f"{'kernel':<34} {'grid_x':>9} {'wg':>5} {'lds':>6} {'vgpr':>5} {'calls':>6} {'median_us':>10} "
                  f"{'min_us':>9} {'max_us':>9}"]
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(trace[0])):
            n = r["Kernel_Name"]
            if "mxq" not in n and "gemv" not in n:
                continue
            short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-34:]
            key = (short, r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])
            d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in sorted(d.items()):
            v.sort()
            lines.append(f"{k[0]:<34} {k[1]:>9} {k[2]:>5} {k[3]:>6} {k[4]:>5} {len(v):6d} {v[len(v)//2]:10.2f} "
                         f"{v[0]:9.2f} {v[-1]:9.2f}")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
