#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter_collection CSVs per kernel.
    python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys

src, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "mxq")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[len(v) // 2:]          # drop the warm-up half
        print(f"   {c:<34} {sum(v)/len(v):16.1f}   (n={len(v)})")
