#!/bin/bash
# usage (through gpurun, repo root): bash tools/probes/hipblaslt_names.sh  ->  gpurun_out/hbl_names.txt
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hbl_names -- python3 $R/tools/probes/hipblaslt_names.py 2048,1024,1536 > /dev/null 2>&1
cd $R
f=$(find gpurun_out/hbl_names -name '*kernel_trace.csv' | tail -1)
python3 - "$f" > gpurun_out/hbl_names.txt <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if not n.startswith("Cijk"):
        continue
    k = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print(len(v), "launches  median us", round(sorted(v)[len(v) // 2], 1), " grid", k[1], " wg", k[2], "\n   ", k[0])
PY
cat gpurun_out/hbl_names.txt
