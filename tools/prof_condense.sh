#!/bin/bash
# Condense what tools/prof_round.sh <tag> left under gpurun_out/ into the committed profiles/r02_* files
# (run here, after the GPU call):   bash tools/prof_condense.sh r02g
set -e
T=${1:-r02}
cd "$(dirname "$0")/.."
python3 tools/prof_summary.py gpurun_out/${T}_bench profiles/r02_bench_gemm8.txt "bench.py --steps 5 --warmup 2 --no-cpu-baseline"
cp gpurun_out/${T}_bench.json profiles/r02_bench_gemm8.json
python3 tools/traffic_json.py gpurun_out ${T} profiles/r02_gemm8_traffic.json
{ echo "# rocprofv3 --pmc passes of tools/gemm_prof.py gemm 2048 4096 4096 (tools/prof_round.sh ${T})"
  python3 tools/pmc_summary.py gpurun_out/${T}_pmc_sq1; python3 tools/pmc_summary.py gpurun_out/${T}_pmc_sq2; } > profiles/r02_gemm8_pmc.txt
grep -v amdgpu.ids gpurun_out/${T}_kernels_bench.txt > profiles/r02_kernels_bench.txt
{ echo "# BASELINE configs[4] arms at M = 32768 (4096^2 Linear): rocprofv3 --pmc passes of tools/gemm_prof.py (tools/prof_round.sh ${T})"
  for lay in mixed w2g16 w4row; do echo "## $lay"; python3 tools/pmc_summary.py gpurun_out/${T}_c5_${lay}_FETCH; python3 tools/pmc_summary.py gpurun_out/${T}_c5_${lay}_WRITE; python3 tools/pmc_summary.py gpurun_out/${T}_c5_${lay}_SQ; done; } > profiles/r02_config5_pmc.txt 2>&1
python3 - <<PY
import json
lines=[l for l in open('gpurun_out/${T}_config5_sweep.log') if l.startswith('{"config"')]
d=json.loads(lines[-1])
d["note"]="tools/prof_round.sh ${T}: launches of >= 8192 tokens take the hoisted-dequant mode automatically (dequant once into a transient fp16 scratch + the MFMA kernel on fp16 tiles, bit-identical results); the dequant pass is inside every timed launch.  Earlier in round 2, fused mode on another box: mixed 965, W2G16 1032, W4ROW 1029, hipBLASLt 1292 TFLOP/s per layer."
json.dump(d,open('profiles/r02_config5_sweep.json','w'),indent=1)
print({k:v["TFLOPs"] for k,v in d["arms"].items()})
PY
