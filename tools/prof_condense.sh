#!/bin/bash
# Condense what tools/prof_round.sh <tag> left under gpurun_out/ into the committed profiles/${P}_* files
# (run here, after the GPU call):   bash tools/prof_condense.sh <tag> <prefix>, e.g.  r03a r03
set -e
T=${1:-r02}
P=${2:-r02}        # prefix of the committed files: profiles/<P>_*
cd "$(dirname "$0")/.."
python3 tools/prof_summary.py gpurun_out/${T}_bench profiles/${P}_bench_gemm8.txt "bench.py --steps 5 --warmup 2 --headline-only"
cp gpurun_out/${T}_bench.json profiles/${P}_bench_gemm8.json
for fig in decode_1gpu fakequant_block config5; do
  if [ -d gpurun_out/${T}_fig_$fig ]; then
    python3 tools/prof_summary.py gpurun_out/${T}_fig_$fig profiles/${P}_fig_$fig.txt "bench.py --figure $fig" > /dev/null
    grep "^{" gpurun_out/${T}_fig_$fig.json | tail -1 > profiles/${P}_fig_$fig.json
  fi
done
python3 tools/traffic_json.py gpurun_out ${T} profiles/${P}_gemm8_traffic.json
{ echo "# rocprofv3 --pmc passes of tools/gemm_prof.py gemm 2048 4096 4096 (tools/prof_round.sh ${T})"
  python3 tools/pmc_summary.py gpurun_out/${T}_pmc_sq1; python3 tools/pmc_summary.py gpurun_out/${T}_pmc_sq2; } > profiles/${P}_gemm8_pmc.txt
grep -v amdgpu.ids gpurun_out/${T}_kernels_bench.txt > profiles/${P}_kernels_bench.txt
{ echo "# BASELINE configs[4] arms at M = 32768 (4096^2 Linear): rocprofv3 --pmc passes of tools/gemm_prof.py (tools/prof_round.sh ${T})"
  for lay in mixed w2g16 w4row; do echo "## $lay"; python3 tools/pmc_summary.py gpurun_out/${T}_c5_${lay}_FETCH; python3 tools/pmc_summary.py gpurun_out/${T}_c5_${lay}_WRITE; python3 tools/pmc_summary.py gpurun_out/${T}_c5_${lay}_SQ; done; } > profiles/${P}_config5_pmc.txt 2>&1
if [ -d gpurun_out/${T}_midm_trace ]; then
  python3 tools/prof_summary.py gpurun_out/${T}_midm_trace profiles/${P}_midm_trace.txt "tools/midm_bench.py --ms 64,128,192 --paths auto --no-torch"
  { echo "# csrc/midm.hip at 128 tokens x 4096^2 (BASELINE configs[0]'s shape): rocprofv3 --pmc passes of tools/gemm_prof.py midm 128 4096 4096 8 (tools/prof_round.sh ${T})"
    echo "# HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction, MI355X_MICROARCH.md section HBM); algorithmic: 9.2 MB packed weight + 1 MB x + 1 MB y,"
    echo "# plus, by design, S x 2 MB of fp32 partial tiles written by the main kernel and read back by the combine kernel (S = 8 slices at 128 tokens)"
    python3 tools/pmc_summary.py gpurun_out/${T}_midm_FETCH_SIZE midm; python3 tools/pmc_summary.py gpurun_out/${T}_midm_WRITE_SIZE midm; python3 tools/pmc_summary.py gpurun_out/${T}_midm_sq midm
    if [ -d gpurun_out/${T}_g8hs_FETCH_SIZE ]; then
      echo "# the product dispatch's path at this shape since round 4 (capi.hip small_tile_launch): csrc/gemm8n.hip (128 x 64 tile) in slices mode (64 tiles x 4 K-slices) + mxq_gemm8n_combine_kernel;"
      echo "# tools/gemm_prof.py gemm8n_slices 128 4096 4096 8 (what path auto runs at this shape) -- 4 x 2 MB of fp32 slabs are written by the first launch and read by the second (the mid-M kernel and the 128 x 128 tile: 8 x 2 MB)"
      python3 tools/pmc_summary.py gpurun_out/${T}_g8hs_FETCH_SIZE gemm8n; python3 tools/pmc_summary.py gpurun_out/${T}_g8hs_WRITE_SIZE gemm8n; python3 tools/pmc_summary.py gpurun_out/${T}_g8hs_sq gemm8n
      echo "# => HBM-side bytes per call: main (2 x 5888 + 8192) KiB + combine (2 x 4119 + 1027) KiB = 29.2 MB for 11.3 MB algorithmic = 2.6 x (mid-M kernel and 128 x 128 tile: 4.0 x)"
    fi; } > profiles/${P}_midm_pmc.txt
fi
if [ -d gpurun_out/${T}_awq_trace ]; then
  python3 tools/prof_summary.py gpurun_out/${T}_awq_trace profiles/${P}_awq_trace.txt "tools/awq_gemm_bench.py (gemm_forward_cuda on the reference operand format: csrc/gemm8a / gemm8ah / gemm8aq + torch's fp16 GEMM beside it)" > /dev/null
fi
python3 - <<PY
import json
lines=[l for l in open('gpurun_out/${T}_config5_sweep.log') if l.startswith('{"config"')]
d=json.loads(lines[-1])
d["hoist_note"]="tools/prof_round.sh ${T}: launches of >= 4096 tokens take the hoisted-dequant mode automatically (dequant once into a transient fp16 scratch + a dense MFMA kernel on fp16 tiles -- csrc/dense256.hip at this size --, bit-identical results); the dequant pass is inside every timed launch."
json.dump(d,open('profiles/${P}_config5_sweep.json','w'),indent=1)
print({k:v["TFLOPs"] for k,v in d["arms"].items()})
PY
