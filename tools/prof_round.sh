#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/prof_round.sh <tag>     ->  gpurun_out/<tag>_*  (summarise with tools/prof_summary.py / pmc_summary.py)
# --pmc passes are separate runs with --kernel-trace only (never combined with the hip/hsa trace domains).
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench trace done"
for shape in "4096 4096" "11008 4096" "4096 11008"; do
  set -- $shape
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_${c}_$1x$2 -- python3 $R/tools/gemm_prof.py gemm 2048 $1 $2 6 > /dev/null 2>> $OUT/${TAG}_pmc.err
  done
  echo "traffic $1x$2 done"
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${TAG}_pmc_sq1 -- python3 $R/tools/gemm_prof.py gemm 2048 4096 4096 8 > /dev/null 2>> $OUT/${TAG}_pmc.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/${TAG}_pmc_sq2 -- python3 $R/tools/gemm_prof.py gemm 2048 4096 4096 8 > /dev/null 2>> $OUT/${TAG}_pmc.err
echo "sq counters done"
cd $R
python3 tools/kernels_bench.py > $OUT/${TAG}_kernels_bench.txt 2>&1
echo "kernels bench done"
