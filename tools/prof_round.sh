#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/prof_round.sh <tag>     ->  gpurun_out/<tag>_*  (summarise with tools/prof_summary.py / pmc_summary.py /
#   traffic_json.py here afterwards, copy the summaries into profiles/)
# --pmc passes are separate runs with --kernel-trace only (never combined with the hip/hsa trace domains); the
# program itself follows "--" (no env / bash -c hop: the profiler's library has initialised the GPU by then).
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_bench -- python3 $R/bench.py --steps 5 --warmup 2 --headline-only > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench trace done"
# the side figures of the bench line (BASELINE configs[2], [3], [4]), each traced on its own: kernel average x launches = the figure
for fig in decode_1gpu fakequant_block config5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_fig_$fig -- python3 $R/bench.py --figure $fig > $OUT/${TAG}_fig_$fig.json 2> $OUT/${TAG}_fig_$fig.err
  echo "figure $fig traced"
done
for shape in "4096 4096" "11008 4096" "4096 11008"; do
  set -- $shape
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_${c}_$1x$2 -- python3 $R/tools/gemm_prof.py gemm 2048 $1 $2 6 > /dev/null 2>> $OUT/${TAG}_pmc.err
  done
  echo "traffic $1x$2 done"
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq1 -- python3 $R/tools/gemm_prof.py gemm 2048 4096 4096 8 > /dev/null 2>> $OUT/${TAG}_pmc.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/${TAG}_pmc_sq2 -- python3 $R/tools/gemm_prof.py gemm 2048 4096 4096 8 > /dev/null 2>> $OUT/${TAG}_pmc.err
echo "sq counters done"
# mid-M split-K kernel (csrc/midm.hip) at BASELINE configs[0]'s shape: kernel trace of the dispatch over a few token counts,
# traffic and SQ counters at 128 tokens x 4096^2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_midm_trace -- python3 $R/tools/midm_bench.py --ms 64,128,192 --paths auto --no-torch > $OUT/${TAG}_midm_trace.txt 2>> $OUT/${TAG}_pmc.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_midm_${c} -- python3 $R/tools/gemm_prof.py midm 128 4096 4096 8 > /dev/null 2>> $OUT/${TAG}_pmc.err
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}_midm_sq -- python3 $R/tools/gemm_prof.py midm 128 4096 4096 8 > /dev/null 2>> $OUT/${TAG}_pmc.err
# the dispatch's own path at that shape since round 4 (path auto): the fused kernel's 128 x 64-tile build in slices mode + its combine launch
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_g8hs_${c} -- python3 $R/tools/gemm_prof.py gemm8n_slices 128 4096 4096 8 > /dev/null 2>> $OUT/${TAG}_pmc.err
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}_g8hs_sq -- python3 $R/tools/gemm_prof.py gemm8n_slices 128 4096 4096 8 > /dev/null 2>> $OUT/${TAG}_pmc.err
echo "midm done"
# BASELINE configs[4] arms at M = 32768: FETCH / WRITE / MFMA-busy per weight layout (4096^2 Linear)
for lay in mixed w2g16 w4row; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_c5_${lay}_FETCH -- python3 $R/tools/gemm_prof.py gemm 32768 4096 4096 4 $lay > /dev/null 2>> $OUT/${TAG}_pmc.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_c5_${lay}_WRITE -- python3 $R/tools/gemm_prof.py gemm 32768 4096 4096 4 $lay > /dev/null 2>> $OUT/${TAG}_pmc.err
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_c5_${lay}_SQ -- python3 $R/tools/gemm_prof.py gemm 32768 4096 4096 4 $lay > /dev/null 2>> $OUT/${TAG}_pmc.err
  echo "config 5 $lay done"
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_hbm_FETCH -- python3 $R/tools/hbm_prof.py > /dev/null 2>> $OUT/${TAG}_pmc.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_hbm_WRITE -- python3 $R/tools/hbm_prof.py > /dev/null 2>> $OUT/${TAG}_pmc.err
echo "hbm kernels done"
# the reference GEMM's operand format on the fused kernel's skeleton (csrc/gemm8a.hip and its builds): kernel trace of tools/awq_gemm_bench.py
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_awq_trace -- python3 $R/tools/awq_gemm_bench.py > $OUT/${TAG}_awq_trace.txt 2>> $OUT/${TAG}_pmc.err
echo "awq trace done"
cd $R
python3 tools/kernels_bench.py > $OUT/${TAG}_kernels_bench.txt 2>&1
echo "kernels bench done"
python3 tools/sweep_config5.py > $OUT/${TAG}_config5_sweep.log 2>&1
echo "config 5 sweep done"
