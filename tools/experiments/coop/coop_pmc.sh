#!/bin/bash
# L2 counters of the cooperative mode: stores into the image that is read (dbg 4) vs into a region nobody reads (dbg 1028)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
for d in 4 1028; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
    tag=$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/coop_pmc_${d}_${tag} -- python3 $R/tools/gemm_prof.py gemm8 2048 4096 4096 12 mixed $d > /dev/null 2>> $OUT/coop_pmc.err
    echo "== dbg $d $c"; python3 $R/tools/pmc_summary.py $OUT/coop_pmc_${d}_${tag} gemm8
  done
done
