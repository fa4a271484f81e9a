#!/bin/bash
# Same-box A/B of two builds of libmxq_hip.so (boxes differ by several %, and so do runs minutes apart):
#   here (build container):   bash tools/ab_libs.sh build <name>      # snapshot the current build as abtmp/lib_<name>.so
#   on the GPU box (gpurun):  bash tools/ab_libs.sh run <a> <b> [rounds] [-- gemm_graph_bench.py args]
# `run` alternates the two libraries <rounds> times and prints tools/gemm_graph_bench.py's lines for each
# (GPU-side time per launch under hipGraph replay).  abtmp/ is scratch (git-ignored).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
case "$1" in
  build)
    mkdir -p abtmp
    make -C mxq_amd/csrc > /dev/null
    cp mxq_amd/libmxq_hip.so abtmp/lib_$2.so
    echo "abtmp/lib_$2.so"
    ;;
  run)
    a=$2; b=$3; rounds=${4:-3}
    shift 4 || true
    [ "$1" = "--" ] && shift
    args=${@:---variants gemm8,gemm8}
    cp mxq_amd/libmxq_hip.so abtmp/lib__restore.so
    for r in $(seq $rounds); do
      for v in $a $b; do
        cp abtmp/lib_$v.so mxq_amd/libmxq_hip.so
        echo "== $v (round $r)"
        timeout -k 10 300 python3 tools/gemm_graph_bench.py $args 2>&1 | grep "M=" || { cp abtmp/lib__restore.so mxq_amd/libmxq_hip.so; exit 1; }
      done
    done
    cp abtmp/lib__restore.so mxq_amd/libmxq_hip.so
    ;;
  *) echo "usage: $0 build <name> | run <a> <b> [rounds] [-- bench args]"; exit 2;;
esac
