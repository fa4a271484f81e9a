#!/bin/bash
# usage: tools/experiments/decode_engine/build.sh  ->  abtmp/lib_decode_engine.so (the product library's objects + this experiment)
set -e
R=$(cd "$(dirname "$0")/../../.." && pwd)
cd "$R/mxq_amd/csrc" && make -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -I"$R/mxq_amd/csrc" -I"$R/include" -x hip -c "$R/tools/experiments/decode_engine/decode_engine.hip" -o /tmp/decode_engine.o $@
objs=""; for f in capi pack gemm gemm8 dense256 midm gemv skinny decode_ops gemv_compat fakequant actquant; do objs="$objs $f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/decode_engine.o -o "$R/abtmp/lib_decode_engine.so"
echo abtmp/lib_decode_engine.so
