#!/bin/bash
# Build a variant of libmxq_hip.so in which ONE translation unit is replaced by another source file (same-process A/B
# on the GPU box: tools/midm_bench.py --paths lib:tools/_variants/lib_X.so,...).  usage: build_variant.sh <name> <unit> <source> [extra compiler flags]
#   e.g. tools/build_variant.sh midmA midm abtmp/midm_v1.hip.txt   ->  tools/_variants/lib_midmA.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; unit=$2; src=$3; shift 3; flags="$@"
mkdir -p "$R/tools/_variants"
cd "$R/mxq_amd/csrc"
make -j8 > /dev/null
extra=""; { [ "$unit" = gemm8 ] || [ "$unit" = gemm8h ] || [ "$unit" = gemm8q ]; } && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -I../../include -I. $extra $flags -x hip -c "$R/$src" -o /tmp/variant_$name.o
objs=""; for f in capi pack gemm gemm8 gemm8h gemm8q gemm8n dense256 midm gemv skinny decode_ops gemv_compat gemm_awq fakequant actquant; do
  if [ "$f" = "$unit" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $f.o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o "$R/tools/_variants/lib_$name.so" 2>/dev/null
echo "tools/_variants/lib_$name.so"
