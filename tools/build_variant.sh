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
# the production flags of THIS unit, asked of the Makefile (`make print-CXXFLAGS-<unit>`: ADVICE r5 -- the hand-copied line had
# fallen behind the Makefile's -amdgpu-kernarg-preload-count=16, biasing every A/B by ~0.24 us per launch against the variant)
cxx=$(make -s print-CXXFLAGS-$unit)
/opt/rocm/bin/hipcc $cxx -I. $flags -x hip -c "$R/$src" -o /tmp/variant_$name.o
objs=""; for f in capi pack gemm gemm8 gemm8h gemm8q gemm8n dense256 midm gemv skinny decode_ops gemv_compat gemm8a gemm8ah gemm8aq skinny_awq fakequant actquant; do
  if [ "$f" = "$unit" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $f.o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o "$R/tools/_variants/lib_$name.so" 2>/dev/null
echo "tools/_variants/lib_$name.so"
