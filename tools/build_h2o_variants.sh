#!/bin/bash
# Builds A/B variants of the H2O kernels as complete libraries tools/_h2o_<name>.so (same ABI; load through PKV_LIB).
#   tools/build_h2o_variants.sh [--src file.hip] name1="-DFLAG=1" name2="" ...
#   --src tools/probes/h2o_wide_pipeline.hip : round 4's 32x32x16 software pipeline (knobs -DH2O_PIPE=0, -DH2O_LB=3, -DH2O_TRACK=1,
#   -DH2O_ABLATE=1) instead of the shipped pyramidkv_amd/csrc/pkv_h2o.hip
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/pyramidkv_amd/csrc
SRC=$C/pkv_h2o.hip
if [ "${1:-}" = "--src" ]; then SRC=$(cd "$(dirname "$2")" && pwd)/$(basename "$2"); shift 2; fi
make -C $C -j16 >/dev/null 2>&1
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-function -fno-slp-vectorize"
OTHERS=$(ls $C/*.o | grep -v "\.dbg\.o" | grep -v pkv_h2o.o)
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  /opt/rocm/bin/hipcc $FLAGS -fvisibility=hidden -I$C $defs -c $SRC -o /tmp/_h2o_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -Wl,--version-script=$C/pkv.map $OTHERS /tmp/_h2o_$name.o -ldl -o $R/tools/_h2o_$name.so
  echo "built tools/_h2o_$name.so ($defs)"
done
