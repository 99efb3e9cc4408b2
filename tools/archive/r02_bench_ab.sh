#!/bin/bash
# A/B of one env knob on the headline bench: bash tools/r02_bench_ab.sh VAR v1 v2 ...
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
var=$1; shift
for rep in 1 2; do for val in "$@"; do
  env $var=$val timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_${var}_$val.json 2> $O/bench_${var}_$val.err
  python - <<PY
import json
b=json.load(open("$O/bench_${var}_$val.json")); print("$var=$val", b["kv_compress_ms_per_layer"], b["call_effective"]["frac_of_8TBps"], {k:v["avg_us"] for k,v in b["roofline_kernels"].items()})
PY
done; done
