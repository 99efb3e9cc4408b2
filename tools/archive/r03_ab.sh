#!/bin/bash
# A/B of env-knob configurations on the headline bench (and the un-expanded-GQA extra):
#   bash tools/r03_ab.sh REPS "name1:VAR=a,VAR2=b" "name2:" ...      (empty list after ':' = defaults)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab3
mkdir -p $O
cd $R
reps=$1; shift
for rep in $(seq 1 $reps); do for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  envs=${envs//,/ }
  env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --only-gqa-extra > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    b=json.load(open("$O/bench_$name.json"))
    g=b.get("extras",{})
    print("%-14s rep$rep call %.2f us  eff %.4f | %s | gqa %.2f us %s" % ("$name", b["kv_compress_ms_per_layer"]*1e3, b["call_effective"]["frac_of_8TBps"],
          {k:v["avg_us"] for k,v in b["roofline_kernels"].items() if not k.endswith("gqa4")}, g.get("unexpanded_gqa_us_per_layer",0),
          {k:v["avg_us"] for k,v in b["roofline_kernels"].items() if k.endswith("gqa4")}))
except Exception as e:
    print("$name failed", e, open("$O/bench_$name.err").read()[-400:])
PY
done; done
