#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
var=$1; shift
for rep in 1 2; do for val in "$@"; do
  env $var=$val timeout 300 python tools/dedup_breakdown.py > $O/dedup_${var}_$val.json 2> $O/dedup_${var}_$val.err
  python - <<PY
import json
j=json.load(open("$O/dedup_${var}_$val.json")); print("$var=$val gqa", {k:(v["update_kv_us"], v["logits"], v["finalize"], v["topk"], v["gather"]) for k,v in j.items()})
PY
done; done
