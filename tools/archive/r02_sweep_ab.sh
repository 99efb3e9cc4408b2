#!/bin/bash
# A/B of one env knob on tools/sweep.py quick (B=1 cap128, B=1 cap2048, B=8 cap2048) and the un-expanded GQA breakdown
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
var=$1; shift
for rep in 1 2; do for val in "$@"; do
  env $var=$val timeout 300 python tools/sweep.py quick > $O/sweep_${var}_$val.json 2> $O/sweep_${var}_$val.err
  env $var=$val timeout 300 python tools/dedup_breakdown.py > $O/dedup_${var}_$val.json 2> $O/dedup_${var}_$val.err
  python - <<PY
import json
j=json.load(open("$O/sweep_${var}_$val.json"))
print("$var=$val", {k[7:]:(v["update_kv_us"], v["logits"]["us"], v["finalize"]["us"]) for k,v in j.items()})
j=json.load(open("$O/dedup_${var}_$val.json")); print("   gqa", {k:(v["update_kv_us"], v["logits"], v["finalize"]) for k,v in j.items()})
PY
done; done
