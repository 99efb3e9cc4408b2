#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for w in 256 512 768 1024 2048; do
PKV_LOGITS_V2_WGS=$w timeout 600 python tools/dedup_breakdown.py > $O/dedup_w$w.json 2>> $O/dedup.err
python - <<PY
import json
d=json.load(open("gpurun_out/dedup_w$w.json")); print("wgs $w", d["k120"])
PY
done
