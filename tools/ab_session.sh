#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python tools/topk_trace2.py > $O/topk_trace2.json 2> $O/topk_trace2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/topk_trace2.json'))
for k,v in d.items(): print(k, v)
PY
