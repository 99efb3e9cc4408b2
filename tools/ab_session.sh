#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/ablate.jsonl
run() { env "$@" timeout 300 python tools/logits_ablate.py >> $O/ablate.jsonl 2>> $O/ablate.err; }
run PKV_LOGITS_V2=1
run PKV_LOGITS_V2=1 PKV_LOGITS_ABLATE=1
run PKV_LOGITS_V2=1 PKV_LOGITS_ABLATE=3
run PKV_LOGITS_V2=1 PKV_LOGITS_ABLATE=4
run PKV_LOGITS_V2=1
python - <<'PY'
import json
for l in open('gpurun_out/ablate.jsonl'):
    d=json.loads(l); print({k:d[k] for k in ('ablate','nt','rm','tile','v2')}, d['B1']['logits_us'], d['B8']['logits_us'], d['B1']['finalize_us'], d['B8']['finalize_us'])
PY
