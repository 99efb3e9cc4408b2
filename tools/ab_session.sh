#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
: > $O/ablate.jsonl
export CASES=1x32768,2x32768,8x32768,1x8192,8x8192
run() { env "$@" timeout 300 python tools/logits_ablate.py >> $O/ablate.jsonl 2>> $O/ablate.err; }
run PKV_LOGITS_V2=0
for w in 768 1024 1280 1536 2048 3072 4096; do run PKV_LOGITS_V2_WGS=$w; done
run PKV_LOGITS_V2=0
cat $O/ablate.jsonl
