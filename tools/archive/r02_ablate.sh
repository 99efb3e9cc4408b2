#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so CASES=1x32768
for g in 4 1; do for ab in 0 1 2 3 4; do
  echo -n "GQA=$g ablate=$ab: "; GQA=$g PKV_LOGITS_ABLATE=$ab timeout 300 python tools/logits_ablate.py 2>/dev/null | tail -1
done; done
