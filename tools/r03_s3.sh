#!/bin/bash
# Round-3 session 3: the two-launch tail (select_parts + gather_merge): suite, then A/B against the three-launch tail.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s3
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
bash tools/r03_ab.sh 2 "fused:" "three:PKV_FUSED_TAIL=0" > $O/ab.txt 2>&1
grep -E "passed|failed" $O/pytest.txt | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.txt | head -30; cat $O/ab.txt
