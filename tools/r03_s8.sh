#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s8
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
bash tools/r03_ab.sh 2 "new:" > $O/ab.txt 2>&1
grep -E "passed|failed" $O/pytest.txt | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.txt | head -20; cut -c1-200 $O/ab.txt
