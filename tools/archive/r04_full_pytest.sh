#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_pytest
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 "$@" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
grep -E "passed|failed|^FAILED|^ERROR|pytest exit" $O/pytest.txt | tail -15
