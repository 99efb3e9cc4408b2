#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pt
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 --durations=8 ${PYTEST_ARGS:-} > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
tail -40 $O/pytest.txt
