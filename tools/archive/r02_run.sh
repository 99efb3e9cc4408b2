#!/bin/bash
# generic: run a pytest selection on the GPU box; PYTEST_ARGS in env
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/run
mkdir -p $O
cd $R
timeout 1500 python -m pytest -m gpu -q --timeout 900 "$@" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
tail -60 $O/pytest.txt
