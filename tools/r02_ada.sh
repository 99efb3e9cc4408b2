#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ada
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600  > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
timeout 600 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
tail -15 $O/pytest.txt; cat $O/ada_bench.json; tail -3 $O/ada_bench.err
