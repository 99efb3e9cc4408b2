#!/bin/bash
# Round-5 tenth GPU session: the prepared-call invalidation test; bench.py with no flags (the driver's N = 1 form), wall time.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s10
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "prepared or index_out or graph_capturable or documented" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
/usr/bin/time -v timeout 900 python bench.py > $O/bench_default_flags.json 2> $O/bench_default_flags.err
tail -5 $O/pytest.txt; grep -E "Elapsed|Maximum resident" $O/bench_default_flags.err; head -c 300 $O/bench_default_flags.json
