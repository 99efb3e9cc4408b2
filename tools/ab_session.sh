#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
echo "smoke exit $?" >> $O/smoke.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
tail -3 $O/smoke.txt; cat $O/bench.json | cut -c1-3000; tail -2 $O/bench.err
