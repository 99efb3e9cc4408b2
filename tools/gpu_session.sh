#!/bin/bash
# One GPU session: probe, parity tests, bench, rocprof kernel trace.  Run via gpurun from the repo root.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python tools/gpu_probe.py > $O/probe.json 2> $O/probe.err
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rocprof.log 2>&1
echo "rocprof exit $?" >> $O/rocprof.log
cd $R
tail -5 $O/pytest.txt; cat $O/bench.json; tail -3 $O/bench.err
