#!/bin/bash
# Round-5 first GPU session: the whole GPU suite on the new ABI / prepared calls / one-launch Ada-SnapKV budgets, then the
# host breakdown, the Ada bench and the driver's bench command.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=5 -x > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
timeout 300 python tools/host_breakdown.py > $O/host_breakdown.json 2> $O/host_breakdown.err
timeout 300 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ada -- python $R/tools/ada_bench.py > $O/prof_ada.log 2>&1)
tail -5 $O/pytest.txt; cat $O/host_breakdown.json; tail -3 $O/host_breakdown.err; head -c 1500 $O/ada_bench.json; tail -3 $O/ada_bench.err; tail -3 $O/bench.err
