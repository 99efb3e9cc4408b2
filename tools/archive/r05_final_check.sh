#!/bin/bash
# Round 5, last GPU session: the whole suite and the driver's bench command on the final host code (kernel sources unchanged
# since tools/r05_full_session.sh; its other artefacts stay valid), results into gpurun_out/full5 for tools/collect_r05.py.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/full5
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=5 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
timeout 600 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
timeout 600 python tools/policy_bench.py > $O/policy_bench.json 2> $O/policy_bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ada -- python $R/tools/ada_bench.py > $O/prof_ada.log 2>&1)
timeout 300 python tools/host_breakdown.py > $O/host_breakdown.json 2> $O/host_breakdown.err
python $R/__graft_entry__.py smoke > $O/smoke.log 2>&1
echo "smoke exit $?" >> $O/smoke.log
tail -6 $O/pytest.txt; tail -2 $O/bench.err; tail -2 $O/smoke.log
