#!/bin/bash
# Round 5, last GPU session (second take: the first one saw examples/host_cabi die in process teardown once).
# 1. host_cabi under load: 16 runs each of the previous and the current example while a Python process keeps the GPU busy
#    (the condition of the one failure: run from inside the test suite); 2. the whole suite, the driver's bench command,
#    the Ada-SnapKV / policy benches, rocprof of the Ada bench, host breakdown, smoke -> gpurun_out/full5 (tools/collect_r05.py).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/full5
mkdir -p $O
cd $R
setsid bash -c 'while true; do timeout 120 python tools/policy_bench.py > /dev/null 2>&1; done' &
LOADER=$!   # its own process group: stopped below by that group id
sleep 20
: > $O/host_cabi_under_load.txt
for exe in examples/host_cabi_prev examples/host_cabi; do
  [ -x $exe ] || continue
  fails=0
  for i in $(seq 1 16); do
    timeout 120 $exe > /tmp/hc.out 2> /tmp/hc.err; rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "--- $exe run $i rc=$rc" >> $O/host_cabi_under_load.txt; tail -2 /tmp/hc.out >> $O/host_cabi_under_load.txt; tail -4 /tmp/hc.err >> $O/host_cabi_under_load.txt; fi
  done
  echo "$exe under load: $fails failures of 16" >> $O/host_cabi_under_load.txt
done
kill -- -$LOADER 2>/dev/null; wait $LOADER 2>/dev/null
sleep 3
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
cat $O/host_cabi_under_load.txt; tail -6 $O/pytest.txt; tail -2 $O/bench.err; tail -2 $O/smoke.log
