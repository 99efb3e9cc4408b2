#!/bin/bash
# One GPU session: parity tests, bandwidth probe, bench, rocprof kernel trace + PMC passes, sweep.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
./tools/bw_probe > $O/bw_probe.json 2> $O/bw_probe.err
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
timeout 900 python tools/sweep.py ${SWEEP:-} > $O/sweep.json 2> $O/sweep.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/rocprof.log 2>&1
echo "rocprof exit $?" >> $O/rocprof.log
if [ "${PMC:-1}" = "1" ]; then
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
python $R/tools/pmc_summary.py $O $O/pmc_traffic.json > $O/pmc_summary.log 2>&1
fi
cd $R
tail -5 $O/pytest.txt; cat $O/bench.json; tail -3 $O/bench.err
