#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/rocprof.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
python $R/tools/pmc_summary.py $O $O/pmc_traffic.json | head -30
