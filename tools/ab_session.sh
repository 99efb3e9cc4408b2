#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
python tools/topk_trace.py > $O/topk_trace.json 2> $O/topk_trace.err
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rocprof.log 2>&1
cd $R
tail -4 $O/pytest.txt
