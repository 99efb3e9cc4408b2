#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
python tools/wg_trace.py > $O/wg_trace.json 2> $O/wg_trace.err
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --seq 8192 > $O/bench8k.json 2> $O/bench8k.err
PKV_FUSE_GATHER_ROWS=0 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_nofuse.json 2> $O/bench_nofuse.err
PKV_FUSE_GATHER_ROWS=0 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --seq 8192 > $O/bench8k_nofuse.json 2> $O/bench8k_nofuse.err
tail -3 $O/pytest.txt
