#!/bin/bash
# Round-3 session 1: full GPU suite on the new kernels, headline bench, knob A/B, self-launched N = 2 (gloo), parity sweep.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=8 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
bash tools/r03_ab.sh 2 "new:" "old:PKV_LOGITS_FEXP=0,PKV_LOGITS_ST=0,PKV_FIN_PRE=0,PKV_FUSE_GATHER=0" "nofexp:PKV_LOGITS_FEXP=0" "st0:PKV_LOGITS_ST=0" "st1:PKV_LOGITS_ST=1" "nopre:PKV_FIN_PRE=0" "nofuse:PKV_FUSE_GATHER=0" > $O/ab.txt 2>&1
PKV_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n2_gloo_selflaunch.json 2> $O/bench_n2_gloo_selflaunch.err
echo "n2 exit $?" >> $O/bench_n2_gloo_selflaunch.err
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n2_nccl_refused.json 2> $O/bench_n2_nccl_refused.err
echo "n2 nccl exit $? (expected 2: one GPU visible)" >> $O/bench_n2_nccl_refused.err
timeout 900 python tools/parity_sweep.py > $O/parity_sweep.log 2>&1
cp gpurun_out/parity_sweep.json $O/ 2>/dev/null
tail -5 $O/pytest.txt; cat $O/ab.txt; head -c 600 $O/bench.json; echo; tail -3 $O/bench_n2_gloo_selflaunch.err; tail -2 $O/bench_n2_nccl_refused.err; tail -2 $O/parity_sweep.log
