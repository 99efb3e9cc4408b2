#!/bin/bash
# Round-3 session 2: suite after the fexp fix, finalize load-order A/B, parity of the hardware-exp statistics, N = 2 self-launch.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s2
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
bash tools/r03_ab.sh 2 "pre1:PKV_FIN_PRE=1" "pre0:PKV_FIN_PRE=0" "pre0_nofexp:PKV_FIN_PRE=0,PKV_LOGITS_FEXP=0" > $O/ab.txt 2>&1
PKV_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n2_gloo_selflaunch.json 2> $O/bench_n2_gloo_selflaunch.err
echo "n2 exit $?" >> $O/bench_n2_gloo_selflaunch.err
PKV_LOGITS_FEXP=1 timeout 600 python tools/parity_sweep.py > $O/parity_sweep_fexp1.log 2>&1; cp gpurun_out/parity_sweep.json $O/parity_sweep_fexp1.json
PKV_LOGITS_FEXP=0 timeout 600 python tools/parity_sweep.py > $O/parity_sweep_fexp0.log 2>&1; cp gpurun_out/parity_sweep.json $O/parity_sweep_fexp0.json
tail -3 $O/pytest.txt; cat $O/ab.txt; tail -2 $O/bench_n2_gloo_selflaunch.err; head -c 300 $O/bench_n2_gloo_selflaunch.json; echo; tail -1 $O/parity_sweep_fexp1.log; tail -1 $O/parity_sweep_fexp0.log
