#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
PKV_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
echo "exit $?" >> $O/bench_n2_gloo.err
cat $O/bench_n2_gloo.json | cut -c1-900; tail -5 $O/bench_n2_gloo.err
