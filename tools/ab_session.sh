#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
PKV_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
tail -1 $O/bench_n2_gloo.json | cut -c1-500; tail -3 $O/bench_n2_gloo.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
