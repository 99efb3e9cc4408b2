#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python tools/policy_bench.py > $O/policy_bench.json 2> $O/policy_bench.err
cat $O/policy_bench.json; tail -3 $O/policy_bench.err
