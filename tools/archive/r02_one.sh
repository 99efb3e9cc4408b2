#!/bin/bash
# run one tool on the GPU box: bash tools/r02_one.sh <script.py> [args]; stdout -> gpurun_out/one/<script>.json
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/one
mkdir -p $O
cd $R
n=$(basename $1 .py)
timeout 1200 python "$@" > $O/$n.json 2> $O/$n.err
echo "exit $?"; tail -3 $O/$n.err; cat $O/$n.json | head -80
