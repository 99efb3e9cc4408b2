#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_h2o3
rm -rf $O; mkdir -p $O
cd $R
rocm-smi --showpower --showclocks --json | head -c 1500; echo
python tools/h2o_dbg2.py tools/_h2o_base.so pyramidkv_amd/libpkv.so tools/_h2o_nomm.so > $O/dbg2.txt 2>&1
cat $O/dbg2.txt
LIBS="tools/_h2o_base.so pyramidkv_amd/libpkv.so"
for v in "$@"; do LIBS="$LIBS tools/_h2o_$v.so"; done
timeout 900 python tools/h2o_power.py $LIBS > $O/h2o_power.txt 2>&1
cat $O/h2o_power.txt
