#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/mg
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x --timeout 600 -k merge > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
tail -3 $O/pytest.txt
bash tools/r02_prof.sh tools/merge_bench.py | grep -E "merge|exit"
