#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/f32_session
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f32.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -k "f32 or ada or headkv" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
grep -E "passed|failed" $O/pytest.txt | tail -2; grep -E "^FAILED|^ERROR|^E  " $O/pytest.txt | head -30
