#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s5
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -q --timeout 900 -x > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so python tools/select_trace.py > $O/select_trace.json 2> $O/select_trace.err
bash tools/r03_ab.sh 2 "fused:" "three:PKV_FUSED_TAIL=0" > $O/ab.txt 2>&1
grep -E "passed|failed" $O/pytest.txt | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.txt | head; cat $O/ab.txt; python - <<PY
import json
j=json.load(open("$O/select_trace.json"))
for k,v in j.items(): print(k, v["select_total"], v["select_cycles"], v["gather_total"], v["gather_merge_cycles"], "C", v["C"])
PY
