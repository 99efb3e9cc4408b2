#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ada_session
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_small_shapes.py -m gpu -q --timeout 900 -k "ada or headkv or windows_up_to or wide_gqa or small_shapes or golden" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
timeout 600 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
grep -E "passed|failed" $O/pytest.txt | tail -3; grep -E "^FAILED|^ERROR|Error" $O/pytest.txt | head -20; python - <<PY
import json
j=json.load(open("$O/ada_bench.json"))
for k,v in j.items(): print(k, v["update_kv_ms"], v["kernels_us"])
PY
