#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_gqa
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f32.py -q --timeout 600 -m gpu > $O/pytest_f32.txt 2>&1; echo "f32 exit $?"; grep -E "passed|failed|^FAILED|Error" $O/pytest_f32.txt | tail -8
PKV_LOGITS_PERSIST=3 timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -m gpu -k "window_scores or snapkv or gqa or unexpanded" > $O/pytest_persist.txt 2>&1; echo "persist exit $?"; grep -E "passed|failed|^FAILED" $O/pytest_persist.txt | tail -5
for pv in 0 2 3 4 6; do
  PKV_LOGITS_PERSIST=$pv timeout 300 python bench.py --steps 10 --warmup 2 --only-gqa-extra --no-cpu-baseline --no-parity > $O/bench_persist_$pv.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/bench_persist_$pv.json").read().strip().splitlines()[-1])
rk = d["roofline_kernels"]
print("persist $pv: headline ms/step", d["ms_per_step"], "logits", rk["logits"]["avg_us"], "| gqa: us/layer", d["extras"]["unexpanded_gqa_us_per_layer"], "logits_gqa4", rk["logits_gqa4"]["avg_us"], rk["logits_gqa4"]["frac"])
PY
done
