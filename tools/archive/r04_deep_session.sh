#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_deep
rm -rf $O; mkdir -p $O
cd $R
PKV_LOGITS_DEEP=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -q --timeout 600 -m gpu -k "window_scores or snapkv or gqa or unexpanded or small or boundary" > $O/pytest_deep.txt 2>&1; echo "deep exit $?"; grep -E "passed|failed|^FAILED" $O/pytest_deep.txt | tail -5
for cfg in "0 0" "1 0" "1 512" "1 683" "0 512" "1 1536"; do
  set -- $cfg
  PKV_LOGITS_DEEP=$1 PKV_LOGITS_V2_WGS=$2 timeout 300 python bench.py --steps 10 --warmup 2 --only-gqa-extra --no-cpu-baseline --no-parity > $O/bench_deep_$1_$2.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/bench_deep_$1_$2.json").read().strip().splitlines()[-1])
rk = d["roofline_kernels"]
print("deep $1 wgs $2: headline ms/step", d["ms_per_step"], "logits", rk["logits"]["avg_us"], rk["logits"]["frac"], "| gqa: us/layer", d["extras"]["unexpanded_gqa_us_per_layer"], "logits_gqa4", rk["logits_gqa4"]["avg_us"], rk["logits_gqa4"]["frac"])
PY
done
