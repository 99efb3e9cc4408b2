#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/it
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "topk or compress or golden or config3_snapkv or ada or full_size" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
timeout 300 python tools/sweep.py quick > $O/sweep.json 2> $O/sweep.err
timeout 300 python tools/ada_bench.py > $O/ada.json 2> $O/ada.err
tail -3 $O/pytest.txt
python - <<PY
import json
j=json.load(open("$O/sweep.json"))
for k,v in j.items(): print(k, v["update_kv_us"], {n:v[n]["us"] for n in ("logits","finalize","topk","gather")})
j=json.load(open("$O/ada.json"))
for k,v in j.items(): print(k, v["update_kv_ms"], v["kernels_us"])
PY
