#!/bin/bash
# quick iteration on the top-k kernel: top-k tests, phase trace, bench per-kernel times
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tk
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "topk or compress or golden or update_kv or pyramid" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so timeout 300 python tools/topk_trace3.py > $O/trace.json 2> $O/trace.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -2 $O/pytest.txt
python - <<PY
import json
j=json.load(open("$O/trace.json"))
for k,v in j.items(): print(k,v["stamps_rel"],v["C"],v["total"],v.get("s13"))
b=json.load(open("$O/bench.json")); print(b["kv_compress_ms_per_layer"], {k:v["avg_us"] for k,v in b["roofline_kernels"].items()})
PY
