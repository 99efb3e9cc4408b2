#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ada
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "ada or headkv or golden or config5 or flat" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
PKV_HOST_POLL=0 timeout 600 python tools/ada_bench.py > $O/ada_bench_sync.json 2> $O/ada_bench_sync.err
timeout 600 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
tail -5 $O/pytest.txt; tail -3 $O/ada_bench.err
python - <<PY
import json
for f in ("ada_bench_sync","ada_bench"):
    j=json.load(open("$O/%s.json"%f))
    for k,v in j.items():
        if "32768" in k: print(f,k,v["update_kv_ms"],v["kernels_us"])
PY
