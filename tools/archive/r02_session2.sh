#!/bin/bash
# Round-2 GPU session 2: new small-k top-k path (algo 1) vs the old one (algo 0), gather rows-per-lane.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
for algo in 0 1; do
  PKV_TOPK_ALGO=$algo timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_algo$algo.json 2> $O/bench_algo$algo.err
  PKV_TOPK_ALGO=$algo PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so timeout 300 python tools/topk_trace3.py > $O/trace_algo$algo.json 2> $O/trace_algo$algo.err
done
for rpt in 0 8 16; do
  PKV_GATHER_RPT=$rpt timeout 300 python tools/sweep.py quick > $O/sweep_rpt${rpt}.json 2> $O/sweep_rpt${rpt}.err
done
tail -3 $O/pytest.txt
python - <<PY
import json
for a in (0,1):
    try:
        b=json.load(open("$O/bench_algo%d.json"%a)); print(a, b["kv_compress_ms_per_layer"], {k:v["avg_us"] for k,v in b["roofline_kernels"].items()})
    except Exception as e: print(a,"ERR",e)
PY
