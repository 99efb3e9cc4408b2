#!/bin/bash
# A/B of two builds of libpkv on the headline bench and the un-expanded-GQA breakdown: bash tools/lib_ab.sh a.so b.so
R=$(pwd); O=$R/gpurun_out/ab; mkdir -p $O
for rep in 1 2; do for lib in "$@"; do
  n=$(basename $lib .so)
  PKV_LIB=$R/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err
  PKV_LIB=$R/$lib timeout 300 python tools/dedup_breakdown.py > $O/dedup_$n.json 2> $O/dedup_$n.err
  python - <<PY
import json
b=json.load(open("$O/bench_$n.json")); print("$n", b["kv_compress_ms_per_layer"], {k:v["avg_us"] for k,v in b["roofline_kernels"].items()})
j=json.load(open("$O/dedup_$n.json")); print("   gqa", {k:(v["update_kv_us"], v["logits"]) for k,v in j.items()})
PY
done; done
