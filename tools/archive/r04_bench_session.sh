#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_bench
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench exit $?"; tail -3 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "scaling", "parity")})
print("roofline", d["roofline"])
for r in d.get("grid", []): print({k: (v if not isinstance(v, dict) else (v["us"], v["frac"])) for k, v in r.items()})
print("extras", d.get("extras"))
for r in d.get("sweep", []): print(r)
print("north", {k: (v["avg_us"], v["frac"]) for k, v in d["roofline_kernels"].items() if v})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("call_effective"))
PY
PKV_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_n8_gloo_one_gpu.json 2> $O/bench_n8.err
echo "bench n8 exit $?"; tail -3 $O/bench_n8.err
python - <<PY
import json
d = json.loads(open("$O/bench_n8_gloo_one_gpu.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "scaling", "n_gpus", "parity", "config", "scaling_legs", "allgather_us", "rccl_nranks")})
PY
