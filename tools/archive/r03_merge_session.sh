#!/bin/bash
# round 3: LOOK-M merge with the bucket + bitmap-ordered scatter and head sizes 64 / 256
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/merge
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "merge or head_sizes" 2>&1 | tail -8 | tee gpurun_out/merge/pytest.txt
timeout 300 python tools/merge_bench.py > gpurun_out/merge/merge_bench.json 2> gpurun_out/merge/merge_bench.err; cat gpurun_out/merge/merge_bench.json
for cfg in "32768 128" "32768 2048" "8192 2048"; do
  tag=$(echo $cfg | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/merge/prof_$tag -- python $R/tools/merge_only.py $cfg > /dev/null 2>&1)
  f=$(find $R/gpurun_out/merge/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "merge" in r["Name"]:
        print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done 2>&1 | tee gpurun_out/merge/kernel_split.txt
