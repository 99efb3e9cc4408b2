# merge-path session: parity tests of the LOOK-M merge, wall time, per-kernel split (old vs pipelined pivot kernel)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/merge; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "merge" --timeout 600 > $O/pytest_merge.txt 2>&1; echo "pytest exit $?" >> $O/pytest_merge.txt
timeout 300 python tools/merge_bench.py > $O/merge_bench.json 2> $O/merge_bench.err
for v in ${MERGE_VARIANTS:-"PKV_MERGE_PIVOT2=0" "PKV_MERGE_PF=1" "PKV_MERGE_PF=2" "PKV_MERGE_PF=2,PKV_MERGE_NST=4" "PKV_MERGE_PF=2,PKV_MERGE_NST=16" "PKV_MERGE_PF=2,PKV_MERGE_NT=0"}; do
for cfg in "32768 128" "8192 128"; do
  tag=$(echo "$v $cfg" | tr ' =' '__')
  (cd /tmp && env $(echo $v | tr "," " ") timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python $R/tools/merge_only.py $cfg > /dev/null 2>&1)
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $v  S, budget: $cfg"; [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "merge" in r["Name"]:
        print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/prof_$tag
done; done > $O/merge_kernel_split.txt 2>&1
