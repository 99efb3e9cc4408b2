#!/bin/bash
# rocprofv3 --kernel-trace --stats of one tool: bash tools/r02_prof.sh <script.py> [args]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof1
rm -rf $O; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python $R/$1 ${@:2} > $O/out.log 2>&1
echo "exit $?"
python - <<PY
import csv,glob
f=sorted(glob.glob("$O/p/**/*kernel_stats.csv",recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s %6s calls avg %9.2f us  min %9.2f  max %9.2f  %s%%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, r["Percentage"]))
PY
