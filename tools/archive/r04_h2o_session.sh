#!/bin/bash
# round 4, H2O session: correctness of the new kernels vs the oracle, A/B of the variant libraries, issue-port counters
#   bash tools/r04_h2o_session.sh [--pmc] variant ...
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_h2o
rm -rf $O; mkdir -p $O
cd $R
PMC=0
if [ "${1:-}" = "--pmc" ]; then PMC=1; shift; fi
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x -k "h2o" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
LIBS="tools/_h2o_base.so pyramidkv_amd/libpkv.so"
for v in "$@"; do LIBS="$LIBS tools/_h2o_$v.so"; done
timeout 900 python tools/h2o_ab.py 32768 $LIBS > $O/h2o_ab.txt 2>&1
cat $O/h2o_ab.txt
if [ $PMC = 1 ]; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h2o -- python $R/tools/h2o_only.py 32768 > $O/prof_h2o.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_h2o_a -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_a.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_h2o_b -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_b.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_h2o_c -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_c.log 2>&1
  cd $R
  python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_h2o_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pkv::h2o" not in n: continue
        acc[n.split("pkv::")[1].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    for f in glob.glob("$O/pmc_h2o_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if k in r["Kernel_Name"]:
                d["VGPR_Count"] = r.get("VGPR_Count"); d["Accum_VGPR_Count"] = r.get("Accum_VGPR_Count"); d["SGPR_Count"] = r.get("SGPR_Count")
                d["LDS_Block_Size"] = r.get("LDS_Block_Size"); d["Grid_Size"] = r.get("Grid_Size"); d["Workgroup_Size"] = r.get("Workgroup_Size")
                break
        else: continue
        break
json.dump(out, open("$O/pmc_h2o.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
  find $O/prof_h2o -name "*kernel_stats.csv" -exec cp {} $O/prof_h2o_kernel_stats.csv \;
  cat $O/prof_h2o_kernel_stats.csv | head -5
  rm -rf $O/prof_h2o $O/pmc_h2o_a $O/pmc_h2o_b $O/pmc_h2o_c
fi
