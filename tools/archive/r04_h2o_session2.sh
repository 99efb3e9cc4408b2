#!/bin/bash
# round 4, H2O session: where does the time go?  ablation (no stream behind the loop), L2 / fabric counters, one failing case
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_h2o2
rm -rf $O; mkdir -p $O
cd $R
python tools/h2o_dbg2.py tools/_h2o_base.so pyramidkv_amd/libpkv.so tools/_h2o_nomm.so > $O/dbg2.txt 2>&1
cat $O/dbg2.txt
LIBS="tools/_h2o_base.so pyramidkv_amd/libpkv.so"
for v in "$@"; do LIBS="$LIBS tools/_h2o_$v.so"; done
timeout 900 python tools/h2o_ab.py 32768 $LIBS > $O/h2o_ab.txt 2>&1
cat $O/h2o_ab.txt
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_a -- python $R/tools/h2o_only.py 32768 > $O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_b -- python $R/tools/h2o_only.py 32768 > $O/pmc_b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_c -- python $R/tools/h2o_only.py 32768 > $O/pmc_c.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_d -- python $R/tools/h2o_only.py 32768 > $O/pmc_d.log 2>&1
cd $R
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pkv::h2o" not in n: continue
        acc[n.split("pkv::")[1].split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$O/pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -3 $O/pmc_a.log
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d
