#!/bin/bash
# round 4: the shipped H2O kernels - tests, power / clock next to round 3's kernels, issue-port counters, rocprofv3 stats
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_h2o_final
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
grep -E "passed|failed|^FAILED|^ERROR|pytest exit" $O/pytest.txt | tail -8
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
timeout 600 python tools/h2o_power.py tools/_h2o_base.so tools/_h2o_wide.so pyramidkv_amd/libpkv.so > $O/h2o_power.txt 2>&1   # r03 / 32x32x16 pipeline (tools/build_h2o_variants.sh --src tools/probes/h2o_wide_pipeline.hip wide="") / shipped
grep -v amdgpu.ids $O/h2o_power.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h2o -- python $R/tools/h2o_only.py 32768 > $O/prof_h2o.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_h2o_a -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_h2o_b -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_h2o_c -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_c.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_h2o_d -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_d.log 2>&1
cd $R
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob("$O/pmc_h2o_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pkv::h2o" not in n: continue
        k = n.split("pkv::")[1].split("<")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {c: r.get(c) for c in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
out = {k: dict({c: sum(v) / len(v) for c, v in d.items()}, **meta[k]) for k, d in acc.items()}
json.dump(out, open("$O/pmc_h2o.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O/prof_h2o -name "*kernel_stats.csv" -exec cp {} $O/prof_h2o_kernel_stats.csv \;
cut -c1-150 $O/prof_h2o_kernel_stats.csv | head -5
rm -rf $O/prof_h2o $O/pmc_h2o_a $O/pmc_h2o_b $O/pmc_h2o_c $O/pmc_h2o_d
