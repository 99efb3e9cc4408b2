#!/bin/bash
# counters of the un-expanded-GQA K scan (logits2_kernel<BF16, 2, true>, C = 32 columns per K row)
R=$(pwd); O=$R/gpurun_out/gqa_pmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export GQA=4 CASES=1x32768
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/a -- python $R/tools/logits_ablate.py > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/b -- python $R/tools/logits_ablate.py > $O/b.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/c -- python $R/tools/logits_ablate.py > $O/c.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/d -- python $R/tools/logits_ablate.py > $O/d.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/gqa_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'logits2' in n or 'finalize' in n: acc[n.split('pkv::')[1][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, {c: '%.4g' % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
tail -2 gpurun_out/gqa_pmc/*.log | cut -c1-300
