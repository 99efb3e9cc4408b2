#!/bin/bash
# Round-5 third GPU session: full suite on the sampled-reference H2O pass, H2O A/B (round-4 kernels / shipped / s_setprio
# variants; N(0,1), scaled and outlier data), the no-store K scan for the two-pass record, Ada-SnapKV one-launch vs three-launch.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s3
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=5 -x > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
cp gpurun_out/parity_fuzz_seed*.json gpurun_out/parity_report.json $O/ 2>/dev/null
{
for cfg in "32768 1 0" "8192 1 0" "4096 1 0" "32768 4 0" "32768 8 0" "32768 1 40" "8192 4 0" "8192 1 40"; do
  set -- $cfg
  echo "== S=$1 scale=$2 outlier=$3"
  H2O_SCALE=$2 H2O_OUTLIER=$( [ "$3" = "0" ] && echo "" || echo $3 ) timeout 600 python tools/h2o_ab.py $1 tools/_h2o_r04ref.so pyramidkv_amd/libpkv.so tools/_h2o_prio1.so tools/_h2o_prio2.so
done
} > $O/h2o_ab.txt 2>&1
{
for ab in 0 1; do for g in 1 4; do
  echo "== ABLATE=$ab GQA=$g"; PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so PKV_LOGITS_ABLATE=$ab GQA=$g CASES=1x32768,8x32768 timeout 300 python tools/logits_ablate.py
done; done
} > $O/logits_ablate_gqa.txt 2>&1
for f in 0 1; do
  PKV_ADA_FUSED=$f timeout 300 python tools/ada_bench.py > $O/ada_bench_fused$f.json 2> $O/ada_bench_fused$f.err
  PKV_ADA_FUSED=$f timeout 300 python tools/topk_k_probe.py > $O/topk_k_probe_fused$f.json 2> $O/topk_k_probe_fused$f.err
done
timeout 300 python tools/host_breakdown.py > $O/host_breakdown.json 2> $O/host_breakdown.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
tail -8 $O/pytest.txt; cat $O/h2o_ab.txt; cat $O/logits_ablate_gqa.txt; tail -3 $O/bench.err
