#!/bin/bash
# Round-6 full GPU session: everything profiles/r06 is built from (tools/collect_r06.py copies the judged artefacts).
#   SKIP_PYTEST=1  the suite ran in its own gpurun call      SKIP_H2O=1  no H2O profile / counter passes
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/full6
rm -rf $O; mkdir -p $O
cd $R
if [ -z "${SKIP_PYTEST:-}" ]; then
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=5 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
fi
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
./tools/bw_probe > $O/bw_probe.json 2> $O/bw_probe.err
# HBM traffic counters first: the bench line attaches them when they belong to the current kernel sources
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-parity > $O/pmc_fetch.log 2>&1 ; timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-parity > $O/pmc_write.log 2>&1 ; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_gather_fetch -- python $R/tools/gather_pmc.py > $O/pmc_gather_fetch.log 2>&1 ; timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_gather_write -- python $R/tools/gather_pmc.py > $O/pmc_gather_write.log 2>&1 ; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_gqa_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only-gqa-extra --no-parity > $O/pmc_gqa_fetch.log 2>&1 ; timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_gqa_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only-gqa-extra --no-parity > $O/pmc_gqa_write.log 2>&1 ; python $R/tools/pmc_summary.py $O $O/pmc_traffic.json > $O/pmc_summary.log 2>&1)
mkdir -p $R/profiles/r06; cp $O/pmc_traffic.json $R/profiles/r06/pmc_traffic.json 2>/dev/null
# issue-port counters of the four headline kernels (what DESIGN.md's "vector-issue-bound" statements rest on)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_issue -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-parity > $O/pmc_issue.log 2>&1 ; timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_issue2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-parity > $O/pmc_issue2.log 2>&1)
# the driver's command
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
timeout 600 python tools/policy_bench.py > $O/policy_bench.json 2> $O/policy_bench.err
timeout 600 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ada -- python $R/tools/ada_bench.py > $O/prof_ada.log 2>&1)
timeout 300 python tools/host_breakdown.py > $O/host_breakdown.json 2> $O/host_breakdown.err
./tools/probes/launch_cost > $O/launch_cost.json 2> $O/launch_cost.err
timeout 900 python tools/parity_sweep.py > $O/parity_sweep.log 2>&1; cp gpurun_out/parity_sweep.json $O/parity_sweep.json 2>/dev/null
timeout 600 python tools/parity_three_way.py > $O/parity_three_way.log 2>&1; cp gpurun_out/parity_three_way.json $O/parity_three_way.json 2>/dev/null
timeout 600 python tools/scale_projection.py > $O/scale_projection.log 2>&1; cp gpurun_out/scale_projection.json $O/scale_projection.json 2>/dev/null
# RCCL with nranks = 1 (process group nccl, one all-gather per prefill) and the self-launched N = 2 code path through gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_rccl_n1.json 2> $O/bench_rccl_n1.log
PKV_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n2_gloo_selflaunch.json 2> $O/bench_n2_gloo_selflaunch.log
PKV_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_n8_gloo_one_gpu.json 2> $O/bench_n8_gloo_one_gpu.log
timeout 120 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2_nccl_one_gpu.json 2> $O/bench_n2_nccl_one_gpu.log; echo "exit $? (2 = refused: one GPU visible, no silent gloo)" >> $O/bench_n2_nccl_one_gpu.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-parity > $O/rocprof_headline.log 2>&1
echo "rocprof headline exit $?" >> $O/rocprof_headline.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof.log 2>&1
echo "rocprof exit $?" >> $O/rocprof.log
python $R/__graft_entry__.py smoke > $O/smoke.log 2>&1
echo "smoke exit $?" >> $O/smoke.log
if [ -z "${SKIP_H2O:-}" ]; then
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h2o -- python $R/tools/h2o_only.py 32768 > $O/prof_h2o.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_h2o_a -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_h2o_c -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_c.log 2>&1
fi
cd $R
# LOOK-M merge: wall time per call and the per-kernel split
timeout 300 python tools/merge_bench.py > $O/merge_bench.json 2> $O/merge_bench.err
for cfg in "32768 128" "32768 2048" "8192 2048"; do
  tag=$(echo $cfg | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_merge_$tag -- python $R/tools/merge_only.py $cfg > /dev/null 2>&1)
  f=$(find $O/prof_merge_$tag -name "*kernel_stats.csv" | head -1)
  echo "== S, budget: $cfg"; [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "merge" in r["Name"]:
        print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done > $O/merge_kernel_split.txt 2>&1
timeout 200 python tools/soak.py 60 > $O/soak.txt 2>&1; echo "soak exit $?" >> $O/soak.txt
tail -4 $O/pytest.txt 2>/dev/null; tail -2 $O/soak.txt; head -c 400 $O/bench.json; echo; tail -2 $O/bench.err; tail -1 $O/parity_sweep.log
