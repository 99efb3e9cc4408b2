#!/bin/bash
# Round-2 full GPU session: everything profiles/r02 is built from (tools/collect_r02.py copies the judged artefacts).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/full
rm -rf $O; mkdir -p $O
cd $R
if [ -z "${SKIP_PYTEST:-}" ]; then   # SKIP_PYTEST=1: the suite ran in its own gpurun call (gpurun_out/full/pytest.txt is kept)
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=5 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
fi
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
./tools/bw_probe > $O/bw_probe.json 2> $O/bw_probe.err
# HBM traffic counters first: the bench line attaches them when they belong to the current kernel sources
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1 ; timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1 ; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_gather_fetch -- python $R/tools/gather_pmc.py > $O/pmc_gather_fetch.log 2>&1 ; timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_gather_write -- python $R/tools/gather_pmc.py > $O/pmc_gather_write.log 2>&1 ; python $R/tools/pmc_summary.py $O $O/pmc_traffic.json > $O/pmc_summary.log 2>&1)
cp $O/pmc_traffic.json $R/profiles/r02/pmc_traffic.json 2>/dev/null
# the driver's command
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
timeout 600 python tools/sweep.py > $O/sweep.json 2> $O/sweep.err
timeout 600 python tools/policy_bench.py > $O/policy_bench.json 2> $O/policy_bench.err
timeout 600 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so timeout 300 python tools/topk_trace3.py > $O/topk_trace.json 2> $O/topk_trace.err
# RCCL with nranks = 1 (process group nccl, one all-gather per prefill and one per layer) and the N = 2 code path through gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_rccl_n1.json 2> $O/bench_rccl_n1.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --allgather layer > $O/bench_rccl_n1_perlayer.json 2> $O/bench_rccl_n1_perlayer.log
PKV_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.log
cd /tmp
# rocprofv3 of the bench command (kernel trace + stats): headline workload alone, then the full line (grid, baselines), then
# PMC traffic in separate passes
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/rocprof_headline.log 2>&1
echo "rocprof headline exit $?" >> $O/rocprof_headline.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof.log 2>&1
echo "rocprof exit $?" >> $O/rocprof.log
python $R/__graft_entry__.py smoke > $O/smoke.log 2>&1
echo "smoke exit $?" >> $O/smoke.log
# H2O: kernel stats + issue-port counters (two SQ passes)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h2o -- python $R/tools/h2o_only.py 32768 > $O/prof_h2o.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_h2o_a -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_h2o_b -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_b.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_h2o_c -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_c.log 2>&1
# the H2O kernels of the parent commit d56ac47 (tools/_h2o_base.so, built by hand) against the current ones, same box
(cd $R && bash tools/r02_h2o_ab.sh) > $O/h2o_ab.log 2>&1
cp $R/gpurun_out/h2o_ab.txt $O/h2o_ab.txt 2>/dev/null
cd $R
tail -4 $O/pytest.txt; head -c 400 $O/bench.json; echo; tail -2 $O/bench.err
