#!/bin/bash
# Round-2 GPU session 1: suite + new bench + gather shape A/B + RCCL nranks=1 + H2O PMC.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
# gather shape A/B (identical results, different work decomposition)
for rpt in 2 4 8; do for xcd in 0 1; do
  PKV_GATHER_RPT=$rpt PKV_GATHER_XCD=$xcd timeout 300 python tools/sweep.py quick > $O/sweep_rpt${rpt}_xcd${xcd}.json 2> $O/sweep_rpt${rpt}_xcd${xcd}.err
done; done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
# RCCL with nranks = 1: process group "nccl" + all_gather_into_tensor per layer
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_rccl_n1.json 2> $O/bench_rccl_n1.err
echo "rccl n1 exit $?" >> $O/bench_rccl_n1.err
# N=2 code path on one GPU through gloo (both legs)
PKV_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
echo "gloo n2 exit $?" >> $O/bench_n2_gloo.err
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h2o -- python $R/tools/h2o_only.py 32768 > $O/prof_h2o.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_h2o_a -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_h2o_b -- python $R/tools/h2o_only.py 32768 > $O/pmc_h2o_b.log 2>&1
cd $R
tail -3 $O/pytest.txt; head -c 600 $O/bench.json; tail -2 $O/bench.err; tail -2 $O/bench_rccl_n1.err
