#!/bin/bash
# Round 5, closing GPU session after the one host-side edit of csrc/pkv_coll.hip (RCCL resolution order): the HBM traffic
# counters again (profiles/r05/pmc_traffic.json is keyed to the kernel sources' hash), the driver's bench command, the whole
# suite, smoke -> gpurun_out/full5 (tools/collect_r05.py).  Kernels themselves are unchanged since tools/r05_full_session.sh.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/full5
mkdir -p $O
cd $R
(cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-parity > $O/pmc_fetch.log 2>&1 ; timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-parity > $O/pmc_write.log 2>&1 ; timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_gather_fetch -- python $R/tools/gather_pmc.py > $O/pmc_gather_fetch.log 2>&1 ; timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_gather_write -- python $R/tools/gather_pmc.py > $O/pmc_gather_write.log 2>&1 ; timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_gqa_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only-gqa-extra --no-parity > $O/pmc_gqa_fetch.log 2>&1 ; timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_gqa_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only-gqa-extra --no-parity > $O/pmc_gqa_write.log 2>&1 ; python $R/tools/pmc_summary.py $O $O/pmc_traffic.json > $O/pmc_summary.log 2>&1)
mkdir -p $R/profiles/r05; cp $O/pmc_traffic.json $R/profiles/r05/pmc_traffic.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
python $R/__graft_entry__.py smoke > $O/smoke.log 2>&1
echo "smoke exit $?" >> $O/smoke.log
timeout 460 python -m pytest tests -m gpu -q --timeout 300 --durations=5 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
date > $O/final_check3.done
tail -3 $O/pmc_summary.log; head -c 300 $O/bench.json; echo; tail -1 $O/bench.err; tail -1 $O/smoke.log; tail -4 $O/pytest.txt
