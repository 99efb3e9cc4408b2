export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/full4
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_gqa_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only-gqa-extra --no-parity > $O/pmc_gqa_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_gqa_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --only-gqa-extra --no-parity > $O/pmc_gqa_write.log 2>&1
tail -2 $O/pmc_gqa_fetch.log | cut -c1-300
find $O/pmc_gqa_fetch -name "*counter_collection.csv" | head
