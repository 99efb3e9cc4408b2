#!/bin/bash
# Round-5 seventh GPU session: the single-wave finish of the budget kernels (suite subset + Ada bench + stamps), two more fuzz seeds.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s7
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_f32.py -m gpu -q --timeout 900 -x -k "fuzz or ada or config5 or headkv or flat" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so timeout 300 python tools/topk_k_probe.py > $O/topk_k_probe_debug.json 2> $O/topk_k_probe_debug.err
timeout 300 python tools/host_breakdown.py > $O/host_breakdown.json 2> $O/host_breakdown.err
timeout 300 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ada -- python $R/tools/ada_bench.py > $O/prof_ada.log 2>&1)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
for seed in 31337 99; do timeout 200 python tools/parity_fuzz.py 90 $seed 5000 > $O/parity_fuzz_$seed.txt 2>&1; done
tail -4 $O/pytest.txt; grep -A9 budget_kernel_stamps $O/topk_k_probe_debug.json; tail -8 $O/host_breakdown.json; tail -2 $O/bench.err; tail -1 $O/parity_fuzz_*.txt
