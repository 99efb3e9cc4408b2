#!/bin/bash
# Round-5 sixth GPU session: Ada-SnapKV host trims + phase stamps of the one-launch budget kernel + the longer fuzz.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s6
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --timeout 900 -x -k "fuzz or ada or config5 or headkv or flat" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so timeout 300 python tools/topk_k_probe.py > $O/topk_k_probe_debug.json 2> $O/topk_k_probe_debug.err
timeout 300 python tools/host_breakdown.py > $O/host_breakdown.json 2> $O/host_breakdown.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
tail -4 $O/pytest.txt; grep -A9 budget_kernel_stamps $O/topk_k_probe_debug.json; tail -8 $O/host_breakdown.json; tail -2 $O/bench.err
