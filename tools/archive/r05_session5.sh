#!/bin/bash
# Round-5 fifth GPU session: selection / Ada-SnapKV tests on the 16-byte in-bucket ranking and 16-byte list loads, then the benches.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s5
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 --durations=5 -x --deselect tests/test_gpu_dist.py::test_config4_world8_one_gpu -k "not h2o" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
timeout 300 python tools/topk_k_probe.py > $O/topk_k_probe.json 2> $O/topk_k_probe.err
PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so timeout 300 python tools/topk_k_probe.py > $O/topk_k_probe_debug.json 2> $O/topk_k_probe_debug.err
timeout 300 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
tail -6 $O/pytest.txt; cat $O/topk_k_probe.json | head -40; tail -3 $O/bench.err
