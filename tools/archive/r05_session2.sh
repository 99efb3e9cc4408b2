#!/bin/bash
# Round-5 second GPU session: Ada-SnapKV after the register-resident budget kernel + host fast path, launch-cost and top-k probes,
# the new fuzz / score-bar tests.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s2
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q --timeout 900 --durations=5 -x -k "fuzz or ada or window_scores or config3 or config5 or documented or one_million or head_sizes or index_out" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
cp gpurun_out/parity_fuzz_seed*.json $O/ 2>/dev/null
./tools/probes/launch_cost > $O/launch_cost.json 2> $O/launch_cost.err
timeout 300 python tools/host_breakdown.py > $O/host_breakdown.json 2> $O/host_breakdown.err
timeout 300 python tools/topk_k_probe.py > $O/topk_k_probe.json 2> $O/topk_k_probe.err
PKV_LIB=$R/pyramidkv_amd/libpkv_debug.so timeout 300 python tools/topk_k_probe.py > $O/topk_k_probe_debug.json 2> $O/topk_k_probe_debug.err
timeout 300 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ada -- python $R/tools/ada_bench.py > $O/prof_ada.log 2>&1)
tail -8 $O/pytest.txt; cat $O/launch_cost.json; cat $O/host_breakdown.json | tail -12; head -c 600 $O/ada_bench.json
