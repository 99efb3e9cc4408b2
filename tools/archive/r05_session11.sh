#!/bin/bash
# Round-5 eleventh GPU session: large-budget Ada-SnapKV through short lists (2 x base) - tests, Ada bench, bench line.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s11
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_f32.py tests/test_gpu_small_shapes.py -m gpu -q --timeout 900 -x -k "fuzz or ada or config5 or headkv or flat or prepared or no_writes" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
timeout 300 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
tail -4 $O/pytest.txt; grep -A8 "S32768_cap2048_unexpanded" $O/ada_bench.json; tail -2 $O/bench.err
