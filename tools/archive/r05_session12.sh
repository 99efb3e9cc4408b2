#!/bin/bash
# Round-5 twelfth GPU session: HeadKV prepared path; final bench line on the final host code.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s12
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f32.py tests/test_monkeypatch_plumbing.py -m gpu -q --timeout 900 -x -k "headkv or prepared or flat or replace_llama" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
timeout 300 python tools/ada_bench.py > $O/ada_bench.json 2> $O/ada_bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench exit $?" >> $O/bench.err
tail -4 $O/pytest.txt; grep -A9 "headkv_S32768" $O/ada_bench.json; tail -2 $O/bench.err
