#!/bin/bash
# round 3: single-stage logits workgroups with LDS-direct K rows (libpkv.so) against the deferred-store build (libpkv_defer.so)
# Record of a finished experiment (results: profiles/r03/ab/).  libpkv_defer.so = the build of HEAD at the time (deferred stores, register-staged K rows); libpkv.so was the LDS-direct variant, since removed,
# copied next to libpkv.so before the session (PKV_LIB selects a build of the same ABI, pyramidkv_amd/_native.py).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/one
D=$R/pyramidkv_amd/libpkv_defer.so
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/one/pytest.txt 2>&1
cat gpurun_out/one/pytest.txt
bash tools/r03_ab.sh 1 "defer:PKV_LIB=$D" "one:" "one_w8192:PKV_LOGITS_V2_WGS=8192" "defer_w8192:PKV_LIB=$D,PKV_LOGITS_V2_WGS=8192" "one_w768:PKV_LOGITS_V2_WGS=768" "one2:" "defer2:PKV_LIB=$D" "one3:" 2>&1 | tee gpurun_out/one/ab.txt
