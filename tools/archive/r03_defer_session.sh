#!/bin/bash
# round 3: logits stores deferred by one stage (libpkv.so) against the previous build (libpkv_base.so), over stages per workgroup
# Record of a finished experiment (results: profiles/r03/ab/).  libpkv_base.so = the build of the commit before 'logits2: the logits of a stage leave one stage later',
# copied next to libpkv.so before the session (PKV_LIB selects a build of the same ABI, pyramidkv_amd/_native.py).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/defer
B=$R/pyramidkv_amd/libpkv_base.so
( timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py -m gpu -x -q 2>&1 | tail -3
  PKV_LOGITS_V2_WGS=16 timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_shapes.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/defer/pytest.txt 2>&1
cat gpurun_out/defer/pytest.txt
bash tools/r03_ab.sh 1 "base:PKV_LIB=$B" "new:" "new_w4096:PKV_LOGITS_V2_WGS=4096" "base_w4096:PKV_LIB=$B,PKV_LOGITS_V2_WGS=4096" "new_w2048:PKV_LOGITS_V2_WGS=2048" "new_w1024:PKV_LOGITS_V2_WGS=1024" "base_w1024:PKV_LIB=$B,PKV_LOGITS_V2_WGS=1024" "new_w512:PKV_LOGITS_V2_WGS=512" "new2:" "base2:PKV_LIB=$B" 2>&1 | tee gpurun_out/defer/ab.txt
