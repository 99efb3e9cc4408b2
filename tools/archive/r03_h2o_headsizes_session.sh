#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/h2o_sizes
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 900 -k "h2o or head_sizes" > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
python - > $O/h2o_time.txt 2>&1 <<PY
import torch, sys
sys.path.insert(0, "$R")
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
for D in (128, 64, 256):
    q, k = (torch.randn(1, 32, 32768, D, device="cuda").to(torch.bfloat16) for _ in range(2))
    P.ops.score_h2o(q, k, 8); torch.cuda.synchronize()
    N.prof_enable(True); N.prof_read(True)
    for _ in range(3): P.ops.score_h2o(q, k, 8)
    torch.cuda.synchronize()
    pr = N.prof_read(True); N.prof_enable(False)
    print(D, {kk: round(v[0]/v[1], 3) for kk, v in pr.items() if v[1]})
PY
grep -E "passed|failed" $O/pytest.txt | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.txt | head; cat $O/h2o_time.txt | tail -5
