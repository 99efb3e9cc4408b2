#!/bin/bash
# Session 3: parity + A/B of logits kernel variants + top-k phase trace + bench with defaults.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1
echo "pytest exit $?" >> $O/pytest.txt
python tools/topk_trace.py > $O/topk_trace.json 2> $O/topk_trace.err
for tile in 128 256; do for nt in 0 1; do
  PKV_LOGITS_TILE=$tile PKV_LOGITS_NT=$nt timeout 600 python tools/sweep.py quick > $O/sweep_tile${tile}_nt${nt}.json 2> $O/sweep_tile${tile}_nt${nt}.err
done; done
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -4 $O/pytest.txt; cat $O/bench.json
