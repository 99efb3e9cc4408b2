#!/bin/bash
# Short GPU session for A/B work: parity suite + one bench line (edit freely; tools/gpu_session.sh is the full one).
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['kv_compress_ms_per_layer'], {k:v['avg_us'] for k,v in d['roofline_kernels'].items()}, d.get('extras',{}).get('gqa_dedup_us_per_layer'))"
