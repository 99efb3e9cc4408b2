"""LOOK-M merge alone under a profiler: `python tools/merge_only.py S cap [D]` runs pkv_merge_compact 10 times on one
[1, 32, S, D] bf16 tensor with the SnapKV selection of that budget (rocprofv3 --kernel-trace --stats gives the per-kernel split)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
S, cap = int(sys.argv[1]), int(sys.argv[2])
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
w = 8
q, k, v = (torch.randn(1, 32, S, D, device="cuda").to(torch.bfloat16) for _ in range(3))
idx = P.ops.select(q, k, w, cap - w, "maxpool", 7)
for _ in range(10):
    P.ops.merge_compact(k, v, idx, w)
torch.cuda.synchronize()
