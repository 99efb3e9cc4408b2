"""LOOK-M pivot merge timing: SnapKVCluster(merge='pivot').update_kv wall time per call (events) next to the plain gather."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from oracle import pkv_oracle as O      # comparator only: the reference's op sequence run eagerly by PyTorch-ROCm on this GPU
res = {}
H, D, w = 32, 128, 8


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for S in (8192, 32768):
    q, k, v = (torch.randn(1, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(3))
    for cap in (128, 2048):
        plain = P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool")
        merge = P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool", merge="pivot")
        idx = P.ops.select(q, k, w, cap - w, "maxpool", 7)
        res[f"S{S}_cap{cap}"] = dict(update_kv_plain_ms=round(timed(lambda: plain.update_kv(k, q, v, None, 1)), 4),
                                     update_kv_merge_ms=round(timed(lambda: merge.update_kv(k, q, v, None, 1)), 4),
                                     merge_only_ms=round(timed(lambda: P.ops.merge_compact(k, v, idx, w)), 4),
                                     reference_ops_eager_merge_ms=round(timed(lambda: O.merge_kv(k, v, idx.long(), w, "pivot"), 3), 4))
print(json.dumps(res, indent=1))
