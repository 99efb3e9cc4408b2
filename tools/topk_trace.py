"""Phase timestamps (shader clock) of topk_kernel row 0 - where do its ~20 us go?"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
for L, k in ((32760, 120), (32760, 234), (32760, 2040), (8184, 120)):
    s = torch.rand(32, L, device="cuda").to(torch.bfloat16) * 1e-3
    buf = torch.zeros(8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        P.ops.topk(s, k)
    N.lib.pkv_debug_topk_trace(buf.data_ptr())
    P.ops.topk(s, k)
    torch.cuda.synchronize()
    N.lib.pkv_debug_topk_trace(None)
    t = buf.cpu().tolist()
    names = ["zero+sync", "passA load+hist", "reduce+find1", "passB+find2", "passC count", "passD compact", "order+store"]
    res[f"L{L}_k{k}"] = {names[i]: t[i + 1] - t[i] if i < 6 else None for i in range(6)}
    res[f"L{L}_k{k}"]["total_cycles"] = t[6] - t[0]
print(json.dumps(res, indent=1))
