"""Phase timestamps (shader clock) of topk_kernel row 0 and finalize_kernel block (1,0)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
for L, k in ((32760, 120), (32760, 2040), (8184, 120)):
    s = torch.rand(32, L, device="cuda").to(torch.bfloat16) * 1e-3
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        P.ops.topk(s, k)
    N.lib.pkv_debug_topk_trace(buf.data_ptr())
    P.ops.topk(s, k)
    torch.cuda.synchronize()
    N.lib.pkv_debug_topk_trace(None)
    t = buf.cpu().tolist()
    res[f"topk_L{L}_k{k}"] = {"loads_arrived": t[7] - t[0], "keys+hist1": t[1] - t[7], "reduce+find1": t[2] - t[1],
                               "passB+find2": t[3] - t[2], "passC": t[4] - t[3], "passD": t[5] - t[4], "order+store": t[6] - t[5],
                               "total": t[6] - t[0]}
for B, S in ((1, 32768), (8, 32768)):
    q = torch.randn(B, 32, S, 128, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, 32, S, 128, device="cuda").to(torch.bfloat16)
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    for _ in range(2):
        P.ops.score_window(q, k, 8, "maxpool", 7)
    N.lib.pkv_debug_topk_trace(buf.data_ptr())
    P.ops.score_window(q, k, 8, "maxpool", 7)
    torch.cuda.synchronize()
    N.lib.pkv_debug_topk_trace(None)
    t = buf.cpu().tolist()[8:]
    res[f"finalize_B{B}"] = {"prologue(stats)": t[1] - t[0], "main(load+exp)": t[2] - t[1], "lds+barrier": t[3] - t[2],
                             "pool+store": t[4] - t[3], "total": t[4] - t[0]}
    del q, k
print(json.dumps(res, indent=1))
