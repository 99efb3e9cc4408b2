"""Phase timestamps (shader clock) of topk_kernel row 0 on realistic (max-pooled softmax) and uniform scores."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
q = torch.randn(1, 32, 32768, 128, device="cuda").to(torch.bfloat16)
kk = torch.randn(1, 32, 32768, 128, device="cuda").to(torch.bfloat16)
real = P.ops.score_window(q, kk, 8, "maxpool", 7)[0].contiguous()
for name, s, k in (("real_k120", real, 120), ("real_k234", real, 234), ("real_k17", real, 17),
                   ("uniform_k120", (torch.rand(32, 32760, device="cuda") * 1e-3).to(torch.bfloat16), 120),
                   ("real_k2040", real, 2040)):
    buf = torch.zeros(16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        P.ops.topk(s, k)
    N.lib.pkv_debug_topk_trace(buf.data_ptr())
    P.ops.topk(s, k)
    torch.cuda.synchronize()
    N.lib.pkv_debug_topk_trace(None)
    t = buf.cpu().tolist()
    res[name] = {"loads": t[7] - t[0], "t1": t[1] - t[7], "t2": t[2] - t[1], "t3": t[3] - t[2], "t4": t[4] - t[3] if t[4] else None,
                 "t5": t[5] - (t[4] if t[4] else t[3]), "t6": t[6] - t[5], "total": t[6] - t[0], "C": t[15]}
print(json.dumps(res, indent=1))
