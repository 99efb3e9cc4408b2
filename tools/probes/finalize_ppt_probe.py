import sys, os, json, torch
sys.path.insert(0, "/root/repo")
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
for B, S in ((8, 32768), (4, 32768), (8, 8192), (1, 32768), (1, 8192), (1, 4096)):
    q = torch.randn(B, 32, S, 128, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, 32, S, 128, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        P.ops.score_window(q, k, 8, "maxpool", 7)
    N.prof_enable(True); N.prof_read(True)
    for _ in range(20):
        P.ops.score_window(q, k, 8, "maxpool", 7)
    torch.cuda.synchronize()
    pr = N.prof_read(True); N.prof_enable(False)
    res["B%d_S%d" % (B, S)] = {kk: round(ms / c * 1e3, 2) for kk, (ms, c) in pr.items() if c}
    del q, k
print(json.dumps(res))
