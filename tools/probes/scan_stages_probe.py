"""K scan (logits2_kernel, expanded K, window 8) per prompt length for a forced workgroup count (PKV_LOGITS_V2_WGS: stages per
workgroup = ceil(stages / that)): device us of the scan and of finalize, [1,32,S,128] bf16."""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {"PKV_LOGITS_V2_WGS": os.environ.get("PKV_LOGITS_V2_WGS", "0")}
for S in (4096, 8192, 16384, 32768):
    q = torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16)
    ks = [torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16) for _ in range(4)]
    for i in range(4):
        P.ops.score_window(q, ks[i], 8, "maxpool", 7)
    N.prof_enable(True); N.prof_read(True)
    for i in range(40):
        P.ops.score_window(q, ks[i & 3], 8, "maxpool", 7)
    torch.cuda.synchronize()
    pr = N.prof_read(True); N.prof_enable(False)
    res["S%d" % S] = {kk: round(ms / c * 1e3, 2) for kk, (ms, c) in pr.items() if c}
print(json.dumps(res))
