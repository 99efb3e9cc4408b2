"""top-k at the SnapKV budget-2048 list length (k = 2040) and around it on real max-pooled score rows: device time per launch and,
with PKV_LIB=pyramidkv_amd/libpkv_debug.so, the phase stamps of row 0 in shader cycles (full path of topk_kernel: 1 = high-byte
histogram, 2 = first select, 3 = low-byte histogram + select, 4 = per-wave counts, 5 = compaction, 6 = ordering + emit)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {"debug_build": bool(N.lib.pkv_debug_build())}
for S in (8192, 32768):
    q = torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16)
    kk = torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16)
    real = P.ops.score_window(q, kk, 8, "maxpool", 7, "sum")[0]
    for k in (600, 1024, 2040, 3978):
        row = {}
        for _ in range(3):
            P.ops.topk(real, k)
        N.prof_enable(True); N.prof_read(True)
        for _ in range(20):
            P.ops.topk(real, k)
        torch.cuda.synchronize()
        pr = N.prof_read(True); N.prof_enable(False)
        row["topk_us"] = round(pr["topk"][0] / pr["topk"][1] * 1e3, 2)
        if res["debug_build"]:
            buf = torch.zeros(16, dtype=torch.int64, device="cuda")
            N.lib.pkv_debug_topk_trace(buf.data_ptr())
            P.ops.topk(real, k)
            torch.cuda.synchronize()
            N.lib.pkv_debug_topk_trace(None)
            t = buf.cpu().tolist()
            row["stamps_rel_cycles"] = {str(i): (t[i] - t[0]) for i in range(1, 15) if t[i]}
        res["S%d_k%d" % (S, k)] = row
print(json.dumps(res, indent=1))
