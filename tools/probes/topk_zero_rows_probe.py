"""top-k on rows that are mostly zero (fp16 underflow at long context): device time per launch and, with PKV_LIB = the debug build,
which phase stamps of topk_kernel were written (the zero branch writes 1 and 6 only; the ordinary small-k path 1, 4, 13, 2, 3, 5, 6;
the full path overwrites 1 - 6 after the small-k path gave up)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
from inputs import make_qkv
res = {"debug_build": bool(N.lib.pkv_debug_build())}
g = torch.Generator().manual_seed(3)
rows = {}
L = 32760
for npos in (100, 350, 2000, 6000):
    s = torch.zeros(32, L)
    for r in range(32):
        pos = torch.randperm(L, generator=g)[:npos]
        s[r, pos] = torch.randint(1, 12, (npos,), generator=g).float() * 2.0 ** -24
    rows["grid_%d_positives" % npos] = s.to(torch.float16)
q, k, v = make_qkv(1, 32, 32768, 128, "fp16", "sink", 6600)
rows["sink_scores"] = P.ops.score_window(q.cuda(), k.cuda(), 8, "maxpool", 7)[0].cpu()
for name, s in rows.items():
    sd = s.cuda()
    for kk in (120, 504):
        row = {"positives_per_row_mean": float((s.float() > 0).sum(-1).float().mean())}
        for _ in range(3):
            P.ops.topk(sd, kk)
        N.prof_enable(True); N.prof_read(True)
        for _ in range(30):
            P.ops.topk(sd, kk)
        torch.cuda.synchronize()
        pr = N.prof_read(True); N.prof_enable(False)
        row["topk_us"] = round(pr["topk"][0] / pr["topk"][1] * 1e3, 2)
        if res["debug_build"]:
            buf = torch.zeros(16, dtype=torch.int64, device="cuda")
            N.lib.pkv_debug_topk_trace(buf.data_ptr())
            P.ops.topk(sd, kk)
            torch.cuda.synchronize()
            N.lib.pkv_debug_topk_trace(None)
            t = buf.cpu().tolist()
            row["stamps_rel_cycles"] = {str(i): (t[i] - t[0]) for i in range(1, 15) if t[i]}
        res["%s_k%d" % (name, kk)] = row
print(json.dumps(res, indent=1))
