"""Per-workgroup phase stamps of score_fused_kernel (100 MHz wall clock)."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
for B in (1, 8):
    S = 32768
    q, k = (torch.randn(B, 32, S, 128, device="cuda").to(torch.bfloat16) for _ in range(2))
    for _ in range(3):
        P.ops.score_window(q, k, 8, "maxpool", 7)
    buf = torch.zeros(2 * 262144, dtype=torch.int64, device="cuda")
    N.lib.pkv_debug_wg_trace(buf.data_ptr())
    P.ops.score_window(q, k, 8, "maxpool", 7)
    torch.cuda.synchronize()
    N.lib.pkv_debug_wg_trace(None)
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 3] > 0]
    t0 = t[:, 0].min()
    u = (t - t0) / 100.0
    pc = lambda x: [round(float(v), 2) for v in np.percentile(x, [0, 10, 50, 90, 100])]
    res[f"B{B}"] = dict(wgs=int(len(t)), start=pc(u[:, 0]), phase1_done=pc(u[:, 1]), met=pc(u[:, 2]), end=pc(u[:, 3]),
                        phase1_len=pc(u[:, 1] - u[:, 0]), wait=pc(u[:, 2] - u[:, 1]), phase2_len=pc(u[:, 3] - u[:, 2]))
    del q, k
print(json.dumps(res, indent=1))
# extra: imbalance analysis of the B=1 run (rerun, keep raw)
S = 32768
q, k = (torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16) for _ in range(2))
q2, k2 = (torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16) for _ in range(2))
for _ in range(3):
    P.ops.score_window(q, k, 8, "maxpool", 7)
P.ops.score_window(q2, k2, 8, "maxpool", 7)
buf = torch.zeros(2 * 262144, dtype=torch.int64, device="cuda")
N.lib.pkv_debug_wg_trace(buf.data_ptr())
P.ops.score_window(q, k, 8, "maxpool", 7)
torch.cuda.synchronize()
N.lib.pkv_debug_wg_trace(None)
t = buf.cpu().numpy().reshape(-1, 4)
n = int((t[:, 3] > 0).sum())
t = t[:n].astype(np.float64)
t0 = t[:, 0].min()
p1 = (t[:, 1] - t[:, 0]) / 100.0
idx = np.arange(n)
out = {"n": n}
out["by_xcd_median_p1"] = [round(float(np.median(p1[idx % 8 == x])), 1) for x in range(8)]
out["by_xcd_max_p1"] = [round(float(p1[idx % 8 == x].max()), 1) for x in range(8)]
wph = 22
ci = idx % wph
out["by_ci_median_p1"] = [round(float(np.median(p1[ci == c])), 1) for c in range(wph)]
grp = idx // wph
out["by_grp_median_p1"] = [round(float(np.median(p1[grp == g])), 1) for g in range(32)]
print(json.dumps(out))
