import os, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from inputs import make_qkv, bits
from oracle import pkv_oracle as O
import pyramidkv_amd as P
def ord16(t):
    b = bits(t).astype(np.int32); return np.where(b & 0x8000, -(b & 0x7FFF), b)
res = {}
for dt in ("bf16", "fp16"):
    for scale in (5, 6, 7, 8, 10):
        q, k, _ = make_qkv(1, 2, 1200, 128, dt, "lattice", 97)
        q *= scale; k *= scale
        want = O.h2o_scores(q, k, 8)
        got = P.ops.score_h2o(q.cuda(), k.cuda(), 8).cpu()
        tiny = want.float().abs() < 1e-35
        got, want = got.clone(), want.clone(); got[tiny] = 0; want[tiny] = 0
        d = np.abs(ord16(got) - ord16(want))
        res["%s x%d" % (dt, scale)] = [int((d > 0).sum()), int((d > 1).sum()), int(d.max())]
print(os.environ.get("PKV_LIB", "default"), json.dumps(res))
