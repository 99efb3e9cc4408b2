"""Time the logits kernel alone (hipEvents inside libpkv) under the env knobs of this process."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
keys = ("PKV_LOGITS_ABLATE", "PKV_LOGITS_NT", "PKV_LOGITS_TILE", "PKV_LOGITS_V2", "PKV_LOGITS_V2_WGS")
out = {k[4:].lower(): os.environ[k] for k in keys if k in os.environ}
cases = [tuple(int(x) for x in c.split("x")) for c in os.environ.get("CASES", "1x32768,8x32768").split(",")]
G = int(os.environ.get("GQA", "1"))          # GQA=4: K with 8 heads next to Q with 32 (un-expanded)
for B, S in cases:
    nset = max(2, min(5, int(1.2e9 // (B * 32 * S * 256))))
    sets = [[torch.randn(B, 32, S, 128, device="cuda").to(torch.bfloat16), torch.randn(B, 32 // G, S, 128, device="cuda").to(torch.bfloat16)] for _ in range(nset)]
    _sw = P.ops.score_window
    P.ops.score_window = lambda q, k, w, pool, ks: _sw(q, k, w, pool, ks, kv_group=G)
    for i in range(6):
        q, k = sets[i % len(sets)]
        P.ops.score_window(q, k, 8, "maxpool", 7)
    torch.cuda.synchronize()
    N.prof_enable(True)
    for i in range(30):
        q, k = sets[i % len(sets)]
        P.ops.score_window(q, k, 8, "maxpool", 7)
    torch.cuda.synchronize()
    r = N.prof_read()
    N.prof_enable(False)
    us = 1e3 * r["logits"][0] / r["logits"][1]
    out[f"B{B}_S{S}"] = [round(us, 2), round(B * (32 // G) * S * 256 / us / 1e3), round(1e3 * r["finalize"][0] / max(1, r["finalize"][1]), 2)]
    P.ops.score_window = _sw
    del sets
print(json.dumps(out))
