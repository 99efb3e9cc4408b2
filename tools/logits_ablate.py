"""Time logits_kernel alone (hipEvents inside libpkv) for the PKV_LOGITS_ABLATE value of this process."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
out = {"ablate": os.environ.get("PKV_LOGITS_ABLATE", "0"), "nt": os.environ.get("PKV_LOGITS_NT", "0"), "rm": os.environ.get("PKV_LOGITS_ROWMAJOR", "0"), "tile": os.environ.get("PKV_LOGITS_TILE", "256"), "v2": os.environ.get("PKV_LOGITS_V2", "1"), "wgs": os.environ.get("PKV_LOGITS_V2_WGS", "0"), "cm": os.environ.get("PKV_LOGITS_CHUNKMAJOR", "1")}
for B in (1, 8):
    S = 32768
    sets = [[torch.randn(B, 32, S, 128, device="cuda").to(torch.bfloat16) for _ in range(2)] for _ in range(3 if B == 8 else 5)]
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
    kk = "score_fused" if r["score_fused"][1] else "logits"
    us = 1e3 * r[kk][0] / r[kk][1]
    out[f"B{B}"] = {"logits_us": round(us, 2), "GBps": round(B * 32 * S * 256 / us / 1e3, 0),
                    "finalize_us": round(1e3 * r["finalize"][0] / max(1, r["finalize"][1]), 2)}
    del sets
print(json.dumps(out))
