"""Un-expanded GQA K scan by batch size: logits kernel time (events on the dispatch) and the fraction of 8 TB/s over the
algorithmic K bytes, B in {1, 2, 4, 8}, 32 query heads over 8 KV heads, S = 32768.  Does the B = 1 figure come from the
kernel's steady state or from its ramp?"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
S, H, HK, D, w = 32768, 32, 8, 128, 8
out = {}
for B in (1, 2, 4, 8):
    sets = [(torch.randn(B, H, S, D, device="cuda").to(torch.bfloat16), torch.randn(B, HK, S, D, device="cuda").to(torch.bfloat16),
             torch.randn(B, HK, S, D, device="cuda").to(torch.bfloat16)) for _ in range(3)]
    for q, k, v in sets:
        P.ops.compress(q, k, v, w, 120, "maxpool", 7, kv_group=H // HK)
    torch.cuda.synchronize()
    N.prof_enable(True); N.prof_read()
    for _ in range(5):
        for q, k, v in sets:
            P.ops.compress(q, k, v, w, 120, "maxpool", 7, kv_group=H // HK)
    torch.cuda.synchronize()
    pr = N.prof_read(); N.prof_enable(False)
    us = {kk: round(ms / max(n, 1) * 1e3, 2) for kk, (ms, n) in pr.items() if n}
    kbytes = B * HK * S * D * 2
    out[f"B{B}"] = dict(kernels_us=us, logits_frac_of_8TBps=round(kbytes / (us["logits"] * 1e-6) / 8e12, 4))
    del sets
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
