"""Per-kernel device time of update_kv when K/V arrive un-expanded (8 KV heads, 32 query heads), B=1, S=32768."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
S = 32768
sets = [[torch.randn(1, h, S, 128, device="cuda").to(torch.bfloat16) for h in (32, 8, 8)] for _ in range(6)]
out = {}
for kk in (17, 120, 234):
    for i in range(6):
        q, k, v = sets[i % 6]
        P.ops.compress(q, k, v, 8, kk, "maxpool", 7, kv_group=4)
    torch.cuda.synchronize()
    N.prof_enable(True)
    for i in range(30):
        q, k, v = sets[i % 6]
        P.ops.compress(q, k, v, 8, kk, "maxpool", 7, kv_group=4)
    torch.cuda.synchronize()
    r = N.prof_read()
    N.prof_enable(False)
    out[f"k{kk}"] = {n: round(1e3 * ms / max(1, c), 2) for n, (ms, c) in r.items() if c}
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(60):
        q, k, v = sets[i % 6]
        P.ops.compress(q, k, v, 8, kk, "maxpool", 7, kv_group=4)
    t1.record(); torch.cuda.synchronize()
    out[f"k{kk}"]["update_kv_us"] = round(t0.elapsed_time(t1) * 1e3 / 60, 2)
print(json.dumps(out, indent=1))
