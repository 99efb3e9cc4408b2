"""Per-kernel roofline sweep over (policy, budget, batch, S): device time per kernel from libpkv's own
hipEvent pairs; prints one JSON object.  Run on the GPU box."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P  # noqa: E402
from pyramidkv_amd import _native as N  # noqa: E402

PEAK = 8000.0
dev = torch.device("cuda", 0)
out = {}
w, H, D, e = 8, 32, 128, 2
cfgs = [(S, cap, B) for S in (8192, 32768) for cap in (128, 2048) for B in (1, 4, 8)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    cfgs = [(32768, 128, 1), (32768, 2048, 1), (32768, 2048, 8)]
for S, cap, B in cfgs:
    k_sel = cap - w
    sets = []
    for i in range(2 if B * S >= 4 * 32768 else 4):
        g = torch.Generator(device=dev).manual_seed(i)
        sets.append(tuple(torch.randn(B, H, S, D, generator=g, device=dev).to(torch.bfloat16) for _ in range(3)))
    for it in range(3):
        q, k, v = sets[it % len(sets)]
        P.ops.compress(q, k, v, w, k_sel, "maxpool", 7)
    torch.cuda.synchronize()
    N.prof_enable(True)
    N.prof_read(reset=True)
    iters = 10
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(iters):
        q, k, v = sets[it % len(sets)]
        P.ops.compress(q, k, v, w, k_sel, "maxpool", 7)
    torch.cuda.synchronize()
    prof = N.prof_read(reset=True)
    N.prof_enable(False)
    ev0.record()
    for it in range(iters):
        q, k, v = sets[it % len(sets)]
        P.ops.compress(q, k, v, w, k_sel, "maxpool", 7)
    ev1.record()
    torch.cuda.synchronize()
    call_us = ev0.elapsed_time(ev1) / iters * 1e3
    n = B * H
    alg = {"logits": n * (S * D * e + w * D * e), "finalize": n * (w * S * e + (S - w) * e),
           "topk": n * ((S - w) * e + k_sel * 4), "gather": 4 * (k_sel + w) * D * e * n}
    row = {"update_kv_us": round(call_us, 2), "tokens_per_s": round(B * S / call_us * 1e6, 0)}
    for name, by in alg.items():
        ms, cnt = prof[name]
        us = ms / cnt * 1e3
        row[name] = {"us": round(us, 2), "GBps": round(by / us / 1e3, 1), "frac_of_8TBps": round(by / us / 1e3 / PEAK, 4),
                     "alg_MB": round(by / 1e6, 2)}
    out[f"snapkv_S{S}_cap{cap}_B{B}"] = row
    del sets
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
