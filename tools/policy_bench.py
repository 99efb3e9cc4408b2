"""Timing of the non-headline policies: H2O (S x S MFMA passes) and Ada-SnapKV (sort + budgets + flat gather)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
H, D, w = 32, 128, 8


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    N.prof_enable(True); N.prof_read(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    prof = N.prof_read(True); N.prof_enable(False)
    return a.elapsed_time(b) / iters, {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items() if v[1]}


for S in (4096, 8192, 32768):
    q, k, v = (torch.randn(1, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(3))
    iters = 3 if S > 8192 else 10
    ms, prof = timed(lambda: P.ops.compress(q, k, v, w, 120, None, 1, h2o=True), iters)
    flops = 2 * 2 * S * S * D * H
    res[f"h2o_S{S}"] = dict(update_kv_ms=round(ms, 3), tokens_per_s=round(S / ms * 1e3), kernels_us=prof,
                            mfma_TFLOPs=round(flops / (ms * 1e-3) / 1e12, 1))
    for cap in (128, 2048):
        cl = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2,
                            normalize=True, layer_idx=0, num_hidden_layers=32)
        ms, prof = timed(lambda: cl.update_kv(k, q, v), 10)
        res[f"adakv_S{S}_cap{cap}"] = dict(update_kv_ms=round(ms, 3), tokens_per_s=round(S / ms * 1e3), kernels_us=prof,
                                           max_head_len=int(cl.max_seqlen_k), klen_sum=int(cl.klen_sum))
    ms, prof = timed(lambda: P.StreamingLLMKVCluster(window_size=124, max_capacity_prompt=128).update_kv(k, q, v, None, 1), 20)
    res[f"streaming_S{S}"] = dict(update_kv_ms=round(ms, 4), kernels_us=prof)
    del q, k, v
print(json.dumps(res, indent=1))
