"""Ada-SnapKV / HeadKV update_kv timing (BASELINE config 5: 32 query heads, 8 KV heads, S = 32768): wall time per call and
device time per kernel, K/V handed over un-expanded (what the adapter does) and expanded (the reference's contract)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {}
H, Hkv, D, w = 32, 8, 128, 8


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    N.prof_enable(True); N.prof_read(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    prof = N.prof_read(True); N.prof_enable(False)
    return ms, {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items() if v[1]}


for S in (8192, 32768):
    q = torch.randn(1, H, S, D, device="cuda").to(torch.bfloat16)
    ku, vu = (torch.randn(1, Hkv, S, D, device="cuda").to(torch.bfloat16) for _ in range(2))
    kx = ku[:, :, None].expand(1, Hkv, 4, S, D).reshape(1, H, S, D).contiguous()
    vx = vu[:, :, None].expand(1, Hkv, 4, S, D).reshape(1, H, S, D).contiguous()
    for cap in (128, 2048):
        for name, (k, v) in (("unexpanded", (ku, vu)), ("expanded", (kx, vx))):
            cl = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2,
                                normalize=True, layer_idx=0, num_hidden_layers=32)
            ms, prof = timed(lambda: cl.update_kv(k, q, v), 10)
            res[f"adakv_S{S}_cap{cap}_{name}"] = dict(update_kv_ms=round(ms, 4), kernels_us=prof, max_head_len=int(cl.max_seqlen_k),
                                                      klen_sum=int(cl.klen_sum))
    hc = [[int(x) for x in torch.randint(40, 400, (H,)).tolist()]]
    cl = P.HeadKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=128, layer_idx=0,
                         num_hidden_layers=32, head_capacity=hc)
    ms, prof = timed(lambda: cl.update_kv(ku, q, vu), 10)
    res[f"headkv_S{S}_unexpanded"] = dict(update_kv_ms=round(ms, 4), kernels_us=prof)
    del q, ku, vu, kx, vx
print(json.dumps(res, indent=1))
