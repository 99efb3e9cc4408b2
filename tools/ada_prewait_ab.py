"""Same-process A/B of config.ada_prewait (AdaKVCluster's prepared call: the narrowed output views and the next call's metadata
buffer are made WHILE the kernels run): us per update_kv, BASELINE config 5 shapes (32 query heads, 8 un-expanded KV heads, bf16),
budget 128 / 2048, S = 4096 / 8192 / 32768; settings alternate A B A B ... in blocks of 50 calls, median of the blocks."""
import json, os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import config as cfg
res = {}
for S in (4096, 8192, 32768):
    q = torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16)
    k, v = (torch.randn(1, 8, S, 128, device="cuda").to(torch.bfloat16) for _ in range(2))
    for cap in (128, 2048):
        cls = [P.AdaKVCluster(window_size=8, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2, normalize=True,
                              layer_idx=i, num_hidden_layers=32) for i in range(4)]
        outs = {}
        for on in (True, False):
            cfg.ada_prewait = on
            for cl in cls:
                for _ in range(3):
                    o = cl.update_kv(k, q, v)
            outs[on] = (o[0].clone(), o[1].clone())
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
        blocks = {True: [], False: []}
        for rep in range(12):
            for on in (True, False):
                cfg.ada_prewait = on
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(50):
                    cls[i & 3].update_kv(k, q, v)
                torch.cuda.synchronize()
                blocks[on].append((time.perf_counter() - t0) / 50 * 1e6)
        res["S%d_budget%d" % (S, cap)] = {"prewait_on_us": round(statistics.median(blocks[True]), 2), "prewait_off_us": round(statistics.median(blocks[False]), 2),
                                          "identical": True}
cfg.ada_prewait = True
print(json.dumps(res, indent=1))
