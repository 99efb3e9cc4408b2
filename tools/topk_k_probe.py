"""top-k on real (mean-reduced, max-pooled) score rows for the list lengths Ada-SnapKV asks for: device time per launch (release
library, events on the dispatch) and - with PKV_LIB=libpkv_debug.so - the phase stamps of row 0 and the candidate count.
  python tools/topk_k_probe.py            PKV_LIB=pyramidkv_amd/libpkv_debug.so python tools/topk_k_probe.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import _native as N
res = {"debug_build": bool(N.lib.pkv_debug_build())}
for S in (8192, 32768):
    q = torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16)
    kk = torch.randn(1, 8, S, 128, device="cuda").to(torch.bfloat16)
    real = P.ops.score_window(q, kk, 8, "maxpool", 7, "mean", kv_group=4)[0]
    for k in (120, 256, 384, 448, 512, 960):
        row = {}
        for _ in range(3):
            P.ops.topk(real, k)
        N.prof_enable(True); N.prof_read(True)
        for _ in range(20):
            P.ops.topk(real, k)
        torch.cuda.synchronize()
        pr = N.prof_read(True); N.prof_enable(False)
        row["topk_us"] = round(pr["topk"][0] / pr["topk"][1] * 1e3, 2)
        if res["debug_build"]:
            buf = torch.zeros(16, dtype=torch.int64, device="cuda")
            N.lib.pkv_debug_topk_trace(buf.data_ptr())
            P.ops.topk(real, k)
            torch.cuda.synchronize()
            N.lib.pkv_debug_topk_trace(None)
            t = buf.cpu().tolist()
            row["stamps_rel"] = {str(i): (t[i] - t[0]) for i in (7, 1, 4, 13, 2, 3, 5, 6) if t[i]}
            row["candidates"] = t[15]
        res["S%d_k%d" % (S, k)] = row
    # the whole front half of AdaKVCluster.update_kv (score -> top-M + lists + row sums -> one-launch budgets), per kernel
    mirror = torch.zeros(32, dtype=torch.int64).pin_memory()
    for M in (512,):
        for _ in range(3):
            P.ops.ada_select(q, kk, 8, "maxpool", 7, M, 120, 0.2, True, kv_group=4, host_mirror=mirror, host_seq=1)
        N.prof_enable(True); N.prof_read(True)
        for _ in range(20):
            P.ops.ada_select(q, kk, 8, "maxpool", 7, M, 120, 0.2, True, kv_group=4, host_mirror=mirror, host_seq=1)
        torch.cuda.synchronize()
        pr = N.prof_read(True); N.prof_enable(False)
        res["S%d_ada_select_M%d" % (S, M)] = {k_: round(v_[0] / v_[1] * 1e3, 2) for k_, v_ in pr.items() if v_[1]}
        if res["debug_build"]:           # phase stamps of the one-launch budget kernel (thread 0): words 16.. of the trace buffer
            buf = torch.zeros(32, dtype=torch.int64, device="cuda")
            N.lib.pkv_debug_topk_trace(buf.data_ptr())
            P.ops.ada_select(q, kk, 8, "maxpool", 7, M, 120, 0.2, True, kv_group=4, host_mirror=mirror, host_seq=1)
            torch.cuda.synchronize()
            N.lib.pkv_debug_topk_trace(None)
            t = buf.cpu().tolist()[16:24]
            res["S%d_budget_kernel_stamps_rel" % S] = dict(zip(["loads+zero", "ratio+keys+hist1", "select1", "hist2", "select2", "counts", "finish"],
                                                               [t[i + 1] - t[i] for i in range(7)]))
print(json.dumps(res, indent=1))
