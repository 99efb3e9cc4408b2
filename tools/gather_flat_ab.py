"""A/B of two libpkv builds on the flat var-len gather (Ada-SnapKV / HeadKV): one process per library (PKV_LIB), per-kernel us of
AdaKVCluster.update_kv (budget 128 / 2048) and HeadKVCluster.update_kv at S = 32768, 32 query heads + 8 un-expanded KV heads, and the
dense SnapKV budget-2048 gather; 200 calls each.   python tools/gather_flat_ab.py libA.so libB.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import pyramidkv_amd as P
    from pyramidkv_amd import _native as N
    S = 32768
    q = torch.randn(1, 32, S, 128, device="cuda").to(torch.bfloat16)
    k, v = (torch.randn(1, 8, S, 128, device="cuda").to(torch.bfloat16) for _ in range(2))
    res = {}

    def run(name, fn, n=200):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        N.prof_enable(True); N.prof_read(True)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        pr = N.prof_read(True); N.prof_enable(False)
        res[name] = {kk: round(ms / c * 1e3, 2) for kk, (ms, c) in pr.items() if c}
    for cap in (128, 2048):
        cl = P.AdaKVCluster(window_size=8, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2, normalize=True, layer_idx=0, num_hidden_layers=32)
        run("adakv_budget%d" % cap, lambda: cl.update_kv(k, q, v))
    hc = [[int(x) for x in torch.randint(40, 400, (32,), generator=torch.Generator().manual_seed(1)).tolist()]]
    hk = P.HeadKVCluster(window_size=8, kernel_size=7, pooling="maxpool", max_capacity_prompt=128, layer_idx=0, num_hidden_layers=32, head_capacity=hc)
    run("headkv", lambda: hk.update_kv(k, q, v))
    sn = P.SnapKVCluster(window_size=8, max_capacity_prompt=2048, kernel_size=7, pooling="maxpool")
    run("snapkv_budget2048_dense", lambda: sn.update_kv(k, q, v, None, 4))
    sn2 = P.SnapKVCluster(window_size=8, max_capacity_prompt=128, kernel_size=7, pooling="maxpool")
    run("snapkv_budget128_dense", lambda: sn2.update_kv(k, q, v, None, 4))
    print(json.dumps(res))
    sys.exit(0)
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, PKV_LIB=os.path.abspath(lib)), capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(lib, line[-1] if line else ("FAILED " + r.stderr[-800:]), flush=True)
