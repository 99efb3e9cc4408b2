"""A/B of two libpkv builds on the selection kernel: one process per library (PKV_LIB).  Rows: real window scores ([1,32,32768,128],
maxpool 7) of N(0,1) inputs in bf16 / fp16 and of the attention-sink inputs in fp16 (thousands of zeros tied at the k-th value);
per-kernel us of one SnapKV update_kv at budgets 64 / 128 / 256 / 512 (events on the dispatches, 100 calls).
  python tools/topk_ztie_ab.py libA.so libB.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import pyramidkv_amd as P
    from pyramidkv_amd import _native as N
    from inputs import make_qkv
    res = {}
    for name, dt, kind in (("gauss_bf16", "bf16", "gauss"), ("gauss_fp16", "fp16", "gauss"), ("sink_fp16", "fp16", "sink")):
        q, k, v = (t.cuda() for t in make_qkv(1, 32, 32768, 128, dt, kind, 6600))
        for cap in (64, 128, 256, 512):
            cl = P.SnapKVCluster(window_size=8, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool")
            for _ in range(5):
                cl.update_kv(k, q, v, None, 1)
            torch.cuda.synchronize()
            N.prof_enable(True); N.prof_read(True)
            for _ in range(100):
                cl.update_kv(k, q, v, None, 1)
            torch.cuda.synchronize()
            pr = N.prof_read(True); N.prof_enable(False)
            res["%s_budget%d" % (name, cap)] = {kk: round(ms / c * 1e3, 2) for kk, (ms, c) in pr.items() if c and kk in ("topk", "finalize", "gather")}
    print(json.dumps(res))
    sys.exit(0)
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, PKV_LIB=os.path.abspath(lib)), capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(lib, line[-1] if line else ("FAILED " + r.stderr[-800:]), flush=True)
