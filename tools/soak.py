"""Determinism soak: random shapes and policies, every call made twice on the same inputs (second time after the workspace
and the caching allocator were churned) - results must be bit-identical.  Looks for races (LDS / global atomics, the pinned
host mirror of Ada-SnapKV, workspace growth), not for parity.  python tools/soak.py [seconds]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(7)
t0, n, bad = time.time(), 0, 0
kinds = {}
while time.time() - t0 < budget:
    pol = rng.choice(["snapkv", "pyramidkv", "h2o", "streamingllm", "adakv", "headkv", "merge", "f32"])
    S = int(rng.choice([rng.randint(70, 600), rng.randint(600, 9000), rng.choice([16384, 32768])])) if pol not in ("h2o",) else int(rng.randint(70, 3000))
    w = int(rng.choice([8, 16, 32, 64, 128]))
    G = int(rng.choice([1, 2, 4]))
    H = G * int(rng.randint(1, 5))
    B = 1 if pol in ("adakv", "headkv") else int(rng.randint(1, 3))
    dt = torch.float32 if pol == "f32" else (torch.bfloat16 if rng.rand() < 0.5 else torch.float16)
    if S <= w + 2:
        continue
    cap = w + int(rng.randint(1, min(S - w, 2500) + 1))
    q = torch.randn(B, H, S, 128, device="cuda").to(dt)
    k = torch.randn(B, H // G, S, 128, device="cuda").to(dt)
    v = torch.randn(B, H // G, S, 128, device="cuda").to(dt)

    def run():
        if pol in ("snapkv", "f32"):
            return P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool").update_kv(k, q, v, None, G)
        if pol == "pyramidkv":
            return P.PyramidKVCluster(num_hidden_layers=32, layer_idx=int(S) % 32, window_size=w, max_capacity_prompt=cap, kernel_size=5,
                                      pooling="avgpool").update_kv(k, q, v, None, G)
        if pol == "h2o":
            return P.H2OKVCluster(window_size=w, max_capacity_prompt=cap).update_kv(k, q, v, None, G)
        if pol == "streamingllm":
            return P.StreamingLLMKVCluster(window_size=w, max_capacity_prompt=cap).update_kv(k, q, v, None, G)
        if pol == "merge":
            return P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool", merge="pivot").update_kv(k, q, v, None, G)
        if pol == "adakv":
            c = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2, normalize=True)
            kf, vf = c.update_kv(k, q, v)
            return kf, vf, c.head_lens.clone()
        caps = [[int(x) for x in rng.randint(1, cap, size=H)]]
        c = P.HeadKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, layer_idx=0, num_hidden_layers=1, head_capacity=caps)
        run.caps = caps
        kf, vf = c.update_kv(k, q, v)
        return kf, vf, c.head_lens.clone()

    if pol == "headkv":
        st = rng.get_state()
    a = [t.clone() for t in run()]
    junk = [torch.empty(int(rng.randint(1, 1 << 24)), device="cuda") for _ in range(3)]
    del junk
    if pol == "headkv":
        rng.set_state(st)
    b = run()
    ok = all(torch.equal(x, y) for x, y in zip(a, b))
    n += 1
    kinds[pol] = kinds.get(pol, 0) + 1
    if not ok:
        bad += 1
        print("NONDETERMINISTIC", pol, B, H, G, S, w, cap, dt, flush=True)
torch.cuda.synchronize()
print("soak: %d cases, %d nondeterministic, by policy %s" % (n, bad, kinds))
sys.exit(1 if bad else 0)
