#!/usr/bin/env python
"""A/B of the one-launch score form (PKV_SCORE_FUSED, csrc/pkv_score.hip score_fused_kernel: K scan + finalize in one launch,
finalize workgroups appended to the scan's grid) against the two-launch form.

Timing: the headline step (32 PyramidKV layer budgets, [1,32,S,128] bf16, 4 rotating input sets), one process per setting (the
knob is read once): us per update_kv (wall, back-to-back calls), host us to issue one call, per-kernel us (pkv_prof).
Identity: K, V and indices of every layer equal the two-launch form's; plus a list of odd shapes (fp16, batches, GQA groups,
ragged lengths, window 32 / avgpool 5, budget 2048) compared the same way, each run 3 times (a stale read shows as a flake)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODD = [  # B, H, Hkv, S, dtype, window, pooling, kernel, k
    (1, 32, 32, 32768, "fp16", 8, "maxpool", 7, 120),
    (2, 32, 8, 8192, "bf16", 8, "maxpool", 7, 234),
    (8, 32, 32, 4096, "bf16", 8, "maxpool", 7, 120),
    (1, 32, 8, 32768, "bf16", 8, "maxpool", 7, 2040),
    (1, 32, 32, 32768, "bf16", 32, "avgpool", 5, 96),
    (1, 8, 2, 5000, "fp16", 16, "avgpool", 5, 300),
    (3, 5, 5, 1111, "bf16", 8, "maxpool", 7, 17),
    (1, 32, 32, 300, "bf16", 8, "maxpool", 7, 100),
    (1, 16, 4, 16384, "bf16", 4, "maxpool", 3, 64),
    (1, 32, 32, 65000, "bf16", 8, "maxpool", 7, 128),
]
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import pyramidkv_amd as P
    from pyramidkv_amd import _native as N
    S, kvg = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1234)
    sets = [tuple(torch.randn(1, 32 if i == 0 else 32 // kvg, S, 128, generator=g, device=dev).to(torch.bfloat16) for i in range(3)) for _ in range(4)]
    ks = []
    for layer in range(32):
        cl = P.PyramidKVCluster(num_hidden_layers=32, layer_idx=layer, window_size=8, max_capacity_prompt=128, kernel_size=7, pooling="maxpool")
        ks.append(cl.layer_budget(S)[1])

    def step(keep=None):
        for layer in range(32):
            q, k, v = sets[layer % 4]
            out = P.ops.compress(q, k, v, 8, ks[layer], "maxpool", 7, kv_group=kvg, return_indices=keep is not None)
            if keep is not None:
                keep.append(tuple(t.clone() for t in out))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 320 * 1e6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host = (time.perf_counter() - t0) / 32 * 1e6
    torch.cuda.synchronize()
    kern = None
    try:
        N.prof_enable(True)
        N.prof_read(reset=True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        kern = {name: round(ms / n * 1e3, 2) for name, (ms, n) in N.prof_read(reset=True).items() if n}
        N.prof_enable(False)
    except Exception as e:      # the timing API differs between rounds: the A/B does not depend on it
        kern = {"error": repr(e)[:200]}
    keep = []
    step(keep)
    torch.cuda.synchronize()
    saved = [tuple(t.cpu() for t in o) for o in keep]
    odd = []
    for (B, H, Hkv, So, dt, w, pool, ksz, k) in ODD:
        dtt = torch.bfloat16 if dt == "bf16" else torch.float16
        gg = torch.Generator(device=dev).manual_seed(B * 1000 + So)
        q = torch.randn(B, H, So, 128, generator=gg, device=dev).to(dtt)
        kk = torch.randn(B, Hkv, So, 128, generator=gg, device=dev).to(dtt)
        vv = torch.randn(B, Hkv, So, 128, generator=gg, device=dev).to(dtt)
        outs = []
        for rep in range(3):
            o = P.ops.compress(q, kk, vv, w, k, pool, ksz, kv_group=H // Hkv, return_indices=True)
            outs.append(tuple(t.cpu() for t in o))
        stable = all(all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:])
        odd.append((outs[0], stable))
    torch.save({"headline": saved, "odd": odd}, sys.argv[4])
    print(json.dumps({"us_per_update_kv": round(best, 2), "host_us": round(host, 2), "kernels_us": kern}))
    sys.exit(0)
import torch
for S, kvg in ((32768, 1), (16384, 1), (8192, 1), (4096, 1), (32768, 4), (8192, 4)):
    ref = None
    for fused in (0, 1):
        tmp = "/tmp/sf_%d.pt" % fused
        r = subprocess.run([sys.executable, __file__, "--one", str(S), str(kvg), tmp], env=dict(os.environ, PKV_SCORE_FUSED=str(fused)),
                           capture_output=True, text=True, timeout=600)
        if r.returncode:
            print(S, kvg, fused, "FAILED", r.stderr[-1500:], flush=True)
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out = torch.load(tmp)
        if ref is None:
            ref = out
        d["identical_to_two_launch"] = all(all(torch.equal(a, b) for a, b in zip(x, y)) for x, y in zip(out["headline"], ref["headline"]))
        d["odd_shapes_identical"] = [all(torch.equal(a, b) for a, b in zip(x[0], y[0])) and x[1] for x, y in zip(out["odd"], ref["odd"])]
        print("S=%d kv_group=%d PKV_SCORE_FUSED=%d %s" % (S, kvg, fused, json.dumps(d)), flush=True)
