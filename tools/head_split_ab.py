#!/usr/bin/env python
"""A/B of the head-group fork / join inside one update_kv (PKV_HEAD_SPLIT, csrc/pkv_api.hip compress_common): the headline
step (32 PyramidKV layer budgets, [1,32,S,128] bf16, 4 rotating input sets) per split, one process per setting (the knob is read
once).  Prints us per update_kv (wall, back-to-back calls), host us to issue one call, and whether K/V/indices equal split 0."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import pyramidkv_amd as P
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
    keep = []
    step(keep)
    torch.cuda.synchronize()
    torch.save([tuple(t.cpu() for t in o) for o in keep], sys.argv[4])
    print(json.dumps({"us_per_update_kv": round(best, 2), "host_us": round(host, 2)}))
    sys.exit(0)
import torch
for S, kvg in ((32768, 1), (16384, 1), (8192, 1), (32768, 4)):
    ref = None
    for split in (0, 16, 24, 28):
        tmp = "/tmp/hs_%d.pt" % split
        r = subprocess.run([sys.executable, __file__, "--one", str(S), str(kvg), tmp], env=dict(os.environ, PKV_HEAD_SPLIT=str(split)), capture_output=True, text=True)
        if r.returncode:
            print(S, kvg, split, "FAILED", r.stderr[-600:])
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out = torch.load(tmp)
        if ref is None:
            ref = out
        d["identical_to_split0"] = all(all(torch.equal(a, b) for a, b in zip(x, y)) for x, y in zip(out, ref))
        print("S=%d kv_group=%d PKV_HEAD_SPLIT=%d %s" % (S, kvg, split, json.dumps(d)), flush=True)
