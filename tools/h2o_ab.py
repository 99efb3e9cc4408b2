"""A/B of H2O kernel variants: every library given on the command line (PKV_LIB) runs the H2O scores at S and reports
the hipEvent times of the two passes plus how many scores differ from the first library's."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import pyramidkv_amd as P
    from pyramidkv_amd import _native as N
    S = int(sys.argv[2])
    g = torch.Generator(device="cuda").manual_seed(5)
    q, k = (torch.randn(1, 32, S, 128, device="cuda", generator=g) for _ in range(2))
    sc_ = float(os.environ.get("H2O_SCALE", "1"))          # q, k x scale: logits x scale^2 (large-norm models)
    q, k = q * sc_, k * sc_
    if os.environ.get("H2O_OUTLIER"):                      # one massive-activation key per head that no query attends to
        k[:, :, 7] = -q[:, :, -64:].mean(2) * float(os.environ["H2O_OUTLIER"])
    q, k = q.to(torch.bfloat16), k.to(torch.bfloat16)
    for _ in range(2):
        sc = P.ops.score_h2o(q, k, 8)
    torch.cuda.synchronize()
    N.prof_enable(True)
    for _ in range(3):
        sc = P.ops.score_h2o(q, k, 8)
    torch.cuda.synchronize()
    r = N.prof_read()
    N.prof_enable(False)
    out = {kk: round(1e3 * v[0] / max(1, v[1]), 1) for kk, v in r.items() if kk.startswith("h2o") and v[1]}
    torch.save(sc.cpu(), sys.argv[3])
    print(json.dumps(out))
    sys.exit(0)
import torch
S = int(sys.argv[1])
ref = None
for lib in sys.argv[2:]:
    env = dict(os.environ, PKV_LIB=os.path.join(ROOT, lib))
    tmp = f"/tmp/h2o_{os.path.basename(lib)}.pt"
    r = subprocess.run([sys.executable, __file__, "--one", str(S), tmp], env=env, capture_output=True, text=True)
    if r.returncode:
        print(lib, "FAILED", r.stderr[-400:]); continue
    d = json.loads(r.stdout.strip().splitlines()[-1])
    sc = torch.load(tmp)
    if ref is None:
        ref = sc
    d["differs_from_first"] = int((sc.view(torch.int16) != ref.view(torch.int16)).sum())
    d["of"] = sc.numel()
    print(lib, json.dumps(d), flush=True)
