"""Where do the H2O scores of one case differ from the oracle?  usage: h2o_dbg2.py lib [lib ...]  (runs itself per library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import pyramidkv_amd as P
    from inputs import make_qkv
    from oracle import pkv_oracle as O
    for dt in ("bf16", "fp16"):
        q, k, _ = make_qkv(1, 2, 1200, 128, dt, "lattice", 97)
        q *= 16; k *= 16
        want = O.h2o_scores(q, k, 8)
        got = P.ops.score_h2o(q.cuda(), k.cuda(), 8).cpu()
        d = (got.view(torch.int16).int() - want.view(torch.int16).int()).abs()
        bad = (d > 0).nonzero()
        print(dt, "mismatches", len(bad), "max ulp", int(d.max()))
        for ix in bad[:8].tolist():
            ix = tuple(ix)
            print("   at", ix, "got", float(got[ix]), "want", float(want[ix]), "ulp", int(d[ix]))
    sys.exit(0)
for lib in sys.argv[1:]:
    print("==", lib, flush=True)
    subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, PKV_LIB=os.path.join(ROOT, lib)))
