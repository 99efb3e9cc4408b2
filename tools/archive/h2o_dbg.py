import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pyramidkv_amd as P
from inputs import make_qkv
from oracle import pkv_oracle as O
for dt in ("fp16", "bf16"):
    tot = bad = 0
    for (S, w) in ((1030, 32), (1030, 8), (1030, 1), (767, 32), (2048, 32), (4096, 8)):
        for seed in range(3):
            q, k, _ = make_qkv(1, 2, S, 128, dt, "gauss", 5000 + S + w + seed)
            want = O.h2o_scores(q, k, w)
            got = P.ops.score_h2o(q.to("cuda"), k.to("cuda"), w).cpu()
            d = (got.view(torch.int16).int() - want.view(torch.int16).int())
            nz = d.nonzero()
            tot += d.numel(); bad += len(nz)
            print(dt, S, w, seed, "mismatches", len(nz), "of", d.numel(), "cols", nz[:, 2].tolist()[:12], "signs", d[d != 0].tolist()[:12])
    print(dt, "pooled", bad / tot)
