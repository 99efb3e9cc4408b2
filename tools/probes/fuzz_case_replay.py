"""Replay single cases of tools/parity_fuzz.py by their parameters (the tool's RNG is replayed on the host to get them): score
mismatch fraction / largest distance vs the oracle, the score bar's verdict, and - merge cases - merge_bar's verdict.
  python tools/probes/fuzz_case_replay.py '<json list of cases>'   (PKV_LIB selects the library)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyramidkv_amd as P
from inputs import make_qkv, bits
from oracle import pkv_oracle as O
import score_bar
from merge_bar import check_merge
out = []
for c in json.loads(sys.argv[1]):
    B, H, G, S, w, dt, kind, pool, ks, kk, seed = (c[x] for x in ("B", "H", "G", "S", "w", "dt", "kind", "pool", "ks", "k", "qkv_seed"))
    q, k, v = make_qkv(B, H, S, 128, dt, kind, seed)
    ku, vu = k[:, ::G].contiguous(), v[:, ::G].contiguous()
    ke, ve = ku.repeat_interleave(G, dim=1), vu.repeat_interleave(G, dim=1)
    qd, kd, vd = q.cuda(), ku.cuda(), vu.cuda()
    want = O.pool_scores(O.window_scores(q, ke, w), pool, ks)
    got = P.ops.score_window(qd, kd, w, pool, ks, kv_group=G).cpu()
    d = np.abs(score_bar.ord16(got) - score_bar.ord16(want))
    r = {"case": {x: c[x] for x in ("pol", "S", "w", "dt", "kind", "pool", "k")}, "mismatch_frac": float((d > 0).mean()), "max_ulp": int(d.max()),
         "hist": np.bincount(d.flatten().clip(0, 6)).tolist(), "want_range": [float(want.float().min()), float(want.float().max())],
         "want_zero_frac": float((want.float() == 0).float().mean())}
    try:
        r["score_bar"] = score_bar.check_window_scores(q, ke, w, pool, ks, "sum", got, lambda: P.ops.score_window(qd, kd, w, None, 1, kv_group=G).cpu(), frac_bar=1.0)
    except AssertionError as e:
        r["score_bar_assert"] = str(e.args)[:300]
    if c["pol"] == "merge":
        idx_d = P.ops.select(qd, kd, w, kk, pool, ks, kv_group=G)
        km, vm = P.ops.merge_compact(kd, vd, idx_d, w, kv_group=G)
        try:
            r["merge_moved"] = check_merge(P.ops, ke, ve, idx_d.cpu().long(), w, km, vm, "replay")
        except AssertionError as e:
            r["merge_assert"] = str(e.args)[:400]
    out.append(r)
print(json.dumps(out, indent=1))
