"""Window scores on the attention-sink inputs, libpkv vs the oracle: where do they differ (per head: count, largest distance in
units of the last place, the value range of the differing scores) for reduce = sum / mean, fp16 / bf16."""
import sys, os, json, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyramidkv_amd as P
from inputs import make_qkv, bits
from oracle import pkv_oracle as O
res = {}
for dt in ("fp16", "bf16"):
    S, Hq, Hkv, w = 8192, 32, 8, 8
    g = Hq // Hkv
    q, k8, v8 = make_qkv(1, Hq, S, 128, dt, "sink", 4100 + S)
    k_un = k8[:, ::g].contiguous()
    k_exp = k_un[:, :, None].expand(1, Hkv, g, S, 128).reshape(1, Hq, S, 128).contiguous()
    for red in ("sum", "mean"):
        for pool, ks in ((None, 1), ("maxpool", 7)):
            got = P.ops.score_window(q.cuda(), k_un.cuda(), w, pool, ks, red, kv_group=g).cpu()[0]
            want = O.pool_scores(O.window_scores(q, k_exp, w, red), pool, ks)[0]
            gb, wb = bits(got).astype(np.int32), bits(want).astype(np.int32)
            d = np.abs(gb - wb)
            per_head = (d > 0).sum(-1)
            worst = int(per_head.argmax())
            m = d[worst] > 0
            res["%s_%s_%s" % (dt, red, pool)] = {"mismatch_frac": float((d > 0).mean()), "max_ulp": int(d.max()), "worst_head": worst,
                "worst_head_mismatches": int(per_head[worst]), "worst_head_values_min_max": [float(want[worst].float().numpy()[m].min()), float(want[worst].float().numpy()[m].max())] if m.any() else None,
                "worst_head_ulp_hist": np.bincount(d[worst][m].clip(0, 8)).tolist() if m.any() else None,
                "worst_head_want_gt_got": int((wb[worst][m] > gb[worst][m]).sum())}
print(json.dumps(res, indent=1))

# are the scores beyond 1 ulp explained by ONE moved product (the position's own, or - round 6 - the row's maximum)?
import score_bar
for dt in ("fp16",):
    S, Hq, Hkv, w = 8192, 32, 8, 8
    g = Hq // Hkv
    q, k8, v8 = make_qkv(1, Hq, S, 128, dt, "sink", 4100 + S)
    k_un = k8[:, ::g].contiguous()
    k_exp = k_un[:, :, None].expand(1, Hkv, g, S, 128).reshape(1, Hq, S, 128).contiguous()
    for red in ("sum", "mean"):
        got = P.ops.score_window(q.cuda(), k_un.cuda(), w, None, 1, red, kv_group=g).cpu()
        want = O.window_scores(q, k_exp, w, red)
        n, exact, near, unexplained, pos = score_bar.explain_window_scores(q, k_exp, w, got, want, red, limit=200)
        print(json.dumps({"explain_%s_%s" % (dt, red): {"beyond_1ulp": n, "reproduced_exactly": exact, "reproduced_within_1ulp": near, "unexplained": unexplained[:5]}}))
