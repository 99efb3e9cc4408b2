"""Parity fuzz: random configurations with a caller-chosen seed, HIP path vs the CPU oracle.  Exact stages (a failure is a
bug): indices == canonical top-k of the kernel's own scores, K/V == the exact gather, Ada-SnapKV budgets == the oracle's
arithmetic on the kernel's scores, LOOK-M merge == the oracle on the selected indices (or explained pivot for pivot by one fp32
unit of a similarity's dot product at a rounding midpoint, tests/merge_bar.py).  Floating-point stage (statistics + a
loose bar): pooled scores vs the oracle - the suite's fixed seeds stay within 1 ulp on <= 0.2 % of the elements; over random
seeds a 1-ulp flip of a LOGIT (fp32 accumulation order of q.k, MFMA vs ATen: ~1e-6 of the elements) moves that probability by
ulp(x) relative = about |x| ulps of the probability, diluted by the window-row sum (w = 8: 1-2 ulp of the score; w = 1: up to
|x| ~ 3-8), so this tool fails a case only beyond 8 ulp or 2 % of the elements and reports how often 1 ulp / 0.5 % were exceeded.  The suite's randomised tests use fixed seeds; this walks new ones.
  python tools/parity_fuzz.py [seconds] [seed] [longest prompt, default 5000]"""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyramidkv_amd as P
from inputs import make_qkv, bits
from merge_bar import check_merge
from score_bar import check_window_scores
from oracle import pkv_oracle as O
DEV = "cuda"


def ord16(t):
    b = bits(t).astype(np.int32)
    return np.where(b & 0x8000, -(b & 0x7FFF), b)


def score_diff(a, b):
    d = np.abs(ord16(a) - ord16(b))
    return float((d > 0).mean()), int(d.max())


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
smax = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
rng = np.random.RandomState(seed)
t0, n, fails, kinds = time.time(), 0, [], {}
fp = dict(score_checks=0, over_1ulp=0, over_half_percent=0, max_ulp=0, max_frac=0.0)


def fp_stats(got, want):
    frac, mx = score_diff(got, want)
    fp["score_checks"] += 1; fp["over_1ulp"] += mx > 1; fp["over_half_percent"] += frac > 5e-3
    fp["max_ulp"] = max(fp["max_ulp"], mx); fp["max_frac"] = max(fp["max_frac"], frac)
    return frac, mx


def fp_check(got, want, what):
    frac, mx = fp_stats(got, want)
    assert mx <= 8 and frac <= 2e-2, (what, frac, mx)
while time.time() - t0 < budget:
    pol = str(rng.choice(["window", "window", "h2o", "adakv", "merge", "pyramid"]))
    S = int(rng.randint(40, smax)) if pol != "h2o" else int(rng.randint(40, 1800))
    w = int(rng.choice([1, 4, 8, 8, 16, 32, 64]))
    if S <= w + 8:
        continue
    G = int(rng.choice([1, 2, 4]))
    H = G * int(rng.randint(1, 5))
    B = 1 if pol == "adakv" else int(rng.randint(1, 3))
    dt = ("bf16", "fp16")[int(rng.randint(0, 2))]
    kind = ("gauss", "lattice", "planted")[int(rng.randint(0, 3))]
    pool, ks = [("maxpool", 7), ("avgpool", 5), ("maxpool", 17), ("avgpool", 13), (None, 1), ("maxpool", 3)][int(rng.randint(0, 6))]
    L = S - w
    kk = int(rng.randint(1, L + 1)) if rng.rand() < 0.7 else int(rng.choice([1, L, min(L, 512), min(L, 513), min(L, 2040), min(L, 4096), min(L, 4097)]))
    q, k, v = make_qkv(B, H, S, 128, dt, kind, int(rng.randint(0, 1 << 30)))
    ku, vu = k[:, ::G].contiguous(), v[:, ::G].contiguous()
    ke, ve = ku.repeat_interleave(G, dim=1), vu.repeat_interleave(G, dim=1)
    qd, kd, vd = q.to(DEV), ku.to(DEV), vu.to(DEV)
    tag = dict(seed=seed, case=n, pol=pol, B=B, H=H, G=G, S=S, w=w, dt=dt, kind=kind, pool=pool, ks=ks, k=kk)
    try:
        if pol in ("window", "pyramid", "merge"):
            want = O.pool_scores(O.window_scores(q, ke, w), pool, ks)
            got = P.ops.score_window(qd, kd, w, pool, ks, kv_group=G).cpu()
            fp_stats(got, want)
            # the suite's bar (tests/score_bar.py): within a unit, or explained by one moved product; no bar on the FRACTION of
            # scores that differ by their allowed unit - it has no principled bound (window 4, fp16, outlier inputs reach 5 %)
            check_window_scores(q, ke, w, pool, ks, "sum", got, lambda: P.ops.score_window(qd, kd, w, None, 1, kv_group=G).cpu(), frac_bar=1.0, what="scores")
            kc, vc, idx = P.ops.compress(qd, kd, vd, w, kk, pool, ks, kv_group=G, return_indices=True)
            idx = idx.cpu().long()
            assert torch.equal(idx, O.topk_canonical(got, kk)), "indices"
            kr, vr = O.gather_compact(ke, ve, idx, w)
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), "gather"
            if pol == "pyramid" and pool is not None:
                layer = int(rng.randint(0, 32))
                cap = w + kk
                cl = P.PyramidKVCluster(num_hidden_layers=32, layer_idx=layer, window_size=w, max_capacity_prompt=cap, kernel_size=ks, pooling=pool)
                kc, vc = cl.update_kv(kd, qd, vd, None, G)
                kr, vr = O.pyramidkv_update_kv(ke, q, ve, w, cap, ks, pool, 32, layer)
                assert kc.shape == kr.shape, ("pyramid shape", tuple(kc.shape), tuple(kr.shape))
            if pol == "merge" and pool is not None and kk + w <= 4000:
                km, vm = P.ops.merge_compact(kd, vd, P.ops.select(qd, kd, w, kk, pool, ks, kv_group=G), w, kv_group=G)
                fp["merge_pivots_moved"] = fp.get("merge_pivots_moved", 0) + check_merge(P.ops, ke, ve, idx, w, km, vm, "merge")   # tests/merge_bar.py
        elif pol == "h2o":
            want = O.h2o_scores(q, ke, w)
            got = P.ops.score_h2o(qd, kd, w, kv_group=G).cpu()
            msk = want.float().abs() >= 1e-35
            fp_check(got[msk], want[msk], "h2o scores")
            kc, vc, idx = P.ops.compress(qd, kd, vd, w, kk, None, 1, kv_group=G, h2o=True, return_indices=True)
            assert torch.equal(idx.cpu().long(), O.topk_canonical(got, kk)), "h2o indices"
            kr, vr = O.gather_compact(ke, ve, idx.cpu().long(), w)
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), "h2o gather"
        else:   # adakv
            if pool is None:
                pool, ks = "maxpool", 7
            floor = float(rng.choice([0.0, 0.2, 0.5, 1.0])); norm = bool(rng.randint(0, 2))
            cap = w + max(1, min(kk, L // 2))
            tag.update(floor=floor, norm=norm, cap=cap)
            from pyramidkv_amd import config as cfg
            cfg.ada_short_lists = int(rng.choice([0, 1, 2, 4]))
            tag.update(short=cfg.ada_short_lists)
            cl = P.AdaKVCluster(window_size=w, kernel_size=ks, pooling=pool, max_capacity_prompt=cap, floor=floor, normalize=norm)
            kf, vf = cl.update_kv(kd, qd, vd)
            cfg.ada_short_lists = 4
            sg = P.ops.score_window(qd, kd, w, pool, ks, "mean", kv_group=G).cpu()[0]
            sidx, caps = O.adakv_head_capacity(sg[None], cap - w, floor, norm)
            caps = caps[0].tolist()
            assert cl.head_lens.cpu().tolist() == [c + w for c in caps], "ada budgets"
            rows_k, rows_v = [], []
            for h in range(H):
                ix = sidx[0, h, :caps[h]].long()
                rows_k += [ke[0, h, ix], ke[0, h, L:]]; rows_v += [ve[0, h, ix], ve[0, h, L:]]
            assert torch.equal(kf.cpu(), torch.cat(rows_k)) and torch.equal(vf.cpu(), torch.cat(rows_v)), "ada flat gather"
    except AssertionError as e:
        fails.append(dict(tag, error=str(e.args)))
        print("FAIL", json.dumps(fails[-1]), flush=True)
    except Exception as e:      # noqa: BLE001 - a fuzz run reports and goes on
        fails.append(dict(tag, error=repr(e)))
        print("ERROR", json.dumps(fails[-1]), flush=True)
    n += 1
    kinds[pol] = kinds.get(pol, 0) + 1
print(json.dumps(dict(seed=seed, seconds=round(time.time() - t0, 1), cases=n, by_policy=kinds, failures=len(fails), floating_point=fp)))
sys.exit(1 if fails else 0)
