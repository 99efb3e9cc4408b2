"""Explain one floating-point outlier of tools/parity_fuzz.py: replay the generator up to (seed, case), find the score element
with the largest ulp distance between the HIP path and the oracle, and test the hypothesis "one LOGIT of that position was
rounded to the neighbouring model-dtype value" (accumulation order of q.k): recompute the oracle's probability of that
element with the logit moved by +-1 ulp and compare with what the kernel produced.
  python tools/fuzz_explain.py seed case      (valid while every earlier case of that seed passed: the replay draws like a pass)"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyramidkv_amd as P
from inputs import make_qkv, bits, DTYPES
from oracle import pkv_oracle as O
seed, target = int(sys.argv[1]), int(sys.argv[2])
smax = int(sys.argv[3]) if len(sys.argv) > 3 else 5000          # the third argument of the parity_fuzz.py run being replayed
rng = np.random.RandomState(seed)
n = 0
while True:
    pol = str(rng.choice(["window", "window", "h2o", "adakv", "merge", "pyramid"]))
    S = int(rng.randint(40, smax)) if pol != "h2o" else int(rng.randint(40, 1800))
    w = int(rng.choice([1, 4, 8, 8, 16, 32, 64]))
    if S <= w + 8:
        continue
    G = int(rng.choice([1, 2, 4])); H = G * int(rng.randint(1, 5))
    B = 1 if pol == "adakv" else int(rng.randint(1, 3))
    dt = ("bf16", "fp16")[int(rng.randint(0, 2))]
    kind = ("gauss", "lattice", "planted")[int(rng.randint(0, 3))]
    pool, ks = [("maxpool", 7), ("avgpool", 5), ("maxpool", 17), ("avgpool", 13), (None, 1), ("maxpool", 3)][int(rng.randint(0, 6))]
    L = S - w
    kk = int(rng.randint(1, L + 1)) if rng.rand() < 0.7 else int(rng.choice([1, L, min(L, 512), min(L, 513), min(L, 2040), min(L, 4096), min(L, 4097)]))
    dseed = int(rng.randint(0, 1 << 30))
    if n == target:
        break
    if pol == "pyramid" and pool is not None:
        rng.randint(0, 32)
    if pol == "adakv":
        rng.choice([0.0, 0.2, 0.5, 1.0]); rng.randint(0, 2); rng.choice([0, 1, 2, 8])
    n += 1
print(json.dumps(dict(seed=seed, case=n, pol=pol, B=B, H=H, G=G, S=S, w=w, dt=dt, kind=kind, pool=pool, ks=ks, k=kk)))
q, k, v = make_qkv(B, H, S, 128, dt, kind, dseed)
ku = k[:, ::G].contiguous(); ke = ku.repeat_interleave(G, dim=1)
got = P.ops.score_window(q.cuda(), ku.cuda(), w, None, 1, kv_group=G).cpu()      # un-pooled sums: the stage in question
want = O.window_scores(q, ke, w)


def ord16(t):
    b = bits(t).astype(np.int64)
    return np.where(b & 0x8000, -(b & 0x7FFF), b)


d = np.abs(ord16(got) - ord16(want))
print("un-pooled scores: max ulp", int(d.max()), "elements beyond 1 ulp", int((d > 1).sum()), "of", d.size)
T = DTYPES[dt]
from inputs import from_bits


def neighbour(x, step):                       # next model-dtype value above (step = +1) / below (-1) a finite scalar tensor x
    xb = int(bits(x.reshape(1)).astype(np.int64)[0])
    up = (xb + 1) if xb < 0x8000 else (xb - 1)
    dn = (xb - 1) if 0 < xb < 0x8000 else (xb + 1 if xb >= 0x8000 else 0x8001)
    return from_bits(np.array([up if step > 0 else dn], dtype=np.uint16), dt)[0]


for (b_, h_, j_) in np.argwhere(d > 1)[:6]:
    qw = q[b_, h_, -w:]
    exact = (qw.double() @ ke[b_, h_, j_].double())              # exact q.k of the w window rows with key j (fp64)
    P0 = qw @ ke[b_, h_].transpose(0, 1)                         # the oracle's un-scaled products [w, S], model dtype (:317)

    def head_score(prod):
        A = prod / (128 ** 0.5)
        if w > 1:
            A = A.clone(); A[:, -w:] += torch.triu(torch.full((w, w), torch.finfo(T).min, dtype=A.dtype), diagonal=1)
        return float(torch.softmax(A, dim=-1, dtype=torch.float32).to(T)[:, j_].float().sum().to(T))
    out = dict(pos=[int(b_), int(h_), int(j_)], got=float(got[b_, h_, j_]), want=float(want[b_, h_, j_]),
               oracle_recomputed=head_score(P0), ulp=int(d[b_, h_, j_]))
    hits = []
    for r in range(w):
        lo, hi = neighbour(P0[r, j_], -1), neighbour(P0[r, j_], +1)
        for step, nb in ((-1, lo), (1, hi)):
            P1 = P0.clone(); P1[r, j_] = nb
            if head_score(P1) == out["got"]:
                hits.append(dict(window_row=r, product_in_oracle=float(P0[r, j_]), product_that_reproduces_the_kernel=float(nb)))
        mid_lo, mid_hi = (float(lo) + float(P0[r, j_])) / 2, (float(hi) + float(P0[r, j_])) / 2
        e = float(exact[r])
        dist = min(abs(e - mid_lo), abs(e - mid_hi))
        out.setdefault("exact_product_to_nearest_rounding_midpoint_in_fp32_ulps", []).append(round(dist / (abs(e) * 2.0 ** -24), 2))
    out["one_product_rounded_the_other_way_reproduces_the_kernel"] = bool(hits)
    out["which"] = hits[:2]
    print(json.dumps(out))
