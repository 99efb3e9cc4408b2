"""Which window row of head 29 (fp16 sink inputs, S = 8192) differs between libpkv and the oracle, and by what factor?  Each window
row r is scored alone (w = 1 on a query tensor whose last row is row r)."""
import sys, os, json, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyramidkv_amd as P
from inputs import make_qkv, bits
from oracle import pkv_oracle as O
S, Hq, Hkv, w = 8192, 32, 8, 8
g = Hq // Hkv
q, k8, v8 = make_qkv(1, Hq, S, 128, "fp16", "sink", 4100 + S)
k_un = k8[:, ::g].contiguous()
k_exp = k_un[:, :, None].expand(1, Hkv, g, S, 128).reshape(1, Hq, S, 128).contiguous()
out = []
for r in range(w):
    q1 = q.clone()
    q1[:, :, -1] = q[:, :, S - w + r]
    got = P.ops.score_window(q1.cuda(), k_un.cuda(), 1, None, 1, "sum", kv_group=g).cpu()[0, 29]
    want = O.window_scores(q1, k_exp, 1, "sum")[0, 29]
    d = (bits(got).astype(np.int32) - bits(want).astype(np.int32))
    nz = want.float() > 1e-6
    ratio = (got.float()[nz] / want.float()[nz]).numpy()
    # the oracle's own logits of this row: the maximum and how close its exact product is to a rounding midpoint
    qrow, kk = q1[0, 29, -1].double(), k_exp[0, 29].double()
    ex = (kk @ qrow)
    jm = int(ex.argmax())
    prod16 = torch.matmul(q1[0, 29, -1:].float(), k_exp[0, 29].float().T).to(torch.float16)[0, jm]
    out.append({"row": r, "mismatches": int((d != 0).sum()), "max_abs_ulp": int(np.abs(d).max()), "median_ratio_got_over_want": float(np.median(ratio)) if len(ratio) else None,
                "argmax_col": jm, "exact_product": float(ex[jm]), "fp16_product_cpu": float(prod16), "fp16_spacing_there": float(np.spacing(np.float16(abs(float(prod16)))))})
print(json.dumps(out, indent=1))
