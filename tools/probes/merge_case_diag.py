"""Diagnose one LOOK-M merge case of tools/parity_fuzz.py: run pkv_merge_compact, read the kernel's drop list and pivots out of
the workspace (layout of pkv_api.hip merge_ws), and compare them with the oracle's similarity matrix (merge_kv_explicit's
arithmetic, vectorised): for every dropped row whose pivot differs - the oracle's similarity at both pivots, the exact (fp64) dot
products behind them, and whether the oracle's output is reproduced when the kernel's pivots are used.
  python tools/probes/merge_case_diag.py B H G S w dtype kind pool ks k qkv_seed"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyramidkv_amd as P
from pyramidkv_amd import ops
from inputs import make_qkv
from oracle import pkv_oracle as O
B, H, G, S, w = (int(x) for x in sys.argv[1:6])
dt, kind, pool, ks, kk, seed = sys.argv[6], sys.argv[7], sys.argv[8], int(sys.argv[9]), int(sys.argv[10]), int(sys.argv[11])
q, k, v = make_qkv(B, H, S, 128, dt, kind, seed)
ku, vu = k[:, ::G].contiguous(), v[:, ::G].contiguous()
ke, ve = ku.repeat_interleave(G, dim=1), vu.repeat_interleave(G, dim=1)
qd, kd, vd = q.cuda(), ku.cuda(), vu.cuda()
idx_d = ops.select(qd, kd, w, kk, pool, ks, kv_group=G)
km, vm = ops.merge_compact(kd, vd, idx_d, w, kv_group=G)
torch.cuda.synchronize()
ws = ops.workspace(1, kd.device).cpu().numpy()
idx = idx_d.cpu().long()
al = lambda x, a=256: (x + a - 1) // a * a
o = 0
off_mask = o; o = al(o + S)
off_bad = o; o = al(o + B * H * 4)
off_n = o; o = al(o + 4)
off_drop = o; o = al(o + S * 4)
off_pivot = o; o = al(o + B * H * S * 4)
ndrop = int(ws[off_n:off_n + 4].view(np.int32)[0])
drop_k = ws[off_drop:off_drop + 4 * ndrop].view(np.int32).astype(np.int64)
piv_k = ws[off_pivot:off_pivot + 4 * B * H * S].view(np.int32).reshape(B * H, S)[:, :ndrop].astype(np.int64)
kmr, vmr = O.merge_kv(ke, ve, idx, w, "pivot")
res = {"ndrop": ndrop, "k_equal": bool(torch.equal(km.cpu(), kmr)), "v_equal": bool(torch.equal(vm.cpu(), vmr)), "heads": []}
union = set(idx.flatten().tolist())
drop = [p for p in range(S) if p not in union]
res["drop_list_equal"] = drop == drop_k.tolist()
tdt = ke.dtype
rnd = lambda x: x.to(tdt).float()
piv_ref = O.merge_pivots(ke, ve, idx, w).numpy()
from merge_bar import check_merge
try:
    res["check_merge_moved"] = check_merge(ops, ke, ve, idx, w, km, vm, "diag")
except AssertionError as e:
    res["check_merge_assertion"] = str(e.args)[:300]
for b in range(B):
    for h in range(H):
        Kf = ke[b, h].float()
        sel = idx[b, h]
        tgt = torch.cat([Kf[S - w:], Kf[sel]], 0)
        unit = lambda X: rnd(X / rnd(torch.sqrt((X * X).sum(-1)))[:, None])
        ud, ut = unit(Kf[drop]), unit(tgt)
        sim = rnd(ud @ ut.T)                                     # an fp32 replica of the similarities (for the printed values only)
        piv_o = piv_ref[b, h]                                    # the reference's own pivots (ATen's model-dtype matmul)
        pk = piv_k[b * H + h]
        diff = np.nonzero(piv_o != pk)[0]
        rows = []
        for i in diff[:12]:
            jo, jk = int(piv_o[i]), int(pk[i])
            ex_o = float((ud[i].double() * ut[jo].double()).sum()); ex_k = float((ud[i].double() * ut[jk].double()).sum())
            rows.append({"drop_row": int(drop[i]), "oracle_pivot": jo, "kernel_pivot": jk, "oracle_sim_at_its_pivot": float(sim[i, jo]),
                         "oracle_sim_at_kernel_pivot": float(sim[i, jk]), "exact_dot_at_oracle_pivot": ex_o, "exact_dot_at_kernel_pivot": ex_k,
                         "exact_rounded_o": float(torch.tensor(ex_o).to(tdt)), "exact_rounded_k": float(torch.tensor(ex_k).to(tdt))})
        res["heads"].append({"b": b, "h": h, "pivots_differ": int(len(diff)), "of": int(len(pk)), "examples": rows,
                             "k_rows_differ": int((km[b, h].cpu() != kmr[b, h]).any(-1).sum()), "v_rows_differ": int((vm[b, h].cpu() != vmr[b, h]).any(-1).sum())})
print(json.dumps(res, indent=1))

# ---- rows whose K or V differ although every pivot agrees: the scatter-mean arithmetic itself ----
detail = []
for b in range(B):
    for h in range(H):
        dk = (km[b, h].cpu().view(torch.int16) != kmr[b, h].view(torch.int16))
        dv = (vm[b, h].cpu().view(torch.int16) != vmr[b, h].view(torch.int16))
        for j in torch.nonzero(dk.any(-1) | dv.any(-1)).flatten().tolist()[:4]:
            pk = piv_k[b * H + h]
            detail.append({"b": b, "h": h, "kept_row": j, "k_elems_differ": int(dk[j].sum()), "v_elems_differ": int(dv[j].sum()),
                           "rows_merged_into_it_key_order": int((pk == j).sum()),
                           "first_k_diff": [(int(e), float(km[b, h, j, e]), float(kmr[b, h, j, e])) for e in torch.nonzero(dk[j]).flatten().tolist()[:3]],
                           "first_v_diff": [(int(e), float(vm[b, h, j, e]), float(vmr[b, h, j, e])) for e in torch.nonzero(dv[j]).flatten().tolist()[:3]]})
print(json.dumps({"rows_with_differences": detail[:12]}, indent=1))
