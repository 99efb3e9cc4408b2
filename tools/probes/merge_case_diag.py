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
for b in range(B):
    for h in range(H):
        Kf = ke[b, h].float()
        sel = idx[b, h]
        tgt = torch.cat([Kf[S - w:], Kf[sel]], 0)
        unit = lambda X: rnd(X / rnd(torch.sqrt((X * X).sum(-1)))[:, None])
        ud, ut = unit(Kf[drop]), unit(tgt)
        sim = rnd(ud @ ut.T)                                     # the oracle's similarities (ATen CPU GEMM order)
        piv_o = (sim == sim.max(-1, keepdim=True).values).float().argmax(-1).numpy()
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
