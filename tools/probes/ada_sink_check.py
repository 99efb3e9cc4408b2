import sys, os, json, torch
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import pyramidkv_amd as P
from inputs import make_qkv
from oracle import pkv_oracle as O
DEV="cuda"
res={}
for dt in ("fp16","bf16"):
    for S in (8192, 32768):
        for cap, norm in ((128, True), (128, False), (2048, True)):
            Hq,Hkv,w=32,8,8
            g=Hq//Hkv
            q,k8,v8 = make_qkv(1,Hq,S,128,dt,"sink",4100+S)
            k_un,v_un = k8[:, ::g].contiguous(), v8[:, ::g].contiguous()
            k_exp = k_un[:, :, None].expand(1,Hkv,g,S,128).reshape(1,Hq,S,128).contiguous()
            v_exp = v_un[:, :, None].expand(1,Hkv,g,S,128).reshape(1,Hq,S,128).contiguous()
            cl = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2, normalize=norm, layer_idx=0, num_hidden_layers=32)
            try:
                kf,vf = cl.update_kv(k_un.to(DEV), q.to(DEV), v_un.to(DEV))
                kf2,vf2 = cl.update_kv(k_un.to(DEV), q.to(DEV), v_un.to(DEV))
                kr,vr,meta = O.adakv_update_kv(k_exp, q, v_exp, w, cap, 7, "maxpool", 0.2, norm)
                same_lens = cl.head_lens.cpu().tolist() == meta.head_lens.tolist()
                same_kv = same_lens and bool(torch.equal(kf.cpu(),kr) and torch.equal(vf.cpu(),vr))
                res[f"{dt}_S{S}_cap{cap}_norm{norm}"]=dict(lens=same_lens, kv=same_kv, second_call_same=bool(torch.equal(kf,kf2)), repeats=cl.ada.repeats,
                    hl=cl.head_lens.cpu().tolist()[:8], ref=meta.head_lens.tolist()[:8])
            except Exception as e:
                res[f"{dt}_S{S}_cap{cap}_norm{norm}"]=repr(e)[:300]
print(json.dumps(res, indent=1))

# ---- where K/V differ: per head, do the two flat segments hold the same ROWS (as multisets)?  how many positions differ? ----
detail = {}
for dt, S, cap in (("fp16", 8192, 2048), ("fp16", 32768, 2048)):
    Hq, Hkv, w = 32, 8, 8
    g = Hq // Hkv
    q, k8, v8 = make_qkv(1, Hq, S, 128, dt, "sink", 4100 + S)
    k_un, v_un = k8[:, ::g].contiguous(), v8[:, ::g].contiguous()
    k_exp = k_un[:, :, None].expand(1, Hkv, g, S, 128).reshape(1, Hq, S, 128).contiguous()
    v_exp = v_un[:, :, None].expand(1, Hkv, g, S, 128).reshape(1, Hq, S, 128).contiguous()
    cl = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2, normalize=True, layer_idx=0, num_hidden_layers=32)
    kf, vf = cl.update_kv(k_un.to(DEV), q.to(DEV), v_un.to(DEV))
    kr, vr, meta = O.adakv_update_kv(k_exp, q, v_exp, w, cap, 7, "maxpool", 0.2, True)
    sg = P.ops.score_window(q.to(DEV), k_un.to(DEV), w, "maxpool", 7, "mean", kv_group=g).cpu()[0]
    so = O.pool_scores(O.window_scores(q, k_exp, w, "mean"), "maxpool", 7)[0]
    cu = meta.cu_klen.tolist()
    kfc = kf.cpu()
    heads = []
    for h in range(Hq):
        a, b = kfc[cu[h]:cu[h + 1]].view(torch.int16), kr[cu[h]:cu[h + 1]].view(torch.int16)
        if torch.equal(a, b):
            continue
        key = lambda t: sorted(map(bytes, t.numpy()))
        heads.append({"head": h, "rows": int(a.shape[0]), "positions_differ": int((a != b).any(-1).sum()), "same_rows_as_multiset": key(a) == key(b),
                      "scores_differ_in_head": int((sg[h].view(torch.int16) != so[h].view(torch.int16)).sum())})
    detail["%s_S%d_cap%d" % (dt, S, cap)] = heads
print(json.dumps(detail, indent=1))
