"""Three-way parity on ONE chip (SURVEY.md section 7 hard part 1 / section 8c: "exact-sequence match vs device reference;
exact-set match; value-multiset vs CPU reference").  TEST INFRASTRUCTURE: imported by tests/, tools/parity_three_way.py
and bench.py's parity blocks only - never by the product.

Three implementations of the same ``update_kv`` on the same tensors:

  hip     libpkv through the C ABI (the product)
  cpu     the reference's op sequence executed by PyTorch on the host CPU (oracle/pkv_oracle.py; == the real
          pyramidkv_utils.py bit for bit, tests/golden) - canonical tie order
  eager   the SAME op sequence executed by PyTorch-ROCm eager on the same MI355X (the oracle functions on HIP tensors;
          ``tensor.topk`` exactly as pyramidkv_utils.py:334 calls it) - what a maintainer who runs the reference on this
          GPU gets

and three pairs, each reported as heads with the identical index SET / index SEQUENCE / compacted K and V bits plus the
histogram of score differences in units of the last place (ulp) of the model dtype:

  hip_vs_cpu     libpkv with its defaults (scale "div", tie order "canonical") against the CPU reference
  hip_vs_eager   libpkv with PKV_SCALE_MODE=rcp + tie_order=aten_rocm (ATen's GPU kernels multiply by the fp32 reciprocal of
                 sqrt(D) and leave the ties of a k <= 32 selection in their own order) against the device reference
  eager_vs_cpu   the reference against itself across backends: the floor no implementation can be asked to beat

Reference lines: pyramidkv_utils.py:317-346 (SnapKV), :205-283 (PyramidKV budgets), :674-757 (Ada-SnapKV).
"""
import contextlib
import io
import os

import torch

from oracle import pkv_oracle as O


def _ord16(t: torch.Tensor) -> torch.Tensor:
    """monotone integer image of a 16-bit float tensor (ulp distances)"""
    b = t.contiguous().view(torch.int16).int() & 0xffff
    return torch.where(b >= 0x8000, -(b & 0x7fff), b)


def ulp_hist(a: torch.Tensor, b: torch.Tensor) -> dict:
    """|a - b| in ulps of the 16-bit model dtype, as counts of 0 / 1 / 2 / >= 3 and the maximum."""
    d = (_ord16(a.cpu()) - _ord16(b.cpu())).abs()
    n = d.numel()
    return {"elements": n, "ulp0": int((d == 0).sum()), "ulp1": int((d == 1).sum()), "ulp2": int((d == 2).sum()),
            "ulp3plus": int((d >= 3).sum()), "max_ulp": int(d.max()) if n else 0,
            "mismatch_frac": round(float((d > 0).float().mean()), 8) if n else 0.0}


def _pair(ia, ka, va, ib, kb, vb) -> dict:
    """heads of a [B,H,k] selection that agree: set / sequence / K,V bits.  Everything on the CPU."""
    ia, ib = ia.cpu().long(), ib.cpu().long()
    seq = (ia == ib).all(-1)
    st = (torch.sort(ia, -1).values == torch.sort(ib, -1).values).all(-1)
    kv = (ka.cpu() == kb.cpu()).flatten(2).all(-1) & (va.cpu() == vb.cpu()).flatten(2).all(-1)
    n = seq.numel()
    return {"heads": n, "identical_set": int(st.sum()), "identical_sequence": int(seq.sum()), "kv_bit_identical": int(kv.sum()),
            "set_rate": float(st.float().mean()), "sequence_rate": float(seq.float().mean()),
            "kv_rate": float(kv.float().mean())}


@contextlib.contextmanager
def hip_knobs(P, scale_mode, tie_order):
    """libpkv's two process-wide knobs (pyramidkv_amd/config.py, read at call time) for the span of one comparison."""
    old = (P.config.scale_mode, P.config.tie_order)
    P.config.scale_mode, P.config.tie_order = scale_mode, tie_order
    try:
        yield
    finally:
        P.config.scale_mode, P.config.tie_order = old


def window_policy(P, q, k, v, w, budgets, pooling="maxpool", kernel_size=7, dev="cuda:0", threads=None):
    """SnapKV / PyramidKV (:317-346) on CPU tensors q, k, v [B,H,S,D]; ``budgets`` = {label: k}.  -> report dict."""
    torch.set_num_threads(threads or min(32, os.cpu_count() or 1))
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    kmax = max(budgets.values())
    with contextlib.redirect_stdout(io.StringIO()):
        s_cpu = O.pool_scores(O.window_scores(q, k, w), pooling, kernel_size)
        s_dev = O.pool_scores(O.window_scores(qd, kd, w), pooling, kernel_size)          # ATen's HIP kernels, same op sequence
    order_cpu = O.topk_canonical(s_cpu, kmax)              # prefix = the canonical top-k of every smaller budget
    s_hip = P.ops.score_window(qd, kd, w, pooling, kernel_size, "sum", "div")
    s_hip_rcp = P.ops.score_window(qd, kd, w, pooling, kernel_size, "sum", "rcp")
    rep = {"shape": list(q.shape), "dtype": str(q.dtype).replace("torch.", ""), "window": w, "pooling": pooling, "kernel_size": kernel_size,
           "scores": {"hip_vs_cpu": ulp_hist(s_hip, s_cpu), "hip_rcp_vs_eager": ulp_hist(s_hip_rcp, s_dev),
                      "eager_vs_cpu": ulp_hist(s_dev, s_cpu), "hip_vs_eager": ulp_hist(s_hip, s_dev)},
           "budgets": {}}
    for label, kk in budgets.items():
        i_cpu = order_cpu[..., :kk]
        k_cpu, v_cpu = O.gather_compact(k, v, i_cpu, w)
        i_dev = O.topk_reference(s_dev, kk)                                               # :334 as the reference calls it, on the device
        k_dev, v_dev = O.gather_compact(kd, vd, i_dev, w)
        with hip_knobs(P, "div", "canonical"):
            k_h, v_h, i_h = P.ops.compress(qd, kd, vd, w, kk, pooling, kernel_size, scale_mode="div", return_indices=True)
        with hip_knobs(P, "rcp", "aten_rocm"):
            k_r, v_r, i_r = P.ops.compress(qd, kd, vd, w, kk, pooling, kernel_size, scale_mode="rcp", return_indices=True)
        rep["budgets"][label] = {"k": kk,
                                 "hip_vs_cpu": _pair(i_h, k_h, v_h, i_cpu, k_cpu, v_cpu),
                                 "hip_vs_eager": _pair(i_r, k_r, v_r, i_dev, k_dev, v_dev),
                                 "eager_vs_cpu": _pair(i_dev, k_dev, v_dev, i_cpu, k_cpu, v_cpu)}
    return rep


def _flat_heads(kf, vf, lens):
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + int(n))
    kf, vf = kf.cpu(), vf.cpu()
    return [(kf[cu[h]:cu[h + 1]], vf[cu[h]:cu[h + 1]]) for h in range(len(lens))]


def _ada_pair(la, fa, lb, fb) -> dict:
    n = len(la)
    same_len = [int(x) == int(y) for x, y in zip(la, lb)]
    kv = [same_len[h] and bool(torch.equal(fa[h][0], fb[h][0]) and torch.equal(fa[h][1], fb[h][1])) for h in range(n)]
    return {"heads": n, "head_budgets_identical": all(same_len), "heads_with_other_budget": n - sum(same_len),
            "kv_bit_identical": sum(kv), "kv_rate": sum(kv) / n,
            "max_budget_difference": max(abs(int(x) - int(y)) for x, y in zip(la, lb))}


def adakv(P, q, k_un, v_un, w, cap, floor=0.2, normalize=True, pooling="maxpool", kernel_size=7, dev="cuda:0", threads=None):
    """Ada-SnapKV (:674-757) with K/V handed to libpkv UN-EXPANDED ([1,Hkv,S,D]) and to both references after repeat_kv."""
    torch.set_num_threads(threads or min(32, os.cpu_count() or 1))
    g = q.shape[1] // k_un.shape[1]
    kx, vx = (t.repeat_interleave(g, dim=1) for t in (k_un, v_un))
    qd, kd, vd = q.to(dev), k_un.to(dev), v_un.to(dev)
    kxd, vxd = kx.to(dev), vx.to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        kc, vc, mc = O.adakv_update_kv(kx, q, vx, w, cap, kernel_size, pooling, floor, normalize)
        # the reference's own calls on the device: sort(descending) at :706 and topk at :713 with ATen's tie order
        ke, ve, me = O.adakv_update_kv(kxd, qd, vxd, w, cap, kernel_size, pooling, floor, normalize, sort_mode="reference")
    f_cpu, f_dev = _flat_heads(kc, vc, mc.head_lens.tolist()), _flat_heads(ke, ve, me.head_lens.cpu().tolist())
    out = {"shape": list(q.shape), "kv_heads": int(k_un.shape[1]), "dtype": str(q.dtype).replace("torch.", ""), "budget": cap}
    for name, (sm, to) in (("hip_vs_cpu", ("div", "canonical")), ("hip_vs_eager", ("rcp", "aten_rocm"))):
        with hip_knobs(P, sm, to):
            cl = P.AdaKVCluster(window_size=w, kernel_size=kernel_size, pooling=pooling, max_capacity_prompt=cap, floor=floor,
                                normalize=normalize, layer_idx=0, num_hidden_layers=32)
            kf, vf = cl.update_kv(kd, qd, vd)
            torch.cuda.synchronize()
            lens = cl.head_lens.cpu().tolist()
        f_hip = _flat_heads(kf, vf, lens)
        ref_l, ref_f = (mc.head_lens.tolist(), f_cpu) if name == "hip_vs_cpu" else (me.head_lens.cpu().tolist(), f_dev)
        out[name] = _ada_pair(lens, f_hip, ref_l, ref_f)
    out["eager_vs_cpu"] = _ada_pair(me.head_lens.cpu().tolist(), f_dev, mc.head_lens.tolist(), f_cpu)
    return out


def summarise(rep: dict) -> dict:
    """One line per (budget, pair) of a window_policy report - what bench.py prints."""
    out = {}
    for label, b in rep["budgets"].items():
        out[label] = {p: {"set": r["set_rate"], "sequence": r["sequence_rate"], "kv": r["kv_rate"]}
                      for p, r in b.items() if isinstance(r, dict)}
    out["score_ulp"] = {p: {"mismatch_frac": r["mismatch_frac"], "max_ulp": r["max_ulp"]} for p, r in rep["scores"].items()}
    return out
