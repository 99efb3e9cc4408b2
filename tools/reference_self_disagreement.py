#!/usr/bin/env python
"""How much does the REFERENCE disagree with itself?  The reference's scores are floating point: ATen's CPU kernels pick their
GEMM blocking and their softmax / sum order by the vector ISA they were dispatched for, so the same update_kv on the same
tensors gives different last-place roundings - and, through them, different top-k orders - on different hosts.  This tool
runs the reference's op sequence (the real pyramidkv/pyramidkv_utils.py when /root/reference exists, else the oracle
restatement, which is bit-identical to it on the 45 golden fixtures) under ATEN_CPU_CAPABILITY = default (avx512 here),
avx2 and "default" (no vector ISA), on one seeded input, and reports the fraction of pooled scores that differ and the
fraction of heads whose selected index SET / SEQUENCE is the same.  That spread is the floor of "identical to the
reference" for everything downstream of a floating-point score; CPU only, no GPU needed.

  python tools/reference_self_disagreement.py [--seq 8192] [--heads 32] [--budget 2048] [--dtype bf16] [--out file.json]
"""
import argparse, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(a):
    import torch
    sys.path.insert(0, ROOT)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
    g = torch.Generator().manual_seed(a.seed)
    q, k, v = (torch.randn(1, a.heads, a.seq, 128, generator=g).to(dt) for _ in range(3))
    use_ref = os.path.isdir("/root/reference/pyramidkv")
    import contextlib, io
    if use_ref:
        sys.path.insert(0, "/root/reference")
        from pyramidkv.pyramidkv_utils import SnapKVCluster
        import torch.nn.functional as F
        import math
        w = 8
        # the scores exactly as pyramidkv_utils.py:317-333 computes them (update_kv returns K/V only, so the lines are replayed)
        attn = torch.matmul(q[..., -w:, :], k.transpose(2, 3)) / math.sqrt(128)
        mask = torch.full((w, w), torch.finfo(attn.dtype).min)
        mask_cond = torch.arange(mask.size(-1))
        mask.masked_fill_(mask_cond < (mask_cond + 1).view(mask.size(-1), 1), 0)
        attn[:, :, -w:, -w:] += mask[None, None, :, :]
        attn = torch.nn.functional.softmax(attn, dim=-1, dtype=torch.float32).to(q.dtype)
        s = F.max_pool1d(attn[:, :, -w:, :-w].sum(dim=-2), kernel_size=7, padding=3, stride=1)
        with contextlib.redirect_stdout(io.StringIO()):
            kc, vc = SnapKVCluster(window_size=w, max_capacity_prompt=a.budget, kernel_size=7, pooling="maxpool").update_kv(k, q, v, None, 1)
    else:
        from oracle import pkv_oracle as O
        s = O.pool_scores(O.window_scores(q, k, 8), "maxpool", 7)
        kc = None
    # canonical order (value desc, index asc) so that only the SCORES differ between the runs, not the tie rule
    from oracle import pkv_oracle as O
    idx = O.topk_canonical(s, a.budget - 8)
    torch.save({"scores": s, "idx": idx, "kc": kc, "cap": torch.backends.cpu.get_cpu_capability(), "ref": use_ref}, a.save)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=8192)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--budget", type=int, default=2048)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--out", default=None)
    ap.add_argument("--save", default=None)
    a = ap.parse_args()
    if a.save:
        return worker(a)
    import torch
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for cap in ("", "avx2", "default"):
            f = os.path.join(td, "r_%s.pt" % (cap or "native"))
            env = dict(os.environ)
            if cap:
                env["ATEN_CPU_CAPABILITY"] = cap
            else:
                env.pop("ATEN_CPU_CAPABILITY", None)
            subprocess.check_call([sys.executable, __file__, "--seq", str(a.seq), "--heads", str(a.heads), "--budget", str(a.budget),
                                   "--dtype", a.dtype, "--seed", str(a.seed), "--save", f], env=env)
            res[cap or "native"] = torch.load(f)
    base = res["native"]
    out = {"workload": "SnapKV window 8 maxpool-7, [1,%d,%d,128] %s, budget %d, N(0,1) seed %d" % (a.heads, a.seq, a.dtype, a.budget, a.seed),
           "source": "pyramidkv/pyramidkv_utils.py (real reference)" if base["ref"] else "oracle/pkv_oracle.py (restatement)",
           "torch": torch.__version__, "native_capability": base["cap"], "against_native": {}}
    for cap, r in res.items():
        if cap == "native":
            continue
        sd = (r["scores"].view(torch.int16) != base["scores"].view(torch.int16))
        seq = (r["idx"] == base["idx"]).all(-1)
        st = (torch.sort(r["idx"], -1).values == torch.sort(base["idx"], -1).values).all(-1)
        row = {"capability": r["cap"], "pooled_scores_differing_frac": float(sd.float().mean()),
               "heads_same_index_sequence": float(seq.float().mean()), "heads_same_index_set": float(st.float().mean()), "heads": int(seq.numel())}
        if r["kc"] is not None and base["kc"] is not None:
            row["heads_same_update_kv_K_bits"] = float((r["kc"] == base["kc"]).flatten(2).all(-1).float().mean())
        out["against_native"][cap] = row
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
