"""Generate golden fixtures by running the REAL reference (/root/reference) on CPU.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference cannot travel to the GPU box, so its outputs are committed here as small
fixtures.  Inputs are NOT stored: they are re-derived from (seed, shape, kind) with the
deterministic CPU generator in ``tests/inputs.py``; each fixture carries an input checksum
so RNG drift is detected rather than silently mis-compared.

Each fixture holds, per case: the config, the reference's compacted K/V bit patterns, the
indices the reference picked (recovered by matching gathered K rows to source rows), and
for AdaKV/HeadKV the var-len metadata.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

from inputs import make_qkv, bits, checksum  # noqa: E402
from pyramidkv import pyramidkv_utils as ref  # noqa: E402  (the real reference)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def recover_indices(K, Kc_past):
    """K [B,H,L,D] source rows, Kc_past [B,H,k,D] gathered rows -> int32 [B,H,k]."""
    B, H, L, D = K.shape
    k = Kc_past.shape[2]
    out = np.zeros((B, H, k), dtype=np.int32)
    Kb, Cb = bits(K), bits(Kc_past)
    for b in range(B):
        for h in range(H):
            table = {}
            for s in range(L):
                table.setdefault(Kb[b, h, s].tobytes(), s)
            for j in range(k):
                out[b, h, j] = table[Cb[b, h, j].tobytes()]
    return out


CASES = []


def case(**kw):
    CASES.append(kw)


# policy, dtype, kind(gauss|lattice), B,H,S, w, cap, ks, pool, extra
for dt in ("bf16", "fp16"):
    for pool, ks in (("maxpool", 7), ("avgpool", 5)):
        case(policy="snapkv", dtype=dt, kind="gauss", B=1, H=4, S=512, w=8, cap=64, ks=ks, pool=pool, seed=11)
        case(policy="snapkv", dtype=dt, kind="lattice", B=2, H=2, S=384, w=32, cap=96, ks=ks, pool=pool, seed=12)
# tie-free cases: 64 planted heavy hitters with well separated scores and no pooling plateau (kernel 1), so even the
# reference's CPU topk order is fully determined -> the HIP path must be BIT-IDENTICAL to the real reference here
for dt in ("bf16", "fp16"):
    case(policy="snapkv", dtype=dt, kind="planted", B=1, H=2, S=2048, w=8, cap=40, ks=1, pool="avgpool", seed=91, tie_free=True)
    case(policy="pyramidkv", dtype=dt, kind="planted", B=1, H=2, S=2048, w=8, cap=24, ks=1, pool="maxpool", seed=92,
         layers=32, layer=5, tie_free=True)
case(policy="snapkv", dtype="fp32", kind="gauss", B=1, H=2, S=300, w=8, cap=40, ks=7, pool="maxpool", seed=13)
case(policy="snapkv", dtype="bf16", kind="gauss", B=1, H=2, S=48, w=8, cap=64, ks=7, pool="maxpool", seed=14)  # passthrough
for layer in (0, 15, 31):
    case(policy="pyramidkv", dtype="bf16", kind="gauss", B=1, H=4, S=1024, w=8, cap=64, ks=7, pool="maxpool",
         seed=21, layers=32, layer=layer)
case(policy="pyramidkv", dtype="fp16", kind="gauss", B=1, H=2, S=100, w=8, cap=64, ks=5, pool="avgpool",
     seed=22, layers=32, layer=3)      # 'snap' branch: cap <= S < 2(cap-w)
case(policy="pyramidkv", dtype="bf16", kind="gauss", B=1, H=2, S=115, w=8, cap=64, ks=7, pool="maxpool",
     seed=23, layers=32, layer=31)     # clamp branch: max_num >= S-w
case(policy="pyramidkv", dtype="bf16", kind="gauss", B=1, H=2, S=115, w=8, cap=64, ks=7, pool="maxpool",
     seed=24, layers=32, layer=0)      # clamp branch, k = S-w = L (keep everything)
case(policy="h2o", dtype="bf16", kind="gauss", B=1, H=2, S=256, w=8, cap=48, ks=0, pool="none", seed=31)
case(policy="h2o", dtype="fp16", kind="lattice", B=1, H=2, S=192, w=16, cap=40, ks=0, pool="none", seed=32)
case(policy="streamingllm", dtype="bf16", kind="gauss", B=2, H=2, S=256, w=60, cap=64, ks=0, pool="none", seed=41)
case(policy="adakv", dtype="bf16", kind="gauss", B=1, H=4, S=512, w=8, cap=64, ks=7, pool="maxpool", seed=51,
     floor=0.2, normalize=True)
case(policy="adakv", dtype="fp16", kind="gauss", B=1, H=4, S=384, w=32, cap=96, ks=5, pool="avgpool", seed=52,
     floor=0.5, normalize=False)
case(policy="adakv", dtype="bf16", kind="gauss", B=1, H=2, S=40, w=8, cap=64, ks=7, pool="maxpool", seed=53,
     floor=0.2, normalize=True)        # base_capacity > L: not compressed
case(policy="headkv", dtype="bf16", kind="gauss", B=1, H=4, S=512, w=8, cap=64, ks=7, pool="maxpool", seed=61,
     head_capacity=[[10, 70, 56, 33]], layer=0)


# LOOK-M pivot merge (merge_kv, :119-170) behind every dense policy.  Appended at the END so that the numbering of the
# earlier fixtures never moves.  tie_free = planted heavy hitters, kernel 1: the selection is fully determined, so the
# merged K/V of the REAL reference are comparable bit for bit with any backend.
for dt in ("bf16", "fp16"):
    case(policy="snapkv", dtype=dt, kind="planted", B=1, H=2, S=2048, w=8, cap=40, ks=1, pool="avgpool", seed=191, tie_free=True,
         merge="pivot")
case(policy="pyramidkv", dtype="bf16", kind="planted", B=1, H=2, S=2048, w=8, cap=24, ks=1, pool="maxpool", seed=192,
     layers=32, layer=5, tie_free=True, merge="pivot")
case(policy="snapkv", dtype="bf16", kind="gauss", B=2, H=2, S=512, w=8, cap=64, ks=7, pool="maxpool", seed=193, merge="pivot")
case(policy="snapkv", dtype="fp16", kind="lattice", B=1, H=4, S=384, w=32, cap=96, ks=5, pool="avgpool", seed=194, merge="pivot")
case(policy="h2o", dtype="bf16", kind="gauss", B=1, H=2, S=256, w=8, cap=48, ks=0, pool="none", seed=195, merge="pivot")
case(policy="streamingllm", dtype="bf16", kind="gauss", B=2, H=2, S=256, w=60, cap=64, ks=0, pool="none", seed=196, merge="pivot")

# Round-2 additions (appended: the numbering above never moves): fp32 tensors through more policies, and tiny prompts where
# the window is most of the prompt (S < w + 12: where H2O's statistics once produced NaN; a pyramid layer with 0 past tokens)
case(policy="snapkv", dtype="fp32", kind="lattice", B=2, H=2, S=384, w=32, cap=96, ks=5, pool="avgpool", seed=201)
case(policy="pyramidkv", dtype="fp32", kind="planted", B=1, H=2, S=2048, w=8, cap=24, ks=1, pool="maxpool", seed=202,
     layers=32, layer=5, tie_free=True)
case(policy="streamingllm", dtype="fp32", kind="gauss", B=1, H=2, S=200, w=16, cap=48, ks=0, pool="none", seed=203)
case(policy="h2o", dtype="bf16", kind="gauss", B=1, H=2, S=20, w=16, cap=18, ks=0, pool="none", seed=204)
case(policy="h2o", dtype="fp16", kind="gauss", B=1, H=2, S=70, w=64, cap=67, ks=0, pool="none", seed=205)
case(policy="snapkv", dtype="bf16", kind="gauss", B=1, H=2, S=9, w=8, cap=9, ks=7, pool="maxpool", seed=206)
case(policy="pyramidkv", dtype="bf16", kind="gauss", B=1, H=2, S=40, w=32, cap=33, ks=5, pool="avgpool", seed=207,
     layers=2, layer=1)                # the last of two layers gets 0 past tokens: topk(0), the window alone
case(policy="adakv", dtype="bf16", kind="gauss", B=1, H=4, S=50, w=8, cap=24, ks=7, pool="maxpool", seed=208,
     floor=0.2, normalize=True)        # M = min(L, H*base) = L: every head's whole order decides the budgets
case(policy="adakv", dtype="fp16", kind="gauss", B=1, H=8, S=130, w=64, cap=70, ks=5, pool="avgpool", seed=209,
     floor=0.0, normalize=True)        # window 64, base capacity 6, no floor
case(policy="headkv", dtype="bf16", kind="gauss", B=1, H=4, S=60, w=8, cap=24, ks=7, pool="maxpool", seed=210,
     head_capacity=[[1, 52, 60, 7]], layer=0)   # a capacity above L is cut to L (:855), another is a single token


def run_case(c):
    q, k, v = make_qkv(c["B"], c["H"], c["S"], 128, c["dtype"], c["kind"], c["seed"])
    w, cap = c["w"], c["cap"]
    pol = c["policy"]
    meta = {}
    if pol == "snapkv":
        cl = ref.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=c["ks"], pooling=c["pool"], merge=c.get("merge"))
        kc, vc = quiet(cl.update_kv, k, q, v, None, 1)
    elif pol == "pyramidkv":
        cl = ref.PyramidKVCluster(num_hidden_layers=c["layers"], layer_idx=c["layer"], window_size=w,
                                  max_capacity_prompt=cap, kernel_size=c["ks"], pooling=c["pool"], merge=c.get("merge"))
        kc, vc = quiet(cl.update_kv, k, q, v, None, 1)
    elif pol == "h2o":
        cl = ref.H2OKVCluster(window_size=w, max_capacity_prompt=cap, merge=c.get("merge"))
        kc, vc = quiet(cl.update_kv, k, q, v, None, 1)
    elif pol == "streamingllm":
        cl = ref.StreamingLLMKVCluster(window_size=w, max_capacity_prompt=cap, merge=c.get("merge"))
        kc, vc = quiet(cl.update_kv, k, q, v, None, 1)
    elif pol in ("adakv", "headkv"):
        if pol == "adakv":
            cl = ref.AdaKVCluster(window_size=w, kernel_size=c["ks"], pooling=c["pool"], max_capacity_prompt=cap,
                                  floor=c["floor"], normalize=c["normalize"], layer_idx=0, num_hidden_layers=32)
        else:
            cl = ref.HeadKVCluster(window_size=w, kernel_size=c["ks"], pooling=c["pool"], max_capacity_prompt=cap,
                                   layer_idx=c["layer"], num_hidden_layers=32, head_capacity=c["head_capacity"])
        kc, vc = quiet(cl.update_kv, k, q, v)
        meta = dict(head_lens=cl.head_lens.numpy(), cu_klen=cl.cu_klen.numpy(), cu_qlen=cl.cu_qlen.numpy(),
                    cu_offset=cl.cu_offset.numpy(), cu_head_offset=cl.cu_head_offset.numpy(),
                    max_seqlen_k=np.int64(cl.max_seqlen_k), klen_sum=np.int64(cl.klen_sum))
    else:
        raise ValueError(pol)
    out = dict(kc=bits(kc), vc=bits(vc), in_checksum=np.int64(checksum(q, k, v)), **meta)
    passthrough = kc is k
    out["passthrough"] = np.bool_(passthrough)
    if pol in ("adakv", "headkv"):
        if kc.shape[0] != c["H"] * c["S"]:
            hl = meta["head_lens"]
            idx, off = [], 0
            for h in range(c["H"]):
                n = int(hl[h]) - w
                rows = kc[off:off + n][None, None]
                idx.append(recover_indices(k[:, h:h + 1, :-w], rows)[0, 0])
                off += int(hl[h])
            out["idx_flat"] = np.concatenate(idx).astype(np.int32)
    elif not passthrough and not c.get("merge"):
        out["idx"] = recover_indices(k[:, :, :-w], kc[:, :, :-w])      # merged rows are no copies of source rows: no indices
    return out


def head_capacity_fixture():
    """HeadKV budgets from the reference's own Llama-3 head-score table via the runner's arithmetic
    (run_longbench.py:225-234 is inline in main(), so it is restated in oracle.headkv_runner_capacity; the fixture
    pins the per-head means of the real table and the budgets for the runner defaults)."""
    import json as _json
    from oracle import pkv_oracle as O
    path = "/root/reference/data/heads_score/Meta-Llama-3-8B-Instruct_retrieval_reasoning_heads.json"
    with open(path) as f:
        head_list = _json.loads(f.readline())
    out = {"means": np.asarray([np.mean(v) for v in head_list.values()], dtype=np.float64)}
    for cap in (128, 2048):
        out[f"cap{cap}"] = O.headkv_runner_capacity(head_list, 32, 32, cap, 1.01).numpy()
    np.savez_compressed(os.path.join(HERE, "headkv_capacity_llama3.npz"), **out)
    print("headkv_capacity_llama3", {k: v.shape for k, v in out.items()})


def main():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    index = []
    force = "--force" in sys.argv
    for i, c in enumerate(CASES):
        name = f"{i:02d}_{c['policy']}_{c['dtype']}_{c['kind']}_S{c['S']}" + ("_merge" if c.get("merge") else "")
        index.append(dict(name=name, **c))
        if os.path.exists(os.path.join(HERE, name + ".npz")) and not force:
            continue                       # fixtures already committed stay byte-identical (re-create with --force)
        out = run_case(c)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: getattr(v, 'shape', v) for k, v in out.items() if k not in ('kc', 'vc')})
    if force or not os.path.exists(os.path.join(HERE, "headkv_capacity_llama3.npz")):
        head_capacity_fixture()
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(dict(torch=torch.__version__, reference="Zefan-Cai/PyramidKV@2024-12-20",
                       cases=index), f, indent=1)


if __name__ == "__main__":
    main()
