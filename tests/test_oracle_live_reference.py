"""The oracle against the REAL reference, live, on random configurations (CPU; only where /root/reference exists - this
container; the GPU box has neither the reference nor this test's subject, so the module skips there).  The committed
fixtures (tests/golden, test_oracle_golden.py) pin fixed cases; this walks seeded random ones through the same statement:
oracle/pkv_oracle.py with the reference's own topk / sort calls == /root/reference/pyramidkv/pyramidkv_utils.py, bit for
bit - K/V, the flat layout's metadata, pass-through identity, the LOOK-M merge."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

REF_ROOT = "/root/reference"
if not os.path.exists(os.path.join(REF_ROOT, "pyramidkv", "pyramidkv_utils.py")):
    pytest.skip("the reference is not on this machine", allow_module_level=True)
sys.path.insert(0, REF_ROOT)
try:
    from pyramidkv import pyramidkv_utils as ref            # the real reference
except Exception as e:                                       # noqa: BLE001 - a missing optional dependency of the reference
    pytest.skip("the reference does not import here: %r" % (e,), allow_module_level=True)

from inputs import make_qkv, bits                            # noqa: E402
from oracle import pkv_oracle as O                           # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def random_case(rng, policy):
    dt = ("bf16", "fp16", "fp32")[int(rng.integers(0, 3))]
    kind = ("gauss", "lattice", "planted")[int(rng.integers(0, 3))]
    w = int(rng.choice([1, 4, 8, 16, 32]))
    S = int(rng.integers(w + 1, 420))
    c = dict(policy=policy, dtype=dt, kind=kind, B=int(rng.integers(1, 3)), H=int(rng.choice([1, 2, 4, 8])), S=S, w=w,
             cap=w + int(rng.integers(1, 120)), ks=int(rng.choice([1, 3, 5, 7])), pool=("maxpool", "avgpool")[int(rng.integers(0, 2))],
             seed=int(rng.integers(1, 1 << 30)))
    if policy == "pyramidkv":
        c["layers"] = int(rng.choice([2, 8, 32]))
        c["layer"] = int(rng.integers(0, c["layers"]))
    if policy in ("snapkv", "pyramidkv", "h2o", "streamingllm") and rng.random() < 0.3:
        c["merge"] = "pivot"
    if policy in ("adakv", "headkv"):
        c["B"] = 1                                            # the flat layout is single-sequence (:738 asserts)
    if policy == "adakv":
        c["floor"] = float(rng.choice([0.0, 0.2, 0.5, 1.0]))
        c["normalize"] = bool(rng.integers(0, 2))
    if policy == "headkv":
        c["head_capacity"] = [[int(x) for x in rng.integers(1, 140, size=c["H"])]]
    return c


def run_reference(c, q, k, v):
    w, cap, pol, mg = c["w"], c["cap"], c["policy"], c.get("merge")
    if pol == "snapkv":
        cl = ref.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=c["ks"], pooling=c["pool"], merge=mg)
    elif pol == "pyramidkv":
        cl = ref.PyramidKVCluster(num_hidden_layers=c["layers"], layer_idx=c["layer"], window_size=w, max_capacity_prompt=cap,
                                  kernel_size=c["ks"], pooling=c["pool"], merge=mg)
    elif pol == "h2o":
        cl = ref.H2OKVCluster(window_size=w, max_capacity_prompt=cap, merge=mg)
    elif pol == "streamingllm":
        cl = ref.StreamingLLMKVCluster(window_size=w, max_capacity_prompt=cap, merge=mg)
    elif pol == "adakv":
        cl = ref.AdaKVCluster(window_size=w, kernel_size=c["ks"], pooling=c["pool"], max_capacity_prompt=cap, floor=c["floor"],
                              normalize=c["normalize"], layer_idx=0, num_hidden_layers=32)
    else:
        cl = ref.HeadKVCluster(window_size=w, kernel_size=c["ks"], pooling=c["pool"], max_capacity_prompt=cap, layer_idx=0,
                               num_hidden_layers=32, head_capacity=c["head_capacity"])
    if pol in ("adakv", "headkv"):
        kc, vc = quiet(cl.update_kv, k, q, v)
        return kc, vc, cl
    kc, vc = quiet(cl.update_kv, k, q, v, None, 1)
    return kc, vc, cl


def run_oracle(c, q, k, v):
    w, cap, pol, mg = c["w"], c["cap"], c["policy"], c.get("merge")
    if pol == "snapkv":
        return O.snapkv_update_kv(k, q, v, w, cap, c["ks"], c["pool"], topk_mode="reference", merge=mg)
    if pol == "pyramidkv":
        return O.pyramidkv_update_kv(k, q, v, w, cap, c["ks"], c["pool"], c["layers"], c["layer"], topk_mode="reference", merge=mg)
    if pol == "h2o":
        return O.h2o_update_kv(k, q, v, w, cap, topk_mode="reference", merge=mg)
    if pol == "streamingllm":
        return O.streamingllm_update_kv(k, q, v, w, cap, merge=mg)
    if pol == "adakv":
        return O.adakv_update_kv(k, q, v, w, cap, c["ks"], c["pool"], c["floor"], c["normalize"], sort_mode="reference")
    return O.headkv_update_kv(k, q, v, w, cap, c["ks"], c["pool"], c["head_capacity"], 0, sort_mode="reference")


POLICIES = ("snapkv", "pyramidkv", "h2o", "streamingllm", "adakv", "headkv")


@pytest.mark.parametrize("policy", POLICIES)
def test_oracle_equals_the_live_reference_on_random_configurations(policy):
    rng = np.random.default_rng(20260925 + POLICIES.index(policy))
    ran = compressed = 0
    for _ in range(40):
        c = random_case(rng, policy)
        q, k, v = make_qkv(c["B"], c["H"], c["S"], 128, c["dtype"], c["kind"], c["seed"])
        try:
            kr, vr, cl = run_reference(c, q, k, v)
        except Exception as e:                                # noqa: BLE001 - the oracle must refuse what the reference refuses
            with pytest.raises(type(e)):
                run_oracle(c, q, k, v)
            continue
        out = run_oracle(c, q, k, v)
        ko, vo = out[0], out[1]
        ran += 1
        if kr is k:                                           # pass-through returns the caller's objects (:219, :315)
            assert ko is k and vo is v, c
            continue
        compressed += 1
        assert kr.shape == ko.shape and np.array_equal(bits(kr), bits(ko)), c
        assert np.array_equal(bits(vr), bits(vo)), c
        if policy in ("adakv", "headkv"):
            meta = out[2]
            for name in ("head_lens", "cu_klen", "cu_qlen", "cu_offset", "cu_head_offset"):
                assert np.array_equal(getattr(meta, name).numpy(), getattr(cl, name).numpy()), (name, c)
            assert meta.max_seqlen_k == int(cl.max_seqlen_k) and meta.klen_sum == int(cl.klen_sum), c
    assert ran >= 30 and compressed >= 20, (ran, compressed)


def _recover_indices(K, Kc_past):
    """K [B,H,L,D] source rows, Kc_past [B,H,k,D] gathered rows -> int64 [B,H,k]; None when a head holds duplicate rows."""
    B, H, L, _ = K.shape
    k = Kc_past.shape[2]
    out = np.zeros((B, H, k), dtype=np.int64)
    Kb, Cb = bits(K), bits(Kc_past)
    for b in range(B):
        for h in range(H):
            table = {Kb[b, h, s].tobytes(): s for s in range(L)}
            if len(table) != L:
                return None
            for j in range(k):
                out[b, h, j] = table[Cb[b, h, j].tobytes()]
    return torch.from_numpy(out)


@pytest.mark.parametrize("policy", ("snapkv", "pyramidkv", "h2o"))
def test_canonical_tie_rule_selects_what_the_live_reference_selected(policy):
    """What the HIP path implements is the oracle's CANONICAL tie rule (descending score, ascending index among equals); the
    reference's CPU topk orders equal scores its own way.  On random configurations the two select the same score-value
    sequence per head (O.equivalent_selection), and the same indices wherever the selected scores are distinct."""
    rng = np.random.default_rng(777 + POLICIES.index(policy))
    checked = identical_rows = 0
    for _ in range(30):
        c = random_case(rng, policy)
        c.pop("merge", None)
        q, k, v = make_qkv(c["B"], c["H"], c["S"], 128, c["dtype"], c["kind"], c["seed"])
        kr, _, _ = run_reference(c, q, k, v)
        if kr is k:
            continue
        w = c["w"]
        ref_idx = _recover_indices(k[:, :, :-w], kr[:, :, :kr.shape[2] - w])
        if ref_idx is None or ref_idx.shape[2] == 0:
            continue
        if policy == "snapkv":
            out = O.snapkv_update_kv(k, q, v, w, c["cap"], c["ks"], c["pool"], topk_mode="canonical", return_indices=True)
        elif policy == "pyramidkv":
            out = O.pyramidkv_update_kv(k, q, v, w, c["cap"], c["ks"], c["pool"], c["layers"], c["layer"], topk_mode="canonical",
                                        return_indices=True)
        else:
            out = O.h2o_update_kv(k, q, v, w, c["cap"], topk_mode="canonical", return_indices=True)
        idx = out[2]
        s = O.h2o_scores(q, k, w) if policy == "h2o" else O.pool_scores(O.window_scores(q, k, w, "sum"), c["pool"], c["ks"])
        assert idx.shape == ref_idx.shape, c
        assert O.equivalent_selection(idx, ref_idx, s), c
        checked += 1
        for b in range(idx.shape[0]):
            for h in range(idx.shape[1]):
                row = s[b, h].float()
                sel = row[idx[b, h]]
                if sel.unique().numel() == sel.numel() and int((row == sel[-1]).sum()) == 1:
                    assert torch.equal(idx[b, h], ref_idx[b, h]), c
                    identical_rows += 1
    assert checked >= 10 and identical_rows >= 3, (checked, identical_rows)
