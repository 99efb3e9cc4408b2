"""Pin the oracle (oracle/pkv_oracle.py) against fixtures produced by the REAL reference
(tests/golden/make_golden.py ran /root/reference/pyramidkv/pyramidkv_utils.py on CPU).

Two statements are checked per fixture:
 1. restatement == reference, bit for bit, when the oracle uses the reference's own
    ``topk``/``sort`` calls (topk_mode/sort_mode = "reference");
 2. the oracle's canonical tie rule selects the same score-value sequence as the reference did
    (identical indices wherever scores are distinct), and the same number of rows per head.
"""
import json
import os

import numpy as np
import pytest
import torch

from inputs import make_qkv, bits, checksum
from oracle import pkv_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))
CASES = INDEX["cases"]


def _load(c):
    z = np.load(os.path.join(GOLD, c["name"] + ".npz"))
    q, k, v = make_qkv(c["B"], c["H"], c["S"], 128, c["dtype"], c["kind"], c["seed"])
    assert checksum(q, k, v) == int(z["in_checksum"]), "input generator drifted from fixture"
    return z, q, k, v


def _run(c, q, k, v, mode):
    pol, w, cap = c["policy"], c["w"], c["cap"]
    pool = None if c["pool"] == "none" else c["pool"]
    mg = c.get("merge")
    if pol == "snapkv":
        return O.snapkv_update_kv(k, q, v, w, cap, c["ks"], pool, topk_mode=mode, return_indices=True, merge=mg)
    if pol == "pyramidkv":
        return O.pyramidkv_update_kv(k, q, v, w, cap, c["ks"], pool, c["layers"], c["layer"],
                                     topk_mode=mode, return_indices=True, merge=mg)
    if pol == "h2o":
        return O.h2o_update_kv(k, q, v, w, cap, topk_mode=mode, return_indices=True, merge=mg)
    if pol == "streamingllm":
        return O.streamingllm_update_kv(k, q, v, w, cap, return_indices=True, merge=mg)
    raise ValueError(pol)


def _scores(c, q, k):
    pol, w = c["policy"], c["w"]
    if pol == "h2o":
        return O.h2o_scores(q, k, w)
    pool = None if c["pool"] == "none" else c["pool"]
    red = "mean" if pol in ("adakv", "headkv") else "sum"
    return O.pool_scores(O.window_scores(q, k, w, red), pool, c["ks"])


DENSE = [c for c in CASES if c["policy"] not in ("adakv", "headkv")]
FLAT = [c for c in CASES if c["policy"] in ("adakv", "headkv")]


@pytest.mark.parametrize("c", DENSE, ids=[c["name"] for c in DENSE])
def test_restatement_bit_identical_to_reference(c):
    z, q, k, v = _load(c)
    kc, vc, idx = _run(c, q, k, v, "reference")
    if bool(z["passthrough"]):
        assert kc is k and vc is v          # reference returns the input objects (:219,:315)
        return
    assert np.array_equal(bits(kc), z["kc"])
    assert np.array_equal(bits(vc), z["vc"])
    if "idx" in z.files:                    # merge fixtures (merge_kv, :119-170) carry the merged K/V only
        assert np.array_equal(idx.numpy().astype(np.int32), z["idx"])


@pytest.mark.parametrize("c", DENSE, ids=[c["name"] for c in DENSE])
def test_canonical_tie_rule_equivalent_to_reference(c):
    z, q, k, v = _load(c)
    if bool(z["passthrough"]) or c.get("merge"):
        return
    kc, vc, idx = _run(c, q, k, v, "canonical")
    ref_idx = torch.from_numpy(z["idx"].astype(np.int64))
    assert idx.shape == ref_idx.shape
    if c["policy"] == "streamingllm":
        assert torch.equal(idx, ref_idx)
        return
    s = _scores(c, q, k)
    assert O.equivalent_selection(idx, ref_idx, s)
    # where the selected scores are all distinct and clear of the threshold, indices must be identical
    for b in range(idx.shape[0]):
        for h in range(idx.shape[1]):
            row = s[b, h].float()
            sel = row[idx[b, h]]
            if sel.numel() == 0:                     # topk(0): a pyramid layer with no past tokens
                continue
            kth = sel[-1]
            distinct = sel.unique().numel() == sel.numel() and int((row == kth).sum()) == 1
            if distinct:
                assert torch.equal(idx[b, h], ref_idx[b, h])


def _run_flat(c, q, k, v, mode):
    if c["policy"] == "adakv":
        return O.adakv_update_kv(k, q, v, c["w"], c["cap"], c["ks"], c["pool"], c["floor"], c["normalize"],
                                 sort_mode=mode)
    return O.headkv_update_kv(k, q, v, c["w"], c["cap"], c["ks"], c["pool"], c["head_capacity"], c["layer"],
                              sort_mode=mode)


@pytest.mark.parametrize("c", FLAT, ids=[c["name"] for c in FLAT])
def test_flat_restatement_bit_identical_to_reference(c):
    z, q, k, v = _load(c)
    kf, vf, meta = _run_flat(c, q, k, v, "reference")
    assert np.array_equal(bits(kf), z["kc"]) and np.array_equal(bits(vf), z["vc"])
    for name in ("head_lens", "cu_klen", "cu_qlen", "cu_offset", "cu_head_offset"):
        assert np.array_equal(getattr(meta, name).numpy(), z[name]), name
    assert meta.max_seqlen_k == int(z["max_seqlen_k"]) and meta.klen_sum == int(z["klen_sum"])


@pytest.mark.parametrize("c", FLAT, ids=[c["name"] for c in FLAT])
def test_flat_canonical_equivalent_to_reference(c):
    z, q, k, v = _load(c)
    kf, vf, meta = _run_flat(c, q, k, v, "canonical")
    # per-head budgets are tie-order independent: metadata must be identical
    for name in ("head_lens", "cu_klen"):
        assert np.array_equal(getattr(meta, name).numpy(), z[name]), name
    if "idx_flat" not in z.files:
        assert np.array_equal(bits(kf), z["kc"])
        return
    s = _scores(c, q, k)
    off = 0
    for h, idx in enumerate(meta.indices):
        n = idx.numel()
        ref_idx = torch.from_numpy(z["idx_flat"][off:off + n].astype(np.int64))
        assert O.equivalent_selection(idx[None], ref_idx[None], s[0, h][None])
        off += n


def test_tie_free_fixtures_canonical_is_bit_identical_to_reference():
    """On the tie-free fixtures the oracle's canonical tie rule and the reference's CPU topk coincide exactly."""
    for c in [c for c in CASES if c.get("tie_free")]:
        z, q, k, v = _load(c)
        kc, vc, idx = _run(c, q, k, v, "canonical")
        assert np.array_equal(bits(kc), z["kc"]) and np.array_equal(bits(vc), z["vc"])
        if "idx" in z.files:
            assert np.array_equal(idx.numpy().astype(np.int32), z["idx"])


@pytest.mark.parametrize("c", [c for c in CASES if c.get("merge")], ids=[c["name"] for c in CASES if c.get("merge")])
def test_merge_explicit_spec_equals_reference(c):
    """oracle.merge_kv_explicit (the arithmetic the HIP merge kernels implement, written out element by element) against
    the REAL reference's merged K/V, starting from the reference's own selection."""
    z, q, k, v = _load(c)
    _, _, idx = _run(c, q, k, v, "reference")
    ke, ve = O.merge_kv_explicit(k, v, idx, c["w"])
    assert np.array_equal(bits(ke), z["kc"]) and np.array_equal(bits(ve), z["vc"])


def test_pyramid_budget_table():
    # cap=128,w=8,32 layers,S>=8192 -> 234,227,...,17 (sum 4016)
    ks = [O.pyramid_budget(128, 8, 32, l, 8192)[1] for l in range(32)]
    assert ks[0] == 234 and ks[1] == 227 and ks[-1] == 17 and sum(ks) == 4016
    ks = [O.pyramid_budget(64, 8, 32, l, 2048)[1] for l in range(32)]
    assert ks[0] == 110 and ks[1] == 107 and ks[-1] == 17
    ks = [O.pyramid_budget(2048, 8, 32, l, 32768)[1] for l in range(32)]
    assert ks[0] == 3978 and ks[-1] == 103
    assert O.pyramid_budget(64, 8, 32, 5, 48)[0] == "passthrough"
    assert O.pyramid_budget(64, 8, 32, 5, 100) == ("snap", 56)
    # clamp branch: max_num >= S-w
    assert O.pyramid_budget(64, 8, 32, 0, 115) == ("pyramid", 107)     # k == L: keep every past token
    assert O.pyramid_budget(64, 8, 32, 31, 115) == ("pyramid", 107 - 31 * 3)


def test_h2o_blocked_equals_materialised():
    for dt in ("bf16", "fp32"):
        q, k, v = make_qkv(1, 2, 640, 128, dt, "gauss", 5)
        a = O.h2o_scores(q, k, 8)
        b = O.h2o_scores_blocked(q, k, 8, block=128)
        assert torch.equal(a, b) if dt == "bf16" else torch.allclose(a, b, rtol=1e-6, atol=1e-9)


def test_flat_append_oracle_layout():
    H, D = 3, 128
    head_lens = torch.tensor([2, 4, 3], dtype=torch.int32)
    cu = torch.tensor([0, 2, 6, 9], dtype=torch.int32)
    cache = torch.arange(9 * D, dtype=torch.float32).view(9, D).to(torch.float16)
    state = -torch.ones(H, D, dtype=torch.float16) * torch.arange(1, H + 1)[:, None]
    out = O.update_flatten_view(cache, state, head_lens, cu)
    assert out.shape == (12, D)
    assert torch.equal(out[0:2], cache[0:2]) and torch.equal(out[2], state[0])
    assert torch.equal(out[3:7], cache[2:6]) and torch.equal(out[7], state[1])
    assert torch.equal(out[8:11], cache[6:9]) and torch.equal(out[11], state[2])
