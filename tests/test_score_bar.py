"""CPU checks of the score bar itself (tests/score_bar.py, oracle.window_score_one_product_moved): the replay reproduces the
oracle when nothing is moved, accepts a score that one moved product explains, and rejects one that nothing explains."""
import numpy as np
import pytest
import torch

import score_bar
from inputs import make_qkv
from oracle import pkv_oracle as O


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("red", ["sum", "mean"])
def test_replay_without_a_move_is_the_oracle(dt, red):
    q, k, _ = make_qkv(2, 2, 300, 128, dt, "gauss", 3)
    for w in (1, 8):
        want = O.window_scores(q, k, w, red)
        for (b, h, j) in ((0, 1, 0), (1, 0, 100), (1, 1, 300 - w - 1)):
            base, moved = O.window_score_one_product_moved(q, k, w, b, h, j, red)
            assert base == want[b, h, j] and 2 * w <= len(moved) <= 4 * w       # own products, then every row's maximum (round 6)
    want = O.h2o_scores(q, k, 8)
    base, moved = O.h2o_score_one_product_moved(q, k, 8, 1, 0, 17)
    assert base == want[1, 0, 17] and len(moved) == 16


def test_bar_accepts_one_moved_product_and_rejects_the_unexplained():
    w, pool, ks = 1, None, 1                               # un-pooled: max pooling would hide a moved score that is not a local maximum
    q, k, _ = make_qkv(1, 2, 400, 128, "bf16", "gauss", 9)
    q = (q.float() * 3).to(q.dtype)                         # logits of ~ +-8: one product moved by an ulp = several ulps of a probability
    want_u = O.window_scores(q, k, w)
    found = None
    for j in range(0, 399):
        base, moved = O.window_score_one_product_moved(q, k, w, 0, 1, j)
        far = [m for m in moved if abs(score_bar._ord_scalar(m[2]) - score_bar._ord_scalar(base)) > 1]
        if far:
            found = (j, far[0][2])
            break
    assert found is not None
    j, moved_score = found
    got_u = want_u.clone()
    got_u[0, 1, j] = moved_score
    rep = score_bar.check_window_scores(q, k, w, pool, ks, "sum", O.pool_scores(got_u, pool, ks), lambda: got_u, frac_bar=0.05)
    assert rep["beyond_1ulp"] == 1 and rep["reproduced_exactly"] == 1 and rep["max_ulp"] > 1
    bad = want_u.clone()
    bad[0, 1, j] = (want_u[0, 1, j].float() * 1.5).to(bad.dtype)             # what no moved product gives
    with pytest.raises(AssertionError, match="no single moved product"):
        score_bar.check_window_scores(q, k, w, pool, ks, "sum", O.pool_scores(bad, pool, ks), lambda: bad, frac_bar=0.05)
    # a pooled difference that the kernel's own un-pooled scores do not carry is a pooling bug, not a rounding
    pooled = O.pool_scores(want_u, pool, ks).clone()
    pooled[0, 0, 5] = (pooled[0, 0, 5].float() * 1.5).to(pooled.dtype)
    with pytest.raises(AssertionError):
        score_bar.check_window_scores(q, k, w, pool, ks, "sum", pooled, lambda: want_u, frac_bar=0.05)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_row_replay_agrees_with_the_position_replay(dt):
    """Round 6: the whole-row replay (one product of a heavy key moved, oracle.window_scores_row_with_product_moved) gives, at the
    moved key's own position, exactly what the per-position replay (window_score_one_product_moved) lists for the same move; and
    every listed heavy key is one of the row's three largest logits."""
    q, k, _ = make_qkv(1, 2, 260, 128, dt, "sink", 12)
    w = 4
    for red in ("sum", "mean"):
        for j in (0, 17, 200):
            _, moved = O.window_score_one_product_moved(q, k, w, 0, 1, j, red)
            for (r, step, val) in moved[:2 * w]:                      # the position's own products
                row = O.window_scores_row_with_product_moved(q, k, w, 0, 1, r, j, step, red)
                assert row[j] == val, (red, j, r, step)
    heavy = O.window_heavy_keys(q, k, w, 0, 1, 3)
    assert len(heavy) == 3 * w and all(0 <= r < w and 0 <= j < 260 for r, j in heavy)
    want = O.window_scores(q, k, w)
    base = O.window_scores_row_with_product_moved(q, k, w, 0, 1, 0, 5, 1)
    assert base.shape == want[0, 1].shape


def test_merge_pivots_are_the_pivots_of_merge_kv():
    """oracle.merge_pivots returns what merge_kv's :150-151 computed: feeding them to the explicit arithmetic of tests/merge_bar.py
    reproduces merge_kv's K and V bit for bit (both dtypes, a case with duplicate keys)."""
    import merge_bar
    for dt, kind in (("bf16", "gauss"), ("fp16", "planted")):
        q, k, v = make_qkv(1, 2, 180, 128, dt, kind, 31)
        w, kk = 4, 20
        s = O.pool_scores(O.window_scores(q, k, w), "maxpool", 7)
        idx = O.topk_canonical(s, kk)
        km, vm = O.merge_kv(k, v, idx, w, "pivot")
        piv = O.merge_pivots(k, v, idx, w)
        union = set(idx.flatten().tolist())
        drop = [p for p in range(180) if p not in union]
        assert piv.shape == (1, 2, len(drop))
        for h in range(2):
            k2, v2 = merge_bar.merge_with_pivots(k[0, h], v[0, h], idx[0, h], drop, piv[0, h].tolist(), w)
            assert torch.equal(k2, km[0, h]) and torch.equal(v2, vm[0, h]), (dt, h)
