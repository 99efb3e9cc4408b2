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
