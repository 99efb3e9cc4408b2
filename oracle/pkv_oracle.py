"""CPU oracle for the prefill-time KV-cache eviction path.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, op for op, of the reference's
``pyramidkv/pyramidkv_utils.py`` ``*KVCluster.update_kv`` bodies (reference
checkout: Zefan-Cai/PyramidKV @ 2024-12-20).  The arithmetic of the reference
lives in PyTorch ATen ops, so the restatement calls the same ATen ops in the
same order with the same dtypes; run on CPU tensors it is bit-identical to the
reference run on CPU (pinned by ``tests/golden/*.npz``, generated from the real
reference by ``tests/golden/make_golden.py``; see ``tests/test_oracle_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``pyramidkv_amd``) never does.

Parity status: the reference ships no tests/golden vectors of its own
(SURVEY.md section 4), so the pin is "reference executed here" -> committed fixtures.

The one place the oracle *adds* a definition is top-k tie order: the reference's
``tensor.topk`` leaves the order of equal scores backend-defined (CPU:
libstdc++ partial_sort, arbitrary).  ``topk_canonical`` fixes it to
(value descending, index ascending), i.e. a stable descending sort, which is
what the reference produces when it runs on a GPU (ATen radix-select + stable
block radix sort).  ``tests`` check that the canonical choice and the reference's
own choice select identical score-value sequences.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

__all__ = [
    "pyramid_budget", "window_logits", "window_scores", "h2o_scores", "h2o_scores_blocked",
    "pool_scores", "topk_canonical", "topk_reference", "gather_compact", "merge_kv", "merge_kv_explicit",
    "snapkv_update_kv", "pyramidkv_update_kv", "h2o_update_kv", "streamingllm_update_kv",
    "adakv_update_kv", "headkv_update_kv", "AdaMeta", "equivalent_selection", "window_score_one_product_moved", "h2o_score_one_product_moved",
]


# --------------------------------------------------------------------------- budgets
def pyramid_budget(max_capacity_prompt: int, window_size: int, num_hidden_layers: int,
                   layer_idx: int, q_len: int, beta: int = 20) -> Tuple[str, int]:
    """Per-layer pyramidal budget, integer arithmetic of pyramidkv_utils.py:205-215 and the
    three branches at :218, :220, :252.  Returns (branch, k) with branch in
    {"passthrough", "snap", "pyramid"}; k = number of *past* tokens kept (window excluded)."""
    min_num = (max_capacity_prompt - window_size) // beta                      # :205
    max_num = (max_capacity_prompt - window_size) * 2 - min_num                # :206
    if max_num >= q_len - window_size:                                         # :209
        max_num = q_len - window_size                                          # :210
        min_num = (max_capacity_prompt - window_size) * 2 - max_num            # :211
    steps = (max_num - min_num) // (num_hidden_layers - 1)                     # :214
    k_layer = max_num - layer_idx * steps                                      # :215
    if q_len < max_capacity_prompt:                                            # :218
        return "passthrough", 0
    if q_len < (max_capacity_prompt - window_size) * 2:                        # :220
        return "snap", max_capacity_prompt - window_size                       # :238
    return "pyramid", k_layer                                                  # :270


# --------------------------------------------------------------------------- scores
def _corner_mask(w: int, dtype: torch.dtype, device) -> torch.Tensor:
    """pyramidkv_utils.py:318-322: fp32 [w,w] tensor, 0 where col <= row, finfo(dtype).min above."""
    mask = torch.full((w, w), torch.finfo(dtype).min, device=device)
    mask_cond = torch.arange(mask.size(-1), device=device)
    mask.masked_fill_(mask_cond < (mask_cond + 1).view(mask.size(-1), 1), 0)
    return mask


def _scale(attn: torch.Tensor, head_dim: int, scale_mode: str) -> torch.Tensor:
    if scale_mode == "div":          # what ATen does on CPU: fp32(a) / fp32(sqrt(D)), one rounding
        return attn / math.sqrt(head_dim)
    if scale_mode == "rcp":          # what ATen does on a GPU for a CPU-scalar divisor:
        inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(math.sqrt(head_dim), dtype=torch.float32)
        return (attn.float() * inv.to(attn.device)).to(attn.dtype)
    raise ValueError(scale_mode)


def window_logits(query_states: torch.Tensor, key_states: torch.Tensor, window_size: int,
                  scale_mode: str = "div") -> torch.Tensor:
    """pyramidkv_utils.py:317-324: A = (Q[-w:] @ K^T)/sqrt(D) in model dtype, then the causal
    corner of the last w x w block gets finfo.min added (fp32 mask, result rounded to model dtype)."""
    head_dim = query_states.shape[-1]
    w = window_size
    attn = torch.matmul(query_states[..., -w:, :], key_states.transpose(2, 3))
    attn = _scale(attn, head_dim, scale_mode)
    mask = _corner_mask(w, attn.dtype, attn.device)
    attn[:, :, -w:, -w:] += mask[None, None, :, :]
    return attn


def window_scores(query_states, key_states, window_size: int, reduce: str = "sum",
                  scale_mode: str = "div") -> torch.Tensor:
    """pyramidkv_utils.py:317-327 (reduce='sum') / :649-661 (reduce='mean', AdaKV/HeadKV):
    fp32 softmax over all S keys, rounded to model dtype, then the w window rows reduced over
    columns [0, S-w).  Returns [B,H,S-w] in model dtype (un-pooled)."""
    w = window_size
    attn = window_logits(query_states, key_states, w, scale_mode)
    attn = F.softmax(attn, dim=-1, dtype=torch.float32).to(query_states.dtype)
    sl = attn[:, :, -w:, :-w]
    return sl.sum(dim=-2) if reduce == "sum" else sl.mean(dim=-2)


def window_scores_row_with_product_moved(query_states, key_states, window_size: int, b: int, h: int, r: int, j: int, step: int,
                                         reduce: str = "sum", scale_mode: str = "div"):
    """``window_scores(...)[b, h, :]`` when the ONE product q_r . k_j of :317 (window row r, key j) is rounded to its neighbour
    (`step` = -1 / +1) - the WHOLE score row of the head, not only position j: where a single key carries a noticeable share of a
    row's softmax mass (attention-sink-like rows: p_j ~ 0.1), moving its logit by one step moves the row's normaliser Z by
    ~p_j x 1.5 %, i.e. every other probability of that row by about one unit of the last place (round 6: fp16 sink inputs,
    S = 8192 - one product 2.6e-6 below a rounding midpoint, 265 of 8184 scores of the head off by 1-2 units and the key itself by
    19, all reproduced exactly by this replay; tools/probes/sink_row_probe.py)."""
    w, T, head_dim = window_size, query_states.dtype, query_states.shape[-1]
    qw = query_states[b, h, -w:, :]
    kk = key_states[b, h]
    S = kk.shape[0]
    P0 = torch.matmul(qw[None, None], kk.transpose(0, 1)[None, None])[0, 0].clone()     # [w, S] model dtype (:317)
    P0[r, j] = _neighbour(P0[r, j], step)
    A = _scale(P0[None, None], head_dim, scale_mode)[0, 0].clone()
    A[-w:, -w:] += _corner_mask(w, T, P0.device)                                        # :322-324
    col = F.softmax(A, dim=-1, dtype=torch.float32).to(T)[:, :S - w]                    # :326
    return col.sum(dim=-2) if reduce == "sum" else col.mean(dim=-2)                     # :327 / :661


def window_heavy_keys(query_states, key_states, window_size: int, b: int, h: int, n: int = 3, scale_mode: str = "div"):
    """[(window row r, key j), ...]: the n largest logits of every window row of head (b, h) - the keys whose products can drag a
    row's normaliser along when they move (see window_scores_row_with_product_moved)."""
    w, T, head_dim = window_size, query_states.dtype, query_states.shape[-1]
    P0 = torch.matmul(query_states[b, h, -w:, :][None, None], key_states[b, h].transpose(0, 1)[None, None])[0, 0]
    A = _scale(P0[None, None], head_dim, scale_mode)[0, 0].clone()
    A[-w:, -w:] += _corner_mask(w, T, P0.device)
    top = A.float().topk(min(n, A.shape[-1]), dim=-1).indices
    return [(r, int(j)) for r in range(w) for j in top[r].tolist()]


def _neighbour(x: torch.Tensor, step: int) -> torch.Tensor:
    """The model-dtype value next above (step > 0) / below (step < 0) the finite 16-bit scalar tensor x."""
    b = int(x.reshape(1).view(torch.int16).item()) & 0xffff
    if b & 0x7fff == 0:                                   # +-0: the smallest value of the wanted sign
        nb = 0x0001 if step > 0 else 0x8001
    elif b < 0x8000:
        nb = b + 1 if step > 0 else b - 1
    else:
        nb = b - 1 if step > 0 else b + 1
    return torch.tensor([nb if nb < 0x8000 else nb - 0x10000], dtype=torch.int16).view(x.dtype)[0]


def window_score_one_product_moved(query_states, key_states, window_size: int, b: int, h: int, j: int,
                                   reduce: str = "sum", scale_mode: str = "div"):
    """What ``window_scores`` (pyramidkv_utils.py:317-327 / :649-661) gives at position (b, h, j) when ONE of the w products
    q_r . k_j of :317 - a model-dtype value, i.e. an fp32 accumulation rounded once - is rounded to its NEIGHBOUR instead:
    the only freedom two correct implementations of :317 have (the accumulation order of the 128 terms; ATen's CPU kernel
    and an MFMA disagree on it for products that sit at a rounding midpoint).  Returns (score as computed here without any
    move, [(window_row, step, score), ...] for the 2w single moves of the position's own products, followed by the 2w single
    moves of each row's MAXIMUM - round 6).  16-bit tensors only."""
    w = window_size
    T = query_states.dtype
    head_dim = query_states.shape[-1]
    qw = query_states[b, h, -w:, :]
    kk = key_states[b, h]
    S = kk.shape[0]
    P0 = torch.matmul(qw[None, None], kk.transpose(0, 1)[None, None])[0, 0]            # [w, S] model dtype (:317)
    mask = _corner_mask(w, T, P0.device)

    def column(prod):                                           # probabilities of column j for all w rows (model dtype)
        A = _scale(prod[None, None], head_dim, scale_mode)[0, 0].clone()
        A[-w:, -w:] += mask                                     # :322-324
        return F.softmax(A, dim=-1, dtype=torch.float32).to(T)[:, j:j + 1]             # :326

    def reduced(col):
        return (col.sum(dim=-2) if reduce == "sum" else col.mean(dim=-2))[0]          # :327 / :661

    base_col = column(P0)
    moved = []
    for r in range(w):
        for step in (-1, 1):
            P1 = P0[r:r + 1].clone()
            P1[0, j] = _neighbour(P0[r, j], step)
            A = _scale(P1[None, None], head_dim, scale_mode)[0, 0].clone()
            if S - w <= j:
                raise ValueError("position inside the observation window")
            row_mask = torch.zeros(S, dtype=torch.float32)
            row_mask[-w:] = mask[r]
            A[0] += row_mask
            pr = F.softmax(A, dim=-1, dtype=torch.float32).to(T)[0, j]
            col = base_col.clone()
            col[r, 0] = pr
            moved.append((r, step, reduced(col)))
    # Round 6: the same single move applied to the row's MAXIMUM instead of the position's own product.  On attention-sink rows
    # the maximum is one large logit (spacing 2^-5 in fp16 around 40): moving it moves the row's whole normalisation, i.e. every
    # position of that row by the same few per cent (tools/probes/sink_score_diff.py).
    A0 = _scale(P0[None, None], head_dim, scale_mode)[0, 0].clone()
    A0[-w:, -w:] += mask
    for r in range(w):
        jm = int(A0[r].float().argmax())
        if jm == j:
            continue                                            # already among the moves above
        for step in (-1, 1):
            P1 = P0[r:r + 1].clone()
            P1[0, jm] = _neighbour(P0[r, jm], step)
            A = _scale(P1[None, None], head_dim, scale_mode)[0, 0].clone()
            row_mask = torch.zeros(S, dtype=torch.float32)
            row_mask[-w:] = mask[r]
            A[0] += row_mask
            pr = F.softmax(A, dim=-1, dtype=torch.float32).to(T)[0, j]
            col = base_col.clone()
            col[r, 0] = pr
            moved.append((r, step, reduced(col)))
    return reduced(base_col), moved


def h2o_scores(query_states, key_states, window_size: int, scale_mode: str = "div") -> torch.Tensor:
    """pyramidkv_utils.py:544-554: ALL S query rows, full SxS (non-causal except the last w x w
    corner), fp32 softmax, column sum over all S rows of columns [0,S-w).  Materialises SxS:
    only for small S.  Returns [B,H,S-w] model dtype."""
    head_dim = query_states.shape[-1]
    w = window_size
    attn = torch.matmul(query_states, key_states.transpose(2, 3))
    attn = _scale(attn, head_dim, scale_mode)
    mask = _corner_mask(w, attn.dtype, attn.device)
    attn[:, :, -w:, -w:] += mask[None, None, :, :]
    attn = F.softmax(attn, dim=-1, dtype=torch.float32).to(query_states.dtype)
    return attn[:, :, :, :-w].sum(dim=-2)


def h2o_score_one_product_moved(query_states, key_states, window_size: int, b: int, h: int, j: int, top_rows: int = 8,
                                scale_mode: str = "div"):
    """``h2o_scores`` (pyramidkv_utils.py:544-554) at position (b, h, j) with ONE product q_i . k_j of :544 rounded to its
    neighbour, for the ``top_rows`` query rows that contribute most to column j (a move in a row whose probability is far
    below the column sum's last place cannot show).  Returns (score without a move, [(row, step, score), ...])."""
    w = window_size
    T = query_states.dtype
    head_dim = query_states.shape[-1]
    qh, kh = query_states[b, h], key_states[b, h]
    S = kh.shape[0]
    P0 = torch.matmul(qh[None, None], kh.transpose(0, 1)[None, None])[0, 0]             # [S, S] model dtype (:544)
    mask = _corner_mask(w, T, P0.device)

    def probs(prod, rows=None):
        A = _scale(prod[None, None], head_dim, scale_mode)[0, 0].clone()
        if rows is None:
            A[-w:, -w:] += mask
        else:
            for t, r in enumerate(rows):
                if r >= S - w:
                    A[t, -w:] += mask[r - (S - w)]
        return F.softmax(A, dim=-1, dtype=torch.float32).to(T)
    col = probs(P0)[:, j].clone()                                                     # [S] model dtype
    base = col[:, None].sum(dim=-2)[0]
    order = torch.argsort(col.float(), descending=True)[:top_rows].tolist()
    moved = []
    for r in order:
        for step in (-1, 1):
            P1 = P0[r:r + 1].clone()
            P1[0, j] = _neighbour(P0[r, j], step)
            c2 = col.clone()
            c2[r] = probs(P1, rows=[r])[0, j]
            moved.append((r, step, c2[:, None].sum(dim=-2)[0]))
    return base, moved


def h2o_scores_blocked(query_states, key_states, window_size: int, block: int = 128,
                       scale_mode: str = "div") -> torch.Tensor:
    """Row-blocked restatement of h2o_scores for sizes where SxS cannot be materialised
    (S=32768 -> 68.7 GB bf16).  Same per-row arithmetic; the column sums are
    accumulated in fp32 across row blocks and rounded once at the end.  Equal to ``h2o_scores``
    whenever the fp32 column sums are order-independent (checked in tests at S<=1024)."""
    B, H, S, D = query_states.shape
    w = window_size
    L = S - w
    acc = torch.zeros(B, H, L, dtype=torch.float32, device=query_states.device)
    kt = key_states.transpose(2, 3)
    mask = _corner_mask(w, query_states.dtype, query_states.device)
    for r0 in range(0, S, block):
        r1 = min(S, r0 + block)
        attn = torch.matmul(query_states[:, :, r0:r1, :], kt)
        attn = _scale(attn, D, scale_mode)
        if r1 > L:                                   # rows of the observation window in this block
            lo = max(r0, L)
            attn[:, :, lo - r0:, -w:] += mask[None, None, lo - L:r1 - L, :]
        attn = F.softmax(attn, dim=-1, dtype=torch.float32).to(query_states.dtype)
        acc += attn[:, :, :, :L].float().sum(dim=-2)
    return acc.to(query_states.dtype)


def pool_scores(scores: torch.Tensor, pooling: Optional[str], kernel_size: int) -> torch.Tensor:
    """pyramidkv_utils.py:328-333.  pooling None = H2O (:561, no pooling)."""
    if pooling is None:
        return scores
    if pooling == "avgpool":
        return F.avg_pool1d(scores, kernel_size=kernel_size, padding=kernel_size // 2, stride=1)
    if pooling == "maxpool":
        return F.max_pool1d(scores, kernel_size=kernel_size, padding=kernel_size // 2, stride=1)
    raise ValueError("Pooling method not supported")


# --------------------------------------------------------------------------- top-k
def topk_reference(scores: torch.Tensor, k: int) -> torch.Tensor:
    """Exactly the reference call, pyramidkv_utils.py:334: backend-defined tie order."""
    return scores.topk(k, dim=-1).indices


def topk_canonical(scores: torch.Tensor, k: int) -> torch.Tensor:
    """Top-k with the tie rule pinned: (value desc, index asc) == stable descending sort.
    Same *value sequence* as topk_reference on any backend; identical indices where values
    are distinct.  NaN sorts as largest (torch semantics)."""
    order = torch.sort(scores, dim=-1, descending=True, stable=True).indices
    return order[..., :k].contiguous()


def equivalent_selection(idx_a: torch.Tensor, idx_b: torch.Tensor, scores: torch.Tensor) -> bool:
    """True iff two index tensors [..,k] select the same score-value sequence and each is a valid
    selection (distinct, in range).  This is the strongest statement that holds between two
    backends' ``topk`` when ties are present."""
    va = torch.gather(scores, -1, idx_a)
    vb = torch.gather(scores, -1, idx_b)
    if not torch.equal(va, vb):
        return False
    for idx in (idx_a, idx_b):
        s = torch.sort(idx, dim=-1).values
        if idx.shape[-1] > 1 and bool((s[..., 1:] == s[..., :-1]).any()):
            return False
        if bool((idx < 0).any()) or bool((idx >= scores.shape[-1]).any()):
            return False
    return True


# --------------------------------------------------------------------------- gather
def gather_compact(key_states, value_states, indices: torch.Tensor, window_size: int):
    """pyramidkv_utils.py:335,341-346: rows in top-k order, then the w window rows in position order."""
    head_dim = key_states.shape[-1]
    w = window_size
    idx = indices.to(torch.int64).unsqueeze(-1).expand(-1, -1, -1, head_dim)
    k_past = key_states[:, :, :-w, :].gather(dim=2, index=idx)
    v_past = value_states[:, :, :-w, :].gather(dim=2, index=idx)
    return (torch.cat([k_past, key_states[:, :, -w:, :]], dim=2),
            torch.cat([v_past, value_states[:, :, -w:, :]], dim=2))


# --------------------------------------------------------------------------- merge (LOOK-M pivot merge)
def merge_kv(key_states, value_states, indices: torch.Tensor, window_size: int, merge: str):
    """pyramidkv_utils.py:119-170, op for op (``indices`` int64 [B,H,k], un-expanded).  The quirks are the spec:
      * a position is "dropped" iff NO (batch, head) selected it (``isin`` over the flattened indices of all heads, :128-133),
        and the window positions count as dropped too (``arange(k_len)``, :128) - they merge onto themselves;
      * keys come back ordered [window, selected] (:146) but values [selected, window] (:148), and the value merge uses the
        KEY order's target numbers (:162) on the VALUE order's rows;
      * every dropped key is averaged with its most similar kept key, (k_drop + k_pivot)/2 (:160), and the results are
        scatter-reduced (mean, include_self) onto the pivots (:161)."""
    if merge != "pivot":
        raise ValueError('Merge method not supported')                                            # :164
    head_dim = key_states.shape[-1]
    drop_keys, drop_values, k_hh_recent, v_hh_recent, max_indices = _merge_front(key_states, value_states, indices, window_size)
    merged_indices = max_indices.unsqueeze(-1).repeat(1, 1, 1, head_dim)                          # :156
    k_hh_selected = torch.gather(input=k_hh_recent, dim=2, index=merged_indices)
    k_hh_merged = (drop_keys + k_hh_selected) / 2                                                 # :158
    k_out = torch.scatter_reduce(input=k_hh_recent, dim=2, index=merged_indices, src=k_hh_merged, reduce='mean',
                                 include_self=True)                                               # :159
    v_hh_selected = torch.gather(input=v_hh_recent, dim=2, index=merged_indices)
    v_hh_merged = (drop_values + v_hh_selected) / 2
    v_out = torch.scatter_reduce(input=v_hh_recent, dim=2, index=merged_indices, src=v_hh_merged, reduce='mean',
                                 include_self=True)                                               # :162
    return k_out, v_out


def merge_pivots(key_states, value_states, indices: torch.Tensor, window_size: int):
    """The pivots of ``merge_kv`` - ``similarity.max(dim=-1)`` of :150-151 by the very same ops: int64 [B, H, dropped rows], the
    kept-row number (KEY order [window, selected]) every dropped row merges into.  tests/merge_bar.py compares the kernel's
    pivots with THESE (an fp32 replica of the fp16 / bf16 ``@`` accumulates in another order and is not the reference)."""
    return _merge_front(key_states, value_states, indices, window_size)[4]


def _merge_front(key_states, value_states, indices, window_size):
    """:121-151 of merge_kv: the dropped rows, the kept rows in the reference's two orders, and the pivots."""
    bsz, num_heads, k_len, head_dim = key_states.shape
    idx = indices.to(torch.int64).unsqueeze(-1).expand(-1, -1, -1, head_dim)
    selected_keys = key_states.gather(dim=2, index=idx)                                           # :125
    selected_values = value_states.gather(dim=2, index=idx)
    all_indices = torch.arange(k_len, device=key_states.device).unsqueeze(0).unsqueeze(0).expand(bsz, num_heads, k_len)
    all_flat = all_indices.flatten()
    is_selected = torch.isin(all_flat, idx.flatten())                                             # :131
    drop_flat = all_flat[~is_selected]
    drop_len = drop_flat.shape[0] // (bsz * num_heads)
    drop_indices = drop_flat.reshape(bsz, num_heads, drop_len).unsqueeze(-1).expand(-1, -1, -1, head_dim)
    drop_keys = key_states.gather(dim=2, index=drop_indices)                                      # :137
    drop_values = value_states.gather(dim=2, index=drop_indices)
    recent_keys = key_states[:, :, -window_size:, :]
    k_hh_recent = torch.cat([recent_keys, selected_keys], dim=2)                                  # :146
    v_hh_recent = torch.cat([selected_values, value_states[:, :, -window_size:, :]], dim=2)       # :148
    similarity = (drop_keys / torch.norm(drop_keys, dim=-1).unsqueeze(-1).repeat(1, 1, 1, head_dim)) @ \
        ((k_hh_recent / (torch.norm(k_hh_recent, dim=-1).unsqueeze(-1).repeat(1, 1, 1, head_dim))).transpose(-1, -2))   # :150
    _, max_indices = similarity.max(dim=-1)                                                       # :151
    return drop_keys, drop_values, k_hh_recent, v_hh_recent, max_indices


def merge_kv_explicit(key_states, value_states, indices: torch.Tensor, window_size: int):
    """The arithmetic of ``merge_kv`` spelled out element by element - the specification the HIP kernels implement
    (checked against the ATen form above in tests/test_oracle_golden.py; small cases only, Python loops):
      norm      n = dtype(sqrt(sum x^2))                       (torch.norm, fp32 accumulate, one rounding)
      cosine    sim = dtype(dot(dtype(x / n_x), dtype(t / n_t)))  (fp32 accumulate), pivot = FIRST maximum
      merged    m = dtype(dtype(x + t_pivot) / 2)
      scatter   out_j = dtype( dtype(t_j + sum_i m_i  [fp32, ascending i])  /  dtype(1 + n_j) )
    where dtype(.) is one round-to-nearest-even to the model dtype.  Note dtype(1 + n_j): above 256 (bf16) / 2048 (fp16)
    merged rows the divisor itself is rounded."""
    B, H, S, D = key_states.shape
    k, w, tdt = indices.shape[-1], window_size, key_states.dtype

    def rnd(x):
        return x.to(tdt).float()

    union = set(indices.flatten().tolist())
    drop = [p for p in range(S) if p not in union]
    ko = torch.empty(B, H, k + w, D, dtype=tdt)
    vo = torch.empty(B, H, k + w, D, dtype=tdt)
    for b in range(B):
        for h in range(H):
            Kf, Vf = key_states[b, h].float(), value_states[b, h].float()
            sel = indices[b, h].to(torch.int64)
            tgt_k = torch.cat([Kf[S - w:], Kf[sel]], 0)             # [window, selected]
            tgt_v = torch.cat([Vf[sel], Vf[S - w:]], 0)             # [selected, window]

            def unit(X):
                n = rnd(torch.sqrt((X * X).sum(-1)))
                return rnd(X / n[:, None])

            sim = rnd(unit(Kf[drop]) @ unit(tgt_k).T)
            pivot = (sim == sim.max(-1, keepdim=True).values).float().argmax(-1)   # first maximum
            acc_k, acc_v, cnt = tgt_k.clone(), tgt_v.clone(), torch.ones(w + k)
            for i, p in enumerate(drop):
                j = int(pivot[i])
                acc_k[j] = acc_k[j] + rnd(rnd(Kf[p] + tgt_k[j]) / 2)
                acc_v[j] = acc_v[j] + rnd(rnd(Vf[p] + tgt_v[j]) / 2)
                cnt[j] += 1
            ko[b, h] = (rnd(acc_k) / rnd(cnt)[:, None]).to(tdt)
            vo[b, h] = (rnd(acc_v) / rnd(cnt)[:, None]).to(tdt)
    return ko, vo


# --------------------------------------------------------------------------- policies
def _select(scores, k, topk_mode):
    return topk_canonical(scores, k) if topk_mode == "canonical" else topk_reference(scores, k)


def _finish(key_states, value_states, idx, window_size, merge, return_indices):
    """:336-346 (and the identical tails of the other policies): merge_kv when ``merge`` is set, else gather + window tail."""
    if merge is not None:
        kc, vc = merge_kv(key_states, value_states, idx, window_size, merge)                       # :337-339
    else:
        kc, vc = gather_compact(key_states, value_states, idx, window_size)
    return (kc, vc, idx) if return_indices else (kc, vc)


def snapkv_update_kv(key_states, query_states, value_states, window_size, max_capacity_prompt,
                     kernel_size, pooling, topk_mode="canonical", scale_mode="div",
                     return_indices=False, merge=None):
    """SnapKVCluster.update_kv, pyramidkv_utils.py:306-347."""
    assert key_states.shape[-2] == query_states.shape[-2]
    q_len = query_states.shape[-2]
    if q_len < max_capacity_prompt:                                           # :314
        return (key_states, value_states, None) if return_indices else (key_states, value_states)
    s = window_scores(query_states, key_states, window_size, "sum", scale_mode)
    s = pool_scores(s, pooling, kernel_size)
    idx = _select(s, max_capacity_prompt - window_size, topk_mode)            # :334
    return _finish(key_states, value_states, idx, window_size, merge, return_indices)


def pyramidkv_update_kv(key_states, query_states, value_states, window_size, max_capacity_prompt,
                        kernel_size, pooling, num_hidden_layers, layer_idx, beta=20,
                        topk_mode="canonical", scale_mode="div", return_indices=False, merge=None):
    """PyramidKVCluster.update_kv, pyramidkv_utils.py:197-283."""
    assert key_states.shape[-2] == query_states.shape[-2]
    q_len = query_states.shape[-2]
    branch, k = pyramid_budget(max_capacity_prompt, window_size, num_hidden_layers, layer_idx, q_len, beta)
    if branch == "passthrough":
        return (key_states, value_states, None) if return_indices else (key_states, value_states)
    s = window_scores(query_states, key_states, window_size, "sum", scale_mode)
    s = pool_scores(s, pooling, kernel_size)
    idx = _select(s, k, topk_mode)
    return _finish(key_states, value_states, idx, window_size, merge, return_indices)


def h2o_update_kv(key_states, query_states, value_states, window_size, max_capacity_prompt,
                  topk_mode="canonical", scale_mode="div", blocked=False, return_indices=False, merge=None):
    """H2OKVCluster.update_kv, pyramidkv_utils.py:533-575."""
    assert key_states.shape[-2] == query_states.shape[-2]
    q_len = query_states.shape[-2]
    if q_len < max_capacity_prompt:                                           # :541
        return (key_states, value_states, None) if return_indices else (key_states, value_states)
    fn = h2o_scores_blocked if blocked else h2o_scores
    s = fn(query_states, key_states, window_size, scale_mode=scale_mode)
    idx = _select(s, max_capacity_prompt - window_size, topk_mode)            # :562
    return _finish(key_states, value_states, idx, window_size, merge, return_indices)


def streamingllm_update_kv(key_states, query_states, value_states, window_size, max_capacity_prompt,
                           return_indices=False, merge=None):
    """StreamingLLMKVCluster.update_kv, pyramidkv_utils.py:595-620: sinks 0..cap-w-1 + last w."""
    assert key_states.shape[-2] == query_states.shape[-2]
    bsz, num_heads, q_len, head_dim = query_states.shape
    if q_len < max_capacity_prompt:                                           # :603
        return (key_states, value_states, None) if return_indices else (key_states, value_states)
    idx = torch.arange(max_capacity_prompt - window_size, dtype=torch.int64, device=key_states.device)
    idx = idx[None, None, :].repeat(bsz, num_heads, 1)                        # :607-608
    return _finish(key_states, value_states, idx, window_size, merge, return_indices)


@dataclass
class AdaMeta:
    """Var-len metadata left on the cluster object, pyramidkv_utils.py:682-698."""
    head_lens: torch.Tensor      # int32 [H]
    cu_klen: torch.Tensor        # int32 [H+1]
    cu_qlen: torch.Tensor        # int32 [H+1]
    cu_offset: torch.Tensor      # int32 [H+1]
    cu_head_offset: torch.Tensor  # int32 [H]
    max_seqlen_k: int
    klen_sum: int
    head_capacity: Optional[List[int]] = None   # cap_h (window excluded), for tests
    indices: Optional[List[torch.Tensor]] = None  # per-head int64 [cap_h]


def _ada_meta(num_heads, k_lens, klen_sum, max_seqlen_k, device) -> AdaMeta:
    head_lens = torch.tensor(k_lens, dtype=torch.int32, device=device)
    cu_headlens = torch.cumsum(head_lens, dim=0, dtype=torch.int32)
    cu_klen = cu_headlens - head_lens
    cu_klen = torch.cat([cu_klen, torch.tensor([klen_sum], dtype=torch.int32, device=device)], dim=0)
    layer_qlens = torch.ones(num_heads, dtype=torch.int32, device=device)
    cu_qlen = torch.cumsum(layer_qlens, dim=0, dtype=torch.int32) - layer_qlens
    cu_qlen = torch.cat([cu_qlen, torch.tensor([num_heads], dtype=torch.int32, device=device)], dim=0)
    return AdaMeta(head_lens, cu_klen, cu_qlen,
                   torch.arange(0, num_heads + 1, dtype=torch.int32, device=device),
                   torch.arange(1, num_heads + 1, dtype=torch.int32, device=device),
                   max_seqlen_k, klen_sum)


def _flat_gather(key_states, value_states, per_head_idx, window_size):
    """pyramidkv_utils.py:733-757: per head gather + window tail, concatenated flat [sum, D]."""
    D = key_states.shape[-1]
    w = window_size
    ks, vs, lens = [], [], []
    for h, ci in enumerate(per_head_idx):
        gi = ci.to(torch.int64).view(1, 1, -1, 1).expand(-1, -1, -1, D)
        kh, vh = key_states[:, h:h + 1], value_states[:, h:h + 1]
        ks.append(torch.cat([kh.gather(2, gi), kh[:, :, -w:, :]], dim=2).view(-1, D))
        vs.append(torch.cat([vh.gather(2, gi), vh[:, :, -w:, :]], dim=2).view(-1, D))
        lens.append(ci.shape[-1] + w)
    return torch.cat(ks, 0), torch.cat(vs, 0), lens


def adakv_head_capacity(attn_score: torch.Tensor, base_capacity: int, floor_ratio: float,
                        normalize: bool, sort_mode: str = "canonical"):
    """pyramidkv_utils.py:706-719: per-head sorted order + global top-(H*base) budget split.
    Returns (sorted_indices [B,H,L] int64, head_capacity int32 [B,H])."""
    bsz, num_heads, length = attn_score.shape
    floor_capacity = int(base_capacity * floor_ratio)                          # :632
    stable = sort_mode == "canonical"
    sorted_attn_score, sorted_idx = attn_score.sort(dim=-1, descending=True, stable=stable)  # :706
    adaptive = sorted_attn_score
    if normalize:                                                              # :709-711
        ratio_weight = sorted_attn_score[..., :base_capacity].sum(dim=-1, keepdim=True) / \
            sorted_attn_score.sum(dim=-1, keepdim=True)
        adaptive = adaptive * ratio_weight
    adaptive = adaptive.reshape(bsz, length * num_heads)                       # :712
    if stable:
        top = torch.sort(adaptive, dim=-1, descending=True, stable=True).indices[..., :num_heads * base_capacity]
    else:
        top = torch.topk(adaptive, k=num_heads * base_capacity, dim=-1).indices  # :713
    top = top // length                                                        # :714
    cap = torch.zeros((bsz, num_heads), device=attn_score.device, dtype=top.dtype)
    cap.scatter_add_(-1, top, torch.ones_like(top, dtype=cap.dtype))           # :716-717
    assert cap.sum().item() == num_heads * base_capacity * bsz                 # :718 (bsz==1 there)
    cap = torch.round(cap * (1 - floor_ratio) + floor_capacity).int()          # :719
    return sorted_idx, cap


def adakv_update_kv(key_states, query_states, value_states, window_size, max_capacity_prompt,
                    kernel_size, pooling, floor, normalize, sort_mode="canonical", scale_mode="div"):
    """AdaKVCluster.update_kv, pyramidkv_utils.py:674-757.  Returns (K_flat, V_flat, AdaMeta)."""
    bsz, num_heads, q_len, head_dim = query_states.shape
    base_capacity = max_capacity_prompt - window_size                          # :630
    s = window_scores(query_states, key_states, window_size, "mean", scale_mode)
    s = pool_scores(s, pooling, kernel_size)
    if base_capacity > s.size(-1):                                             # :700-703
        meta = _ada_meta(num_heads, [q_len] * num_heads, q_len * num_heads, q_len, key_states.device)
        return key_states.reshape(-1, head_dim), value_states.reshape(-1, head_dim), meta
    sorted_idx, cap = adakv_head_capacity(s, base_capacity, floor, normalize, sort_mode)
    assert bsz == 1                                                            # :724
    per_head = [sorted_idx[0, h, :int(cap[0, h])] for h in range(num_heads)]   # :734
    kf, vf, lens = _flat_gather(key_states, value_states, per_head, window_size)
    meta = _ada_meta(num_heads, lens, sum(lens), max(lens), key_states.device)
    meta.head_capacity = [int(c) for c in cap[0]]
    meta.indices = per_head
    return kf, vf, meta


def headkv_runner_capacity(head_list, num_hidden_layers, num_attention_heads, max_capacity_prompts, head_beta=1.01):
    """run_longbench.py:227-233, statement for statement (``head_list`` = the parsed score JSON)."""
    import numpy as np
    head_score_list = [np.mean(l[1]) for l in head_list.items()]
    head_score_list = torch.tensor(head_score_list / sum(head_score_list))
    total_attention = head_score_list.reshape(num_hidden_layers, num_attention_heads)
    total_pool_capacity = (max_capacity_prompts // head_beta) * num_hidden_layers * num_attention_heads
    min_num = (max_capacity_prompts - max_capacity_prompts // head_beta)
    return torch.round(total_attention * total_pool_capacity + min_num).int()


def headkv_update_kv(key_states, query_states, value_states, window_size, max_capacity_prompt,
                     kernel_size, pooling, head_capacity, layer_idx, sort_mode="canonical",
                     scale_mode="div"):
    """HeadKVCluster.update_kv, pyramidkv_utils.py:808-878: AdaKV with precomputed capacities
    ``head_capacity[layer][head]``."""
    bsz, num_heads, q_len, head_dim = query_states.shape
    base_capacity = max_capacity_prompt - window_size
    s = window_scores(query_states, key_states, window_size, "mean", scale_mode)
    s = pool_scores(s, pooling, kernel_size)
    if base_capacity > s.size(-1):                                             # :834-837
        meta = _ada_meta(num_heads, [q_len] * num_heads, q_len * num_heads, q_len, key_states.device)
        return key_states.reshape(-1, head_dim), value_states.reshape(-1, head_dim), meta
    stable = sort_mode == "canonical"
    _, sorted_idx = s.sort(dim=-1, descending=True, stable=stable)             # :840
    assert bsz == 1
    per_head = [sorted_idx[0, h, :int(head_capacity[layer_idx][h])] for h in range(num_heads)]  # :855
    kf, vf, lens = _flat_gather(key_states, value_states, per_head, window_size)
    meta = _ada_meta(num_heads, lens, sum(lens), max(lens), key_states.device)
    meta.head_capacity = [int(head_capacity[layer_idx][h]) for h in range(num_heads)]
    meta.indices = per_head
    return kf, vf, meta


def update_flatten_view(cache: torch.Tensor, state: torch.Tensor, head_lens: torch.Tensor,
                        cu_klen: torch.Tensor) -> torch.Tensor:
    """Decode-time flat-cache append, csrc/csrc/cuda_api.cu:11-85: for each head h copy its
    head_lens[h] old rows (starting at cu_klen[h]) to offset cu_klen[h]+h of a new
    [origin_len + H, D] buffer and put state[h] right after them."""
    H, D = state.shape
    out = torch.empty(cache.shape[0] + H, D, dtype=cache.dtype, device=cache.device)
    for h in range(H):
        n, src = int(head_lens[h]), int(cu_klen[h])
        dst = src + h                                   # cuda_api.cu:28
        out[dst:dst + n] = cache[src:src + n]           # :35-46
        out[int(cu_klen[h + 1]) + h] = state[h]         # :29,48-52
    return out
