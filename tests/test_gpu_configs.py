"""BASELINE.json configurations at full size, HIP path vs the oracle on the same seeded inputs (round-2 VERDICT item 1):

  config 3  SnapKV + H2O, budgets {128, 2048}, S = 32768          (H reduced so the CPU oracle finishes in seconds)
  config 5  Mistral-7B GQA: Q 32 heads, K/V 8 heads UN-EXPANDED, S = 32768, Ada-SnapKV floor 0.2
  one B = 8 end-to-end case

Bars: indices and compacted K/V BIT-IDENTICAL to the oracle (canonical tie order); scores within 1 ulp of the model dtype
on at most SCORE_FRAC of the elements.  The measured identical-selection rates are asserted as measured (1.0), not as a
loose lower bound; they are also written to gpurun_out/parity_report.json.
"""
import json
import os

import numpy as np
import pytest
import torch

from inputs import make_qkv, bits
from oracle import pkv_oracle as O
from score_bar import check_window_scores
from test_gpu_parity import DEV, ord16, score_diff, _report

pytestmark = pytest.mark.gpu
SCORE_FRAC = 2e-3
H2O_SCORE_FRAC = 1e-3


@pytest.fixture(scope="module")
def P():
    import pyramidkv_amd
    return pyramidkv_amd


def _identical(idx_hip, idx_ref):
    """(fraction of heads with the identical index SEQUENCE, fraction with the identical SET)."""
    a, b = idx_hip.cpu().long(), idx_ref.long()
    seq = (a == b).all(-1).float().mean().item()
    st = (torch.sort(a, -1).values == torch.sort(b, -1).values).all(-1).float().mean().item()
    return seq, st


@pytest.mark.parametrize("cap", [128, 2048])
def test_config3_snapkv_32k_vs_oracle(P, cap):
    """SnapKV update_kv at S = 32768 (BASELINE config 3): indices, K and V bit-identical to the oracle."""
    B, H, S, w = 1, 32, 32768, 8           # the headline tensor shape [1, 32, 32768, 128]
    q, k, v = make_qkv(B, H, S, 128, "bf16", "gauss", 3100 + cap)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cl = P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool")
    kc, vc = cl.update_kv(k.to(DEV), q.to(DEV), v.to(DEV), None, 1)
    _, _, idx = P.ops.compress(q.to(DEV), k.to(DEV), v.to(DEV), w, cap - w, "maxpool", 7, return_indices=True)
    kr, vr, ridx = O.snapkv_update_kv(k, q, v, w, cap, 7, "maxpool", return_indices=True)
    seq, st = _identical(idx, ridx)
    _report(f"config3/snapkv/S32768cap{cap}", dict(heads_identical_sequence=seq, heads_identical_set=st))
    # Measured (the bar is the measured value): budget 128 -> every head bit-identical; budget 2048 -> every head selects the
    # oracle's token SET and 31 of 32 heads the oracle's ORDER; in the other one, two tokens whose scores differ by one ulp between
    # the two implementations (exp / summation order) swap places.  The reference disagrees with ITSELF by as much: the real
    # pyramidkv_utils.py under ATEN_CPU_CAPABILITY=avx2 vs avx512 keeps the index sequence in 98.4 % of the heads at S = 8192,
    # budget 2048, bf16, scalar vs avx512 in 93.8 % (tools/reference_self_disagreement.py,
    # profiles/r04/reference_self_disagreement_bf16_S8192_budget2048.json).
    assert st == 1.0
    assert seq >= (1.0 if cap == 128 else 31 / 32)
    so = O.pool_scores(O.window_scores(q, k, w), "maxpool", 7)
    sel = ord16(torch.gather(so, -1, idx.cpu().long()))                  # oracle scores in the HIP order
    assert int((sel[..., 1:] - sel[..., :-1]).max()) <= 1, "HIP order is not the oracle's order up to 1-ulp score ties"
    same = (idx.cpu().long() == ridx).all(-1)[0]
    kcc, vcc = kc.cpu(), vc.cpu()
    for h in range(H):
        if bool(same[h]):
            assert torch.equal(kcc[0, h], kr[0, h]) and torch.equal(vcc[0, h], vr[0, h])
        else:                                                            # same rows, two of them swapped
            order_a, order_b = torch.argsort(idx[0, h].cpu().long()), torch.argsort(ridx[0, h])
            assert torch.equal(kcc[0, h, :cap - w][order_a], kr[0, h, :cap - w][order_b])
            assert torch.equal(kcc[0, h, cap - w:], kr[0, h, cap - w:])


def test_config2_pyramidkv_8k_all_32_layers_vs_oracle(P):
    """BASELINE config 2 at the model's own head count: PyramidKV budget 128, S = 8192, bf16, H = 32, EVERY one of the 32
    layer budgets through ``PyramidKVCluster.update_kv``: compacted K/V (hence the indices, order included) bit-identical to
    the oracle's update_kv."""
    B, H, S, w, cap, NL = 1, 32, 8192, 8, 128, 32
    q, k, v = make_qkv(B, H, S, 128, "bf16", "gauss", 2200)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    s = O.pool_scores(O.window_scores(q, k, w), "maxpool", 7)
    budgets = [O.pyramid_budget(cap, w, NL, layer, S) for layer in range(NL)]
    order = O.topk_canonical(s, max(kk for _, kk in budgets))          # prefix of the (value desc, index asc) order = top-k of any smaller k
    seqs, sets = [], []
    for layer, (branch, kk) in enumerate(budgets):
        assert branch == "pyramid"
        cl = P.PyramidKVCluster(num_hidden_layers=NL, layer_idx=layer, window_size=w, max_capacity_prompt=cap,
                                kernel_size=7, pooling="maxpool")
        assert cl.layer_budget(S) == (branch, kk)
        kc, vc = cl.update_kv(kd, qd, vd, None, 1)
        _, _, idx = P.ops.compress(qd, kd, vd, w, kk, "maxpool", 7, return_indices=True)
        ridx = order[..., :kk]
        seq, st = _identical(idx, ridx)
        seqs.append(seq)
        sets.append(st)
        kr, vr = O.gather_compact(k, v, ridx, w)
        if layer in (0, 31):                                            # the restatement's own update_kv, not only its stages
            kr2, vr2 = O.pyramidkv_update_kv(k, q, v, w, cap, 7, "maxpool", NL, layer)
            assert torch.equal(kr, kr2) and torch.equal(vr, vr2)
        assert st == 1.0, (layer, st)
        if seq == 1.0:
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), layer
    _report("config2/pyramidkv/S8192cap128/H32/all_layers",
            dict(heads_identical_sequence_min=min(seqs), heads_identical_set_min=min(sets), layers=NL,
                 heads_identical_sequence_mean=sum(seqs) / NL))
    assert min(seqs) == 1.0, seqs            # measured: every head of every layer in the oracle's order


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_config3_h2o_32k_vs_oracle(P, dt):
    """H2O update_kv at S = 32768 (BASELINE config 3; the reference itself cannot run it: it materialises 68.7 GB), budgets 128
    and 2048, B = 2 sequences x 4 heads, bf16 and fp16 (round 4; was one bf16 head): scores vs the row-blocked oracle (== the
    reference's arithmetic, checked against it at S <= 1024), indices and K/V vs the oracle's canonical selection."""
    B, H, S, w = 2, 4, 32768, 8
    q, k, v = make_qkv(B, H, S, 128, dt, "gauss", 3200)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    sg = P.ops.score_h2o(qd, kd, w)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    want = O.h2o_scores_blocked(q, k, w, block=512)
    frac, mx = score_diff(sg.cpu(), want)
    rep = dict(score_mismatch_frac=frac, score_max_ulp=mx)
    assert mx <= 1 and frac <= H2O_SCORE_FRAC, (frac, mx)
    for cap in (128, 2048):
        kc, vc, idx = P.ops.compress(qd, kd, vd, w, cap - w, None, 1, h2o=True, return_indices=True)
        assert torch.equal(idx.cpu().long(), O.topk_canonical(sg.cpu(), cap - w))       # exact under the kernel's scores
        ridx = O.topk_canonical(want, cap - w)
        kr, vr = O.gather_compact(k, v, ridx, w)
        ia = idx.cpu().long()
        seq_h = (ia == ridx).all(-1)
        set_h = (torch.sort(ia, -1).values == torch.sort(ridx, -1).values).all(-1)
        kv_h = (kc.cpu() == kr).flatten(2).all(-1) & (vc.cpu() == vr).flatten(2).all(-1)
        rep[f"cap{cap}"] = dict(heads_identical_sequence=float(seq_h.float().mean()), heads_identical_set=float(set_h.float().mean()),
                                kv_identical=float(kv_h.float().mean()))
        # a score is a sum over 32768 rows; where the kernels' score differs from the oracle's by one unit (<= 1e-3 of them)
        # two neighbours of the ranking may swap, and at budget 2048 the last place may change hands
        # measured: the oracle's token SET in every head except one fp16 head at budget 2048 (its order in 8 / 8 heads at budget 128,
        # 6 / 8 (bf16) and 1 / 8 (fp16) at budget 2048: fp16 H2O scores of 32k rows sit in a band a few units wide)
        assert float(set_h.float().mean()) >= (1.0 if (cap == 128 or dt == "bf16") else 0.875), (cap, set_h)
        if cap == 128:
            assert bool(seq_h.all()), seq_h
        assert bool((kv_h | ~seq_h).all()), "same index sequence but different K/V bits"
        # Why an order / set differs, per head and in the record (review of round 4): the two implementations' scores agree
        # within `mx` units of the last place, so the kernel can rank a before b against the oracle only when the ORACLE's own
        # scores of a and b lie within 2 * mx units - asserted for every adjacent pair of the kernel's order and for every
        # token the kernel keeps that the oracle drops (against the oracle's k-th score).
        wo = ord16(want)
        heads = []
        for b in range(B):
            for h in range(H):
                if bool(seq_h[b, h]):
                    continue
                mine, ref = ia[b, h].numpy(), ridx[b, h].numpy()
                so_mine = wo[b, h][mine]
                inv = int(max(0, (so_mine[1:] - so_mine[:-1]).max()))
                kth = int(wo[b, h][ref[-1]])
                extra = sorted(set(mine.tolist()) - set(ref.tolist()))
                short = [int(kth - wo[b, h][t]) for t in extra]
                first = int(np.argmax(mine != ref))
                heads.append(dict(batch=b, head=h, positions_that_differ=int((mine != ref).sum()), first_differing_position=first,
                                  largest_rise_of_the_oracle_scores_in_kernel_order_ulp=inv,
                                  tokens_kept_here_not_by_the_oracle=len(extra), their_oracle_score_below_the_kth_ulp=short[:8],
                                  oracle_score_band_of_the_last_64_kept_ulp=int(wo[b, h][ref[-64]] - kth) if len(ref) >= 64 else None))
                assert inv <= 2 * max(1, mx), (cap, b, h, inv)
                assert all(0 <= s_ <= 2 * max(1, mx) for s_ in short), (cap, b, h, short)
        rep[f"cap{cap}"]["heads_in_another_order"] = heads
    _report(f"config3/h2o/S32768/B2H4/{dt}", rep)


def test_config5_mistral_gqa_adakv_32k_vs_oracle(P):
    """BASELINE config 5 exactly: Mistral-7B attention shapes (32 query heads, 8 KV heads, D = 128), K/V handed over
    UN-EXPANDED, S = 32768, Ada-SnapKV (floor 0.2, normalize, maxpool-7, window 8, budget 128): head budgets, var-len
    metadata and the flat K/V bit-identical to the oracle run on the repeat_kv-expanded tensors."""
    Hq, Hkv, S, w, cap = 32, 8, 32768, 8, 128
    g = Hq // Hkv
    q, k8, v8 = make_qkv(1, Hq, S, 128, "bf16", "gauss", 5100)
    k_un, v_un = k8[:, ::g].contiguous(), v8[:, ::g].contiguous()                       # [1, 8, S, D]
    k_exp = k_un[:, :, None].expand(1, Hkv, g, S, 128).reshape(1, Hq, S, 128).contiguous()
    v_exp = v_un[:, :, None].expand(1, Hkv, g, S, 128).reshape(1, Hq, S, 128).contiguous()
    cl = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2,
                        normalize=True, layer_idx=0, num_hidden_layers=32)
    kf, vf = cl.update_kv(k_un.to(DEV), q.to(DEV), v_un.to(DEV))
    kr, vr, meta = O.adakv_update_kv(k_exp, q, v_exp, w, cap, 7, "maxpool", 0.2, True)
    same_lens = cl.head_lens.cpu().tolist() == meta.head_lens.tolist()
    same_kv = same_lens and bool(torch.equal(kf.cpu(), kr) and torch.equal(vf.cpu(), vr))
    _report("config5/adakv_gqa_32k", dict(head_lens_identical=same_lens, kv_identical=same_kv,
                                          head_lens=cl.head_lens.cpu().tolist()))
    assert same_lens and cl.cu_klen.cpu().tolist() == meta.cu_klen.tolist()
    assert cl.klen_sum == meta.klen_sum and cl.max_seqlen_k == meta.max_seqlen_k
    assert same_kv
    # the expanded hand-over (the reference's contract) gives the same bytes
    cl2 = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2,
                         normalize=True, layer_idx=0, num_hidden_layers=32)
    kf2, vf2 = cl2.update_kv(k_exp.to(DEV), q.to(DEV), v_exp.to(DEV))
    assert torch.equal(kf2, kf) and torch.equal(vf2, vf)


def test_batch8_end_to_end_vs_oracle(P):
    """B = 8 sequences in one call (the upper end of the north star's batch range): PyramidKV layer budgets, indices and
    K/V bit-identical to the oracle for every (batch, head)."""
    B, H, S, w, cap = 8, 4, 8192, 8, 128
    q, k, v = make_qkv(B, H, S, 128, "bf16", "gauss", 8800)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    for layer in (0, 31):
        cl = P.PyramidKVCluster(num_hidden_layers=32, layer_idx=layer, window_size=w, max_capacity_prompt=cap,
                                kernel_size=7, pooling="maxpool")
        kc, vc = cl.update_kv(kd, qd, vd, None, 1)
        kr, vr, ridx = O.pyramidkv_update_kv(k, q, v, w, cap, 7, "maxpool", 32, layer, return_indices=True)
        _, _, idx = P.ops.compress(qd, kd, vd, w, ridx.shape[-1], "maxpool", 7, return_indices=True)
        seq, st = _identical(idx, ridx)
        _report(f"batch8/pyramidkv/layer{layer}", dict(rows_identical_sequence=seq, rows_identical_set=st))
        assert seq == 1.0 and st == 1.0
        assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)


def _row_with_total(rng, L, total_units, hi):
    """L integers in [0, hi) whose sum is exactly total_units (units of 2^-7)."""
    n = rng.integers(0, hi, size=L).astype(np.int64)
    diff, i = int(total_units - n.sum()), 0
    while diff != 0:
        step = int(np.clip(diff, -n[i], hi - 1 - n[i]))
        n[i] += step
        diff -= step
        i += 1
    return n


def adakv_tie_scores(dt):
    """Score rows [H, L] made of multiples of 2^-7 below 2 (exact in bf16 and fp16; every partial sum is exact in fp32 in
    ANY summation order) whose row totals sit exactly on rounding ties of the model dtype (bf16: 257, 259, 128.5 - the
    midpoints of 256|258, 258|260, 128|129; fp16: 2049, 2051, 1024.5) or just beside one."""
    L = 4088
    totals = {torch.bfloat16: [257.0, 259.0, 128.5, 257.0078125], torch.float16: [2049.0, 2051.0, 1024.5, 2049.0078125]}[dt]
    rng = np.random.default_rng(5)
    rows = []
    for t in totals:
        units = int(round(t * 128))
        rows.append(_row_with_total(rng, L, units, max(2, min(256, 2 * units // L + 2))).astype(np.float64) / 128.0)
    s = torch.tensor(np.stack(rows), dtype=torch.float32).to(dt)
    assert torch.equal(s.double().sum(-1), torch.tensor(totals, dtype=torch.float64))       # the cast rounded nothing
    return s


def test_adakv_ratio_sums_at_rounding_ties(P):
    """Ada-SnapKV normalisation (pyramidkv_utils.py:709-711) at its rounding points: `.sum()` -> model dtype (ties to even),
    the model-dtype division, the scaling of every score and the resulting head budgets must equal the oracle's bit for bit.
    The kernel accumulates the sums in fp64, ATen in fp32 in its own order - on these inputs both are exact, so the test
    pins the roundings and nothing else."""
    for dt in (torch.bfloat16, torch.float16):
        s = adakv_tie_scores(dt)
        for base, floor in ((120, 0.2), (500, 0.0)):
            sidx_ref, cap_ref = O.adakv_head_capacity(s[None], base, floor, True, "canonical")
            si, sv = P.ops.sort_rows(s.to(DEV))
            assert torch.equal(si.cpu().long(), sidx_ref[0])
            capd = P.ops.ada_budget(sv, base, floor, True)
            assert capd.cpu().tolist() == cap_ref[0].tolist(), (str(dt), base, capd.cpu().tolist(), cap_ref[0].tolist())


def test_release_library_ignores_ablation_knob(P):
    """VERDICT hygiene item: PKV_LOGITS_ABLATE (an ablation that produces WRONG results) must not exist in the release
    library.  A subprocess with the variable set must produce the same scores as this process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert P._native.lib.pkv_debug_build() == 0
    code = ("import sys,torch;sys.path.insert(0,%r);sys.path.insert(0,%r);import pyramidkv_amd as P;from inputs import make_qkv;"
            "q,k,v=make_qkv(1,2,2048,128,'bf16','gauss',5);s=P.ops.score_window(q.cuda(),k.cuda(),8,'maxpool',7).cpu();"
            "print(int(s.view(torch.int16).long().sum()))") % (root, os.path.join(root, "tests"))
    outs = []
    for env_extra in ({}, {"PKV_LOGITS_ABLATE": "2"}):
        env = dict(os.environ, **env_extra)
        env.pop("PKV_LIB", None)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]
    buf = torch.zeros(16, dtype=torch.int64, device=DEV)
    assert P._native.lib.pkv_debug_topk_trace(buf.data_ptr()) == -5          # PKV_ERR_UNSUPPORTED in the release build
    assert P._native.lib.pkv_debug_topk_trace(None) == 0


# ----------------------------------------------------------------------------------------- LOOK-M pivot merge (merge_kv)
def _merge_close(a, b):
    """fraction of elements that differ, and the largest difference in ulps of the model dtype."""
    da = np.abs(ord16(a).astype(np.int64) - ord16(b).astype(np.int64))
    return float((da > 0).mean()), int(da.max())


@pytest.mark.parametrize("dt,kind", [("bf16", "lattice"), ("fp16", "lattice"), ("bf16", "gauss"), ("fp16", "gauss")])
def test_merge_compact_vs_oracle_given_indices(P, dt, kind):
    """pkv_merge_compact vs oracle.merge_kv (== the reference's merge_kv, pinned by the *_merge golden fixtures) on the SAME
    indices.  Lattice inputs: every norm^2 / dot product is exact in fp32 in any order -> bit-identical.  Gaussian inputs:
    the fp32 accumulation orders of the norm and the similarity GEMM differ between any two implementations; a pivot may
    flip where two similarities tie within one rounding - the fraction of differing output elements is bounded."""
    B, H, S, w, k = 2, 4, 1500, 8, 40
    q, K, V = make_qkv(B, H, S, 128, dt, kind, 7100)
    g = torch.Generator().manual_seed(5)
    idx = torch.stack([torch.stack([torch.randperm(S - w, generator=g)[:k] for _ in range(H)]) for _ in range(B)])
    kr, vr = O.merge_kv(K, V, idx, w, "pivot")
    km, vm = P.ops.merge_compact(K.to(DEV), V.to(DEV), idx.to(DEV).int(), w)
    fk, uk = _merge_close(km.cpu(), kr)
    fv, uv = _merge_close(vm.cpu(), vr)
    _report(f"merge/{dt}/{kind}", dict(k_mismatch_frac=fk, k_max_ulp=uk, v_mismatch_frac=fv, v_max_ulp=uv))
    if kind == "lattice":
        assert torch.equal(km.cpu(), kr) and torch.equal(vm.cpu(), vr)
    else:
        assert fk == 0.0 and fv == 0.0, (fk, fv)          # measured: bit-identical on these seeds as well


def test_merge_count_rounding_above_256(P):
    """More than 256 dropped rows reach one kept row: the reference divides by the count ROUNDED to the model dtype
    (scatter_reduce keeps its count tensor in bf16).  One kept key is planted so that ~everything merges into it."""
    B, H, S, w, k = 1, 2, 1200, 8, 16
    q, K, V = make_qkv(B, H, S, 128, "bf16", "lattice", 7200)
    idx = torch.arange(k)[None, None, :].repeat(B, H, 1) * 3
    K[:, :, :] = (K.float() * 0.25).to(K.dtype)
    K[:, :, 9] = 1.0                                      # selected position 9 (= idx 3): the pivot of most rows
    K[:, :, 200:] += 1.0                                  # rows 200.. point the same way
    kr, vr = O.merge_kv(K, V, idx, w, "pivot")
    km, vm = P.ops.merge_compact(K.to(DEV), V.to(DEV), idx.to(DEV).int(), w)
    assert torch.equal(km.cpu(), kr) and torch.equal(vm.cpu(), vr)


def test_merge_zero_norm_dropped_keys_follow_torch_max(P):
    """A dropped key row of zeros has norm 0: x / n is 0/0 = NaN, every similarity of that row is NaN, and torch.max (:151)
    returns the FIRST NaN - kept row 0 (round-2 advisor finding: the kernel left such rows un-merged)."""
    B, H, S, w, k = 1, 2, 700, 8, 24
    q, K, V = make_qkv(B, H, S, 128, "bf16", "lattice", 7300)
    idx = (torch.arange(k)[None, None, :].repeat(B, H, 1) * 5 + 1)
    for pos in (0, 333, 650):                             # never selected (selected positions are 1 mod 5 below 120)
        K[:, :, pos] = 0
    kr, vr = O.merge_kv(K, V, idx, w, "pivot")
    km, vm = P.ops.merge_compact(K.to(DEV), V.to(DEV), idx.to(DEV).int(), w)
    assert np.array_equal(bits(km), bits(kr)) and np.array_equal(bits(vm), bits(vr))


def test_merge_zero_norm_kept_key_follows_torch_max(P):
    """A KEPT key row of zeros: its unit-norm form is NaN, so every dropped row's similarity to that column is NaN and
    torch.max (:151) stops at the first NaN - all dropped rows of the head merge into the first such kept row.  Also a
    non-finite dropped row next to it (its similarities are NaN everywhere: kept row 0).  The pivot kernel takes its NaN-aware
    path only for heads / waves that can see such a row (flag from the targets kernel, ballot over the wave's rows)."""
    B, H, S, w, k = 1, 3, 900, 8, 40
    q, K, V = make_qkv(B, H, S, 128, "bf16", "lattice", 7350)
    idx = (torch.arange(k)[None, None, :].repeat(B, H, 1) * 7 + 2)
    K[:, 1, int(idx[0, 1, 5])] = 0                        # head 1: the 6th selected key (kept row w + 5) has zero norm
    K[:, 2, 500] = 0                                      # head 2: a dropped zero row (500 is not 2 mod 7 below 282... not selected)
    assert 500 not in idx[0, 2].tolist()
    kr, vr = O.merge_kv(K, V, idx, w, "pivot")
    km, vm = P.ops.merge_compact(K.to(DEV), V.to(DEV), idx.to(DEV).int(), w)
    assert np.array_equal(bits(km), bits(kr)) and np.array_equal(bits(vm), bits(vr))


def test_merge_golden_fixtures_and_clusters(P):
    """The REAL reference's merged K/V (tests/golden/*_merge.npz) vs the HIP clusters with merge='pivot': bit-identical on
    the tie-free fixtures (selection fully determined); on the others the clusters must equal the oracle's merge of the
    HIP path's own indices (the reference's CPU top-k breaks ties arbitrarily, and the merge depends on the indices)."""
    gold = os.path.join(os.path.dirname(__file__), "golden")
    cases = [c for c in json.load(open(os.path.join(gold, "index.json")))["cases"] if c.get("merge")]
    assert len(cases) >= 7
    for c in cases:
        z = np.load(os.path.join(gold, c["name"] + ".npz"))
        q, k, v = make_qkv(c["B"], c["H"], c["S"], 128, c["dtype"], c["kind"], c["seed"])
        qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
        pol, w, cap = c["policy"], c["w"], c["cap"]
        if pol == "snapkv":
            cl = P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=c["ks"], pooling=c["pool"], merge="pivot")
            kk, h2o = cap - w, False
        elif pol == "pyramidkv":
            cl = P.PyramidKVCluster(num_hidden_layers=c["layers"], layer_idx=c["layer"], window_size=w, max_capacity_prompt=cap,
                                    kernel_size=c["ks"], pooling=c["pool"], merge="pivot")
            kk, h2o = cl.layer_budget(c["S"])[1], False
        elif pol == "h2o":
            cl = P.H2OKVCluster(window_size=w, max_capacity_prompt=cap, merge="pivot")
            kk, h2o = cap - w, True
        else:
            cl = P.StreamingLLMKVCluster(window_size=w, max_capacity_prompt=cap, merge="pivot")
            kk, h2o = cap - w, False
        km, vm = cl.update_kv(kd, qd, vd, None, 1)
        assert tuple(km.shape) == z["kc"].shape
        if c.get("tie_free") or pol == "streamingllm":
            assert np.array_equal(bits(km), z["kc"]) and np.array_equal(bits(vm), z["vc"]), c["name"]
            continue
        pool = None if c["pool"] == "none" else c["pool"]
        idx = P.ops.select(qd, kd, w, kk, pool, c["ks"] or 1, h2o=h2o).cpu().long()
        kr, vr = O.merge_kv(k, v, idx, w, "pivot")
        fk, _ = _merge_close(km.cpu(), kr)
        fv, _ = _merge_close(vm.cpu(), vr)
        assert fk <= (0.0 if c["kind"] == "lattice" else 0.02) and fv <= (0.0 if c["kind"] == "lattice" else 0.02), (c["name"], fk, fv)


def test_merge_randomised_configs(P):
    """16 seeded random merge configurations - batch, heads, un-expanded GQA K/V (kv_group), strided [B,S,H,D] storage, S not a
    multiple of anything, window, k, dtype, lattice inputs (every norm / dot product exact in fp32 in any order): the merged
    K/V are bit-identical to oracle.merge_kv run on the repeat_kv-expanded tensors."""
    rng = np.random.default_rng(17)
    for case in range(16):
        B = int(rng.integers(1, 3))
        g = int(rng.choice([1, 1, 2, 4]))
        Hk = int(rng.integers(1, 4))
        H = Hk * g
        w = int(rng.choice([1, 4, 8, 16, 32]))
        S = int(rng.integers(w + 60, 3000))
        k = int(rng.integers(1, min(S - w, 300) + 1))
        dt = "bf16" if case % 2 else "fp16"
        _, K, V = make_qkv(B, H, S, 128, dt, "lattice", 7300 + case)
        K[:, :, ::3] = (K[:, :, ::3].float() * 0.5).to(K.dtype)                    # different norms, still exact
        k_un, v_un = K[:, ::g].contiguous(), V[:, ::g].contiguous()
        k_exp = k_un[:, :, None].expand(B, Hk, g, S, 128).reshape(B, H, S, 128).contiguous()
        v_exp = v_un[:, :, None].expand(B, Hk, g, S, 128).reshape(B, H, S, 128).contiguous()
        gen = torch.Generator().manual_seed(case)
        idx = torch.stack([torch.stack([torch.randperm(S - w, generator=gen)[:k] for _ in range(H)]) for _ in range(B)])
        kr, vr = O.merge_kv(k_exp, v_exp, idx, w, "pivot")
        if case % 3 == 0:      # [B, S, Hk, D] storage seen through a transposed view
            kd = k_un.to(DEV).transpose(1, 2).contiguous().transpose(1, 2)
            vd = v_un.to(DEV).transpose(1, 2).contiguous().transpose(1, 2)
        else:
            kd, vd = k_un.to(DEV), v_un.to(DEV)
        km, vm = P.ops.merge_compact(kd, vd, idx.to(DEV).int(), w, kv_group=g)
        assert torch.equal(km.cpu(), kr) and torch.equal(vm.cpu(), vr), (case, B, H, g, S, w, k, dt)


def test_merge_cluster_unexpanded_gqa_equals_expanded(P):
    """SnapKVCluster(merge='pivot') handed K/V before repeat_kv (4 query heads per KV head) == the same cluster on the
    expanded tensors == the oracle on the HIP path's own indices."""
    B, Hq, Hkv, S, w, cap = 1, 8, 2, 2048, 8, 64
    g = Hq // Hkv
    q, k8, v8 = make_qkv(B, Hq, S, 128, "bf16", "gauss", 7400)
    k_un, v_un = k8[:, ::g].contiguous(), v8[:, ::g].contiguous()
    k_exp = k_un[:, :, None].expand(B, Hkv, g, S, 128).reshape(B, Hq, S, 128).contiguous()
    v_exp = v_un[:, :, None].expand(B, Hkv, g, S, 128).reshape(B, Hq, S, 128).contiguous()
    cl = P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool", merge="pivot")
    ka, va = cl.update_kv(k_un.to(DEV), q.to(DEV), v_un.to(DEV), None, g)
    kb, vb = cl.update_kv(k_exp.to(DEV), q.to(DEV), v_exp.to(DEV), None, g)
    assert torch.equal(ka, kb) and torch.equal(va, vb)
    idx = P.ops.select(q.to(DEV), k_un.to(DEV), w, cap - w, "maxpool", 7, kv_group=g).cpu().long()
    kr, vr = O.merge_kv(k_exp, v_exp, idx, w, "pivot")
    assert torch.equal(ka.cpu(), kr) and torch.equal(va.cpu(), vr)


@pytest.mark.parametrize("D", [64, 256])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_merge_head_sizes_64_and_256(P, D, dt):
    """The LOOK-M merge at head sizes 64 and 256 (kernels templated on the MFMA k-steps D / 32): lattice inputs (norms and dot
    products exact in fp32 in any order) -> bit-identical to oracle.merge_kv, incl. un-expanded GQA K/V, a [B,S,H,D] view,
    more kept rows than one LDS tile of the pivot kernel (k + w = 208 > 144 / 72) and a group of more than 256 rows."""
    B, Hk, g, S, w, k = 2, 2, 2, 2203, 8, 200
    H = Hk * g
    _, K, V = make_qkv(B, H, S, D, dt, "lattice", 7600 + D)
    K[:, :, ::3] = (K[:, :, ::3].float() * 0.5).to(K.dtype)
    K[:, :, 11] = 1.0                                       # a kept key most of the tail points at: one large group
    K[:, :, 1500:] += 1.0
    k_un, v_un = K[:, ::g].contiguous(), V[:, ::g].contiguous()
    k_exp = k_un[:, :, None].expand(B, Hk, g, S, D).reshape(B, H, S, D).contiguous()
    v_exp = v_un[:, :, None].expand(B, Hk, g, S, D).reshape(B, H, S, D).contiguous()
    gen = torch.Generator().manual_seed(D)
    idx = torch.stack([torch.stack([torch.randperm(1388, generator=gen)[:k] + 12 for _ in range(H)]) for _ in range(B)])
    idx[:, :, 0] = 11
    kr, vr = O.merge_kv(k_exp, v_exp, idx, w, "pivot")
    kd = k_un.to(DEV).transpose(1, 2).contiguous().transpose(1, 2)            # [B,S,Hk,D] storage
    km, vm = P.ops.merge_compact(kd, v_un.to(DEV), idx.to(DEV).int(), w, kv_group=g)
    fk, _ = _merge_close(km.cpu(), kr)
    fv, _ = _merge_close(vm.cpu(), vr)
    _report(f"merge_head_size/{D}/{dt}", dict(k_mismatch_frac=fk, v_mismatch_frac=fv))
    # unit-norm rows are not lattice values: a similarity may round differently under another fp32 order and move one
    # dropped row to another pivot (two output rows); measured 0.0 on these seeds
    assert fk <= 0.005 and fv <= 0.005, (D, dt, fk, fv)
    # the cluster on Gaussian inputs: the HIP path's own indices, merged by the oracle
    q, kg, vg = make_qkv(1, 4, 1500, D, dt, "gauss", 7700 + D)
    cl = P.SnapKVCluster(window_size=w, max_capacity_prompt=64, kernel_size=7, pooling="maxpool", merge="pivot")
    ka, va = cl.update_kv(kg.to(DEV), q.to(DEV), vg.to(DEV), None, 1)
    sel = P.ops.select(q.to(DEV), kg.to(DEV), w, 64 - w, "maxpool", 7).cpu().long()
    kr, vr = O.merge_kv(kg, vg, sel, w, "pivot")
    fk, _ = _merge_close(ka.cpu(), kr)
    fv, _ = _merge_close(va.cpu(), vr)
    assert fk <= 0.02 and fv <= 0.02, (D, dt, fk, fv)


def test_merge_many_kept_rows_and_one_huge_group(P):
    """Budget 2048-sized kept sets (k + w = 2056 kept rows per head) and a group larger than the scatter kernel's list pass
    (2048 entries): every dropped row of the tail merges into one planted key.  Lattice inputs -> bit-identical."""
    B, H, S, w, k = 1, 2, 9000, 8, 2048
    _, K, V = make_qkv(B, H, S, 128, "bf16", "lattice", 7800)
    K[:, :, :] = (K.float() * 0.25).to(K.dtype)
    K[:, :, 5] = 1.0
    K[:, :, 4000:] += 1.0
    gen = torch.Generator().manual_seed(3)
    idx = torch.stack([torch.stack([torch.randperm(3494, generator=gen)[:k] + 6 for _ in range(H)]) for _ in range(B)])
    idx[:, :, 7] = 5
    kr, vr = O.merge_kv(K, V, idx, w, "pivot")
    km, vm = P.ops.merge_compact(K.to(DEV), V.to(DEV), idx.to(DEV).int(), w)
    fk, _ = _merge_close(km.cpu(), kr)
    fv, _ = _merge_close(vm.cpu(), vr)
    _report("merge_many_kept_rows", dict(k_mismatch_frac=fk, v_mismatch_frac=fv))
    assert fk <= 0.002 and fv <= 0.002, (fk, fv)
    # the planted row itself: its group has more than 2048 entries (several list passes) and its count rounds in bf16
    j = w + 7
    assert torch.equal(km.cpu()[:, :, j], kr[:, :, j])


def test_merge_long_sequence_more_than_8192_kept_rows(P):
    """S = 40 001 and 8 208 kept rows per head: the bucket kernel's LDS histogram takes two ranges of kept rows, the scatter
    kernel's position bitmap several words per thread, the pivot kernel 57 LDS tiles.  D = 64 keeps the oracle's
    (dropped x kept) similarity matrix at 0.5 GB."""
    B, H, S, w, k, D = 1, 1, 40001, 8, 8200, 64
    _, K, V = make_qkv(B, H, S, D, "bf16", "lattice", 7900)
    K[:, :, ::3] = (K[:, :, ::3].float() * 0.5).to(K.dtype)
    gen = torch.Generator().manual_seed(9)
    idx = torch.randperm(S - w, generator=gen)[:k][None, None, :]
    kr, vr = O.merge_kv(K, V, idx, w, "pivot")
    km, vm = P.ops.merge_compact(K.to(DEV), V.to(DEV), idx.to(DEV).int(), w)
    fk, _ = _merge_close(km.cpu(), kr)
    fv, _ = _merge_close(vm.cpu(), vr)
    _report("merge_long_sequence", dict(k_mismatch_frac=fk, v_mismatch_frac=fv))
    assert fk <= 0.002 and fv <= 0.002, (fk, fv)


# ----------------------------------------------------------------------------------------- head sizes other than 128
@pytest.mark.parametrize("D", [64, 256])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_head_sizes_64_and_256_window_policies(P, D, dt):
    """The window policies, the gather and the flat var-len path at head sizes 64 and 256 (the reference is shape-generic,
    pyramidkv_utils.py:317; its supported model families use 128): scores within 1 ulp of the oracle, indices == canonical top-k
    of the kernel's own scores, K/V == exact gather incl. un-expanded GQA and strided views; Ada-SnapKV budgets and flat K/V
    from the kernel's own scores == the oracle's arithmetic on them.  H2O and the LOOK-M merge run at these head sizes too (round 3)."""
    B, Hk, g, S, w, k = 2, 2, 2, 3001, 8, 77
    H = Hk * g
    q, kf, vf = make_qkv(B, H, S, D, dt, "gauss", 9500 + D)
    k_un, v_un = kf[:, ::g].contiguous(), vf[:, ::g].contiguous()
    k_exp = k_un[:, :, None].expand(B, Hk, g, S, D).reshape(B, H, S, D).contiguous()
    v_exp = v_un[:, :, None].expand(B, Hk, g, S, D).reshape(B, H, S, D).contiguous()
    qd = q.to(DEV)
    kd = k_un.to(DEV).transpose(1, 2).contiguous().transpose(1, 2)            # [B,S,Hk,D] storage
    vd = v_un.to(DEV)
    for pool, ks in (("maxpool", 7), ("avgpool", 5)):
        sg = P.ops.score_window(qd, kd, w, pool, ks, kv_group=g)
        check_window_scores(q, k_exp, w, pool, ks, "sum", sg.cpu(), lambda: P.ops.score_window(qd, kd, w, None, 1, kv_group=g).cpu(),
                            frac_bar=SCORE_FRAC, what=(D, dt, pool))
        kc, vc, idx = P.ops.compress(qd, kd, vd, w, k, pool, ks, kv_group=g, return_indices=True)
        want = O.topk_canonical(sg.cpu(), k)
        assert torch.equal(idx.cpu().long(), want)
        kr, vr = O.gather_compact(k_exp, v_exp, want, w)
        assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)
    # a large budget (8 rows per lane in the gather) and StreamingLLM
    kc, vc, idx = P.ops.compress(qd, kd, vd, w, 1500, "maxpool", 7, kv_group=g, return_indices=True)
    kr, vr = O.gather_compact(k_exp, v_exp, idx.cpu().long(), w)
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)
    kc, vc = P.StreamingLLMKVCluster(window_size=60, max_capacity_prompt=64).update_kv(k_exp.to(DEV), qd, v_exp.to(DEV), None, 1)
    kr, vr = O.streamingllm_update_kv(k_exp, q, v_exp, 60, 64)
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)
    # Ada-SnapKV (batch 1)
    q1, k1, v1 = q[:1], k_exp[:1].contiguous(), v_exp[:1].contiguous()
    cl = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=64, floor=0.2, normalize=True)
    kfl, vfl = cl.update_kv(k1.to(DEV), q1.to(DEV), v1.to(DEV))
    s1 = P.ops.score_window(q1.to(DEV), k1.to(DEV), w, "maxpool", 7, "mean").cpu()
    sidx, caps = O.adakv_head_capacity(s1, 64 - w, 0.2, True)
    assert cl.head_lens.cpu().tolist() == [int(c) + w for c in caps[0]]
    kr, vr, _ = O._flat_gather(k1, v1, [sidx[0, h, :int(caps[0, h])] for h in range(H)], w)
    assert torch.equal(kfl.cpu(), kr) and torch.equal(vfl.cpu(), vr)
    # decode-time flat append at this head size
    cache = P.DynamicCacheSplitHeadFlatten()
    cache.update(kfl, vfl, 0)
    nk = torch.randn(1, H, 1, D).to(kfl.dtype)
    knew, _ = cache.update(nk.to(DEV), nk.to(DEV), 0, {"head_lens": cl.head_lens, "cu_klen": cl.cu_klen})
    want = O.update_flatten_view(kfl.cpu(), nk[0, :, 0], cl.head_lens.cpu(), cl.cu_klen.cpu())
    assert torch.equal(knew.cpu(), want)
    # H2O at this head size (round 3; k-steps = D / 32 like the window scores): scores vs the oracle, update_kv vs the
    # canonical selection under the kernel's scores
    from test_gpu_parity import H2O_MISMATCH_FRAC
    sg = P.ops.score_h2o(qd, kd, w, kv_group=g).cpu()
    want = O.h2o_scores(q, k_exp, w)
    frac, mx = score_diff(sg, want)
    assert mx <= 1 and frac <= H2O_MISMATCH_FRAC, (D, dt, frac, mx)
    kc, vc = P.H2OKVCluster(window_size=w, max_capacity_prompt=64).update_kv(kd, qd, vd, None, g)
    kr, vr = O.gather_compact(k_exp, v_exp, O.topk_canonical(sg, 64 - w), w)
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)
    with pytest.raises(ValueError):
        P.ops.score_window(torch.zeros(1, 1, 64, 96, dtype=torch.bfloat16, device=DEV), torch.zeros(1, 1, 64, 96, dtype=torch.bfloat16, device=DEV), 8)


def test_h2o_strided_views_gqa_batch(P):
    """H2O scores with token-major Q/K views (row stride = heads * D), un-expanded GQA K and B = 2; S not a multiple of the
    64-row LDS tile.  The tile loads are raw buffer loads whose extent comes from the strides: this is their test."""
    from test_gpu_parity import H2O_MISMATCH_FRAC
    B, H, G, S, w = 2, 4, 2, 1100, 8
    g = torch.Generator().manual_seed(77)
    q_tm = torch.randn(B, S, H, 128, generator=g).to(torch.bfloat16)             # [B,S,H,D] as the projection writes it
    k_tm = torch.randn(B, S, H // G, 128 + 64, generator=g).to(torch.bfloat16)   # padded rows: stride (H/G) * 192
    q = q_tm.permute(0, 2, 1, 3)
    k = k_tm[..., 32:160].permute(0, 2, 1, 3)                                    # 64-byte offset into every row
    want = O.h2o_scores(q.contiguous(), k.contiguous().repeat_interleave(G, dim=1), w)
    got = P.ops.score_h2o(q.to(DEV), k_tm.to(DEV)[..., 32:160].permute(0, 2, 1, 3), w, kv_group=G).cpu()
    frac, mx = score_diff(got, want)
    assert mx <= 1 and frac <= H2O_MISMATCH_FRAC, (frac, mx)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_h2o_tile_boundaries(P, dt):
    """H2O scores around every boundary of the kernels' tiling (64-row LDS tiles, 256 resident rows per workgroup, the
    masked w x w corner straddling tiles): S from 9 to 1030, windows 1 / 8 / 32 / 64, GQA groups 1 and 2."""
    from test_gpu_parity import H2O_MISMATCH_FRAC
    worst, bad, total = 0.0, 0, 0
    for S in (9, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 320, 511, 513, 767, 1030):
        for w in (1, 8, 32, 64):
            if w >= S:
                continue
            G = 2 if (S + w) % 2 else 1
            q, k, _ = make_qkv(1, 2, S, 128, dt, "gauss", 1000 + S + w)
            kk = k[:, ::G].contiguous()
            want = O.h2o_scores(q, kk.repeat_interleave(G, dim=1), w)
            got = P.ops.score_h2o(q.to(DEV), kk.to(DEV), w, kv_group=G).cpu()
            frac, mx = score_diff(got, want)
            worst = max(worst, frac)
            bad += round(frac * got.numel())
            total += got.numel()
            # a tiling bug shows as many or large differences (this test found NaN scores for S < w + 12, where a lane
            # group saw only masked keys); what is allowed is the one-unit rounding noise of a sum of S rounded
            # probabilities: scattered, either sign, ~7e-5 of the elements in bf16 and ~7e-4 in fp16 (8x finer grid) -
            # a handful per tensor of this size, and the pooled rate below
            assert mx <= 1 and frac <= max(H2O_MISMATCH_FRAC, 8.0 / got.numel()), (S, w, G, frac, mx)
    _report(f"h2o_tile_boundaries/{dt}", dict(worst_mismatch_frac=worst, pooled_mismatch_frac=bad / total))
    assert bad / total <= H2O_MISMATCH_FRAC, (bad, total)          # measured: bf16 8.8e-5, fp16 7.3e-4


def test_one_million_token_prompt(P):
    """S = 1 048 576 (H = 2, bf16): 64-bit offsets everywhere, the long-row top-k (32 segments), 8192 K-scan stages per head.
    Scores vs the oracle (O(w S D) on the CPU), selection = canonical top-k of the kernel's scores, K/V exact copies."""
    B, H, S, w, cap = 1, 2, 1 << 20, 8, 128
    g = torch.Generator().manual_seed(123)
    q = torch.randn(B, H, S, 128, generator=g).to(torch.bfloat16)
    k = torch.randn(B, H, S, 128, generator=g).to(torch.bfloat16)
    v = torch.randn(B, H, S, 128, generator=g).to(torch.bfloat16)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    got = P.ops.score_window(qd, kd, w, "maxpool", 7).cpu()
    rep1m = check_window_scores(q, k, w, "maxpool", 7, "sum", got, lambda: P.ops.score_window(qd, kd, w, None, 1).cpu(), frac_bar=SCORE_FRAC)
    frac, mx = rep1m["mismatch_frac"], rep1m["max_ulp"]
    kc, vc, idx = P.ops.compress(qd, kd, vd, w, cap - w, "maxpool", 7, return_indices=True)
    idx = idx.cpu().long()
    assert torch.equal(idx, O.topk_canonical(got, cap - w))
    kr, vr = O.gather_compact(k, v, idx, w)
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)
    kc2, vc2 = P.StreamingLLMKVCluster(window_size=w, max_capacity_prompt=cap).update_kv(kd, qd, vd, None, 1)
    kr, vr = O.streamingllm_update_kv(k, q, v, w, cap)
    assert torch.equal(kc2.cpu(), kr) and torch.equal(vc2.cpu(), vr)
    _report("one_million_tokens", dict(score_mismatch_frac=frac, score_max_ulp=mx))


def test_batch_64_by_32_heads(P):
    """2048 (batch, head) rows in one call (B = 64, H = 32, S = 1024, un-expanded GQA 4): grid dimensions and row offsets far
    from the single-sequence case; selection and K/V vs the oracle on the kernel's scores, scores vs the oracle."""
    B, H, G, S, w, cap = 64, 32, 4, 1024, 8, 72
    q, k, v = make_qkv(B, H, S, 128, "bf16", "lattice", 640)
    kk, vv = k[:, ::G].contiguous(), v[:, ::G].contiguous()
    ke, ve = kk.repeat_interleave(G, dim=1), vv.repeat_interleave(G, dim=1)
    qd, kd, vd = q.to(DEV), kk.to(DEV), vv.to(DEV)
    want = O.pool_scores(O.window_scores(q, ke, w), "avgpool", 5)
    got = P.ops.score_window(qd, kd, w, "avgpool", 5, kv_group=G).cpu()
    frac, mx = score_diff(got, want)
    assert mx <= 1 and frac <= SCORE_FRAC, (frac, mx)
    kc, vc, idx = P.ops.compress(qd, kd, vd, w, cap - w, "avgpool", 5, kv_group=G, return_indices=True)
    idx = idx.cpu().long()
    assert torch.equal(idx, O.topk_canonical(got, cap - w))
    kr, vr = O.gather_compact(ke, ve, idx, w)
    assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)


def test_adakv_and_headkv_long_rows(P):
    """Ada-SnapKV / HeadKV beyond one top-k workgroup's LDS (L > 57 344: the top-M lists come through the segmented top-k):
    S = 70 001, 8 query heads over 2 un-expanded KV heads; budgets, metadata and flat K/V vs the oracle's arithmetic on the
    kernel's own scores."""
    S, H, G, w, cap = 70001, 8, 4, 8, 136
    q, k, v = make_qkv(1, H, S, 128, "bf16", "gauss", 7001)
    kk, vv = k[:, ::G].contiguous(), v[:, ::G].contiguous()
    ke, ve = kk.repeat_interleave(G, dim=1), vv.repeat_interleave(G, dim=1)
    qd, kd, vd = q.to(DEV), kk.to(DEV), vv.to(DEV)
    got_s = P.ops.score_window(qd, kd, w, "maxpool", 7, "mean", kv_group=G).cpu()
    sidx, caps = O.adakv_head_capacity(got_s, cap - w, 0.2, True)
    ada = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2, normalize=True)
    kf, vf = ada.update_kv(kd, qd, vd)
    hk_caps = [[100, 1, 300, 7, 64, 128, 90, 11]]
    hk = P.HeadKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, layer_idx=0, num_hidden_layers=1,
                         head_capacity=hk_caps)
    kh, vh = hk.update_kv(kd, qd, vd)
    for cl, (kx, vx), per_head in ((ada, (kf, vf), [int(c) for c in caps[0]]), (hk, (kh, vh), hk_caps[0])):
        assert cl.head_lens.cpu().tolist() == [c + w for c in per_head]
        kx, vx, off = kx.cpu(), vx.cpu(), 0
        for h in range(H):
            n = per_head[h]
            idx = sidx[0, h, :n]
            assert torch.equal(kx[off:off + n], ke[0, h, idx]) and torch.equal(vx[off:off + n], ve[0, h, idx]), h
            assert torch.equal(kx[off + n:off + n + w], ke[0, h, -w:]) and torch.equal(vx[off + n:off + n + w], ve[0, h, -w:]), h
            off += n + w
        assert off == kx.shape[0] == cl.klen_sum


def test_adakv_capacity_beyond_one_topk_workgroup(P):
    """Ada-SnapKV with H * base > 4096 whose largest head capacity does not fit one top-k workgroup (2 L + 4 k > 160 KB of LDS):
    the budgets come from the un-sorted rows once, the order from the complete sort (no second budget pass, round 4).  Two heads
    at S = 32768, base 26000, one of them peaky: capacities, metadata and flat K/V vs the oracle's arithmetic on the kernel's
    own scores."""
    S, H, w, cap = 32768, 2, 8, 26008
    q, k, v = make_qkv(1, H, S, 128, "bf16", "gauss", 7100)
    q[:, 0] *= 3                                                      # head 0: sharp rows, few large scores
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    got_s = P.ops.score_window(qd, kd, w, "maxpool", 7, "mean").cpu()
    sidx, caps = O.adakv_head_capacity(got_s, cap - w, 0.2, False)
    per_head = [int(c) for c in caps[0]]
    assert not P.ops.topk_fits(H, S - w, max(per_head)), per_head     # the case this test is about
    ada = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=0.2, normalize=False)
    kf, vf = ada.update_kv(kd, qd, vd)
    assert ada.head_lens.cpu().tolist() == [c + w for c in per_head]
    kx, vx, off = kf.cpu(), vf.cpu(), 0
    for h in range(H):
        n = per_head[h]
        idx = sidx[0, h, :n]
        assert torch.equal(kx[off:off + n], k[0, h, idx]) and torch.equal(vx[off:off + n], v[0, h, idx]), h
        assert torch.equal(kx[off + n:off + n + w], k[0, h, -w:]) and torch.equal(vx[off + n:off + n + w], v[0, h, -w:]), h
        off += n + w
    assert off == kx.shape[0] == ada.klen_sum


def test_merge_long_odd_prompt(P):
    """LOOK-M merge on a 40 001-token prompt (several 8192-pivot chunks per kept row, odd tail), un-expanded GQA 2, two
    batches: bit-identical to the oracle's merge of the same indices."""
    B, H, G, S, w, cap = 2, 4, 2, 40001, 8, 136
    q, k, v = make_qkv(B, H, S, 128, "bf16", "gauss", 4001)
    kk, vv = k[:, ::G].contiguous(), v[:, ::G].contiguous()
    ke, ve = kk.repeat_interleave(G, dim=1), vv.repeat_interleave(G, dim=1)
    qd, kd, vd = q.to(DEV), kk.to(DEV), vv.to(DEV)
    idx = P.ops.select(qd, kd, w, cap - w, "maxpool", 7, kv_group=G)
    km, vm = P.ops.merge_compact(kd, vd, idx, w, kv_group=G)
    kr, vr = O.merge_kv(ke, ve, idx.cpu().long(), w, "pivot")
    assert torch.equal(km.cpu(), kr) and torch.equal(vm.cpu(), vr)


# ----------------------------------------------------------------------------------------- real-distribution leg (round 6)
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_sink_distribution_32k(P, dt):
    """`make_qkv(kind="sink")`: logits with std 6 and an attention-sink key every window query scores at +40.  In fp16 all but a
    few hundred pooled scores per head underflow to exactly 0, so the k-th largest value of budget 2048 (and of the lower
    PyramidKV layers) is 0 with ~30 000 ties: the selection leaves topk_kernel's small-k prefilter for its general path
    (pkv_topk.hip "heavy ties").  Bars: the scores meet the suite's score bar; the indices are EXACTLY the canonical top-k
    (value desc, index asc) of the kernels' own scores - the exact stage under mass ties; K/V are the gather of those rows;
    against the oracle's selection the token SET is identical (zeros are zeros in both implementations)."""
    B, H, S, w = 1, 8, 32768, 8
    q, k, v = make_qkv(B, H, S, 128, dt, "sink", 6600)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    got = P.ops.score_window(qd, kd, w, "maxpool", 7).cpu()
    rep = check_window_scores(q, k, w, "maxpool", 7, "sum", got, lambda: P.ops.score_window(qd, kd, w, None, 1).cpu(), frac_bar=SCORE_FRAC)
    s = O.pool_scores(O.window_scores(q, k, w), "maxpool", 7)
    order_ref = O.topk_canonical(s, 2040)
    order_own = O.topk_canonical(got, 2040)
    out = {"scores": rep, "nonzero_scores_per_head": [int(x) for x in (s != 0).sum(-1).flatten()]}
    for kk in (17, 120, 234, 2040):
        kc, vc, idx = P.ops.compress(qd, kd, vd, w, kk, "maxpool", 7, return_indices=True)
        ia = idx.cpu().long()
        assert torch.equal(ia, order_own[..., :kk]), (dt, kk, "not the canonical top-k of the kernels' own scores")
        kr, vr = O.gather_compact(k, v, ia, w)
        assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)
        seq, st = _identical(idx, order_ref[..., :kk])
        kth = torch.gather(s, -1, order_ref[..., kk - 1:kk])
        out["k%d" % kk] = dict(heads_identical_sequence=seq, heads_identical_set=st,
                               ties_at_kth_value_mean=float((s == kth).sum(-1).float().mean()))
    _report(f"sink/{dt}/S32768", out)
    for kk in (17, 120, 234, 2040):
        assert out["k%d" % kk]["heads_identical_set"] == 1.0, (dt, kk, out["k%d" % kk])


@pytest.mark.parametrize("red", ["sum", "mean"])
def test_sink_heavy_key_drags_a_whole_score_row(P, red):
    """Round 6: fp16 sink inputs, S = 8192, 32 query heads on 8 un-expanded KV heads.  In ONE window row of ONE head a key carrying
    8 % of the row's softmax mass has its product q.k 2.6e-6 below an fp16 rounding midpoint (218.4375): ATen's CPU GEMM rounds it
    up, the MFMA - correctly - down.  That one step moves the row's normaliser by 0.1 %: 265 of the head's 8184 un-pooled scores
    differ by 1-2 units and the key itself by 19, every other head agrees to 1e-4.  The score bar accepts exactly this - the oracle
    with that ONE product moved reproduces the head's whole score row (`oracle.window_scores_row_with_product_moved`) - and counts
    it as one moved product.  (Checked on the four query heads of that KV head.  Elsewhere in this tensor one score of head 3 is 2
    units off although every window row's probability there is within ONE unit of the oracle's: the fp32 row sum lands on an exact
    half - the other rows contribute fp16 subnormals, i.e. exact half-units - and ties-to-even sends the two sums apart; a property
    of this grid-valued input that the per-score bar does not state)"""
    S, Hq, Hkv, w = 8192, 32, 8, 8
    g = Hq // Hkv
    q, k8, _ = make_qkv(1, Hq, S, 128, "fp16", "sink", 4100 + S)
    k_un = k8[:, ::g].contiguous()
    k_exp = k_un[:, :, None].expand(1, Hkv, g, S, 128).reshape(1, Hq, S, 128).contiguous()
    qd, kd = q.to(DEV), k_un.to(DEV)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    hs = slice(28, 32)                                         # the query heads of KV head 7; head 29 is the one
    got = P.ops.score_window(qd, kd, w, "maxpool", 7, red, kv_group=g).cpu()[:, hs]
    rep = check_window_scores(q[:, hs], k_exp[:, hs], w, "maxpool", 7, red, got,
                               lambda: P.ops.score_window(qd, kd, w, None, 1, red, kv_group=g).cpu()[:, hs], frac_bar=5e-2, what="sink heavy key")   # max-pooling spreads every differing score over 7 positions
    _report(f"sink/heavy_key/{red}", rep)
    assert rep["heads_moved_by_one_heavy_product"] == 1 or rep["max_ulp"] <= 1, rep      # 1 on the MFMA path


# ----------------------------------------------------------------------------------------- BASELINE config 1 at the model's real shape
def llama3_8b_shaped(dev, layers=32, vocab=1024):
    """A random-init Llama with Llama-3-8B's attention / MLP dimensions (32 layers, hidden 4096, 32 / 8 heads, D = 128,
    intermediate 14336; the vocabulary is shrunk - there are no weights on disk), built directly on the device in bf16."""
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(vocab_size=vocab, hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                      num_key_value_heads=8, head_dim=128, max_position_embeddings=65536, rope_theta=500000.0)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        with torch.device(dev):
            model = LlamaForCausalLM(cfg)
    finally:
        torch.set_default_dtype(old)
    return cfg, model.eval()


def test_config1_llama3_8b_dims_all_32_layers(P):
    """BASELINE config 1 (PyramidKV budget 64, seq_len 2048) through `replace_llama("pyramidkv")` on a model with Llama-3-8B's
    real dimensions - 32 layers, 32 query / 8 KV heads - instead of the 4-layer toy of test_replace_llama_end_to_end_on_gpu
    (reference llama_model.py:157-172, :2609-2612; run_longbench.py:253-261): every layer's cache has its pyramid length
    110 ... 17 (+ window), and EVERY layer's compacted cache equals the oracle's update_kv on that layer's own captured
    K / Q / V (K/V expanded by repeat_kv as the reference hands them over)."""
    transformers = pytest.importorskip("transformers")
    from pyramidkv_amd import monkeypatch as mp
    NL, S, cap, w = 32, 2048, 64, 8
    cfg, model = llama3_8b_shaped(DEV)
    ids = torch.randint(0, cfg.vocab_size, (1, S), generator=torch.Generator().manual_seed(1)).to(DEV)
    captured = {}
    orig_update = P.PyramidKVCluster.update_kv

    def recording_update(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        if self.layer_idx not in captured:
            captured[self.layer_idx] = tuple(t.detach().cpu() for t in (key_states, query_states, value_states))
        return orig_update(self, key_states, query_states, value_states, attention_mask, num_key_value_groups)

    try:
        mp.replace_llama("pyramidkv")
        P.PyramidKVCluster.update_kv = recording_update
        for layer in model.model.layers:
            c = layer.self_attn.config
            c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = w, cap, 7, "maxpool", None
        with torch.no_grad():
            out = model(ids, past_key_values=transformers.DynamicCache(config=cfg), use_cache=True, logits_to_keep=1)
        torch.cuda.synchronize()
        cache = out.past_key_values
        budgets = [O.pyramid_budget(cap, w, NL, i, S) for i in range(NL)]
        assert [b for b, _ in budgets] == ["pyramid"] * NL and budgets[0][1] == 110 and budgets[-1][1] == 17
        assert [cache.layers[i].keys.shape[2] for i in range(NL)] == [kk + w for _, kk in budgets]
        assert sorted(captured) == list(range(NL))
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        seq_same = 0
        for li in range(NL):
            kx, qx, vx = captured[li]
            assert kx.shape == (1, 8, S, 128) and qx.shape == (1, 32, S, 128)           # handed over before repeat_kv
            kx, vx = (t.repeat_interleave(4, dim=1) for t in (kx, vx))
            kr, vr = O.pyramidkv_update_kv(kx, qx, vx, w, cap, 7, "maxpool", NL, li)
            lay = cache.layers[li]
            assert lay.keys.shape == kr.shape, li
            same = torch.equal(lay.keys.cpu(), kr) and torch.equal(lay.values.cpu(), vr)
            seq_same += int(same)
            if not same:      # the same token SET per head (two rows one score-ulp apart may swap: tests/score_bar.py)
                ks_a = torch.sort(lay.keys.cpu().float().flatten(2), dim=2).values
                ks_b = torch.sort(kr.float().flatten(2), dim=2).values
                assert torch.equal(ks_a, ks_b), (li, "another token set")
        _report("config1/llama3_8b_dims/S2048cap64", dict(layers=NL, layers_bit_identical=seq_same, cache_lens=[kk + w for _, kk in budgets]))
        assert seq_same == NL, seq_same
        assert cache.get_seq_length() == S
    finally:
        P.PyramidKVCluster.update_kv = orig_update
        mp.restore()
        del model
        torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------- three-way same-chip parity (round 6)
# libpkv vs the reference's op sequence on the host CPU vs the SAME op sequence on PyTorch-ROCm eager on this GPU
# (tests/three_way.py; SURVEY.md section 7 hard part 1 / section 8c).  tools/parity_three_way.py runs the full grid and
# writes profiles/r06/parity_three_way.json; these tests assert its statements on BASELINE configurations 2, 3 and 5.
import three_way as T3  # noqa: E402


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_three_way_config2_pyramidkv_8k(P, dt):
    """All 32 PyramidKV layer budgets (234 ... 17) at S = 8192: libpkv selects the token SET of the CPU reference AND of the
    device reference in every head of every layer (with scale "rcp" + tie order "aten_rocm" against the latter), and its
    index SEQUENCE agrees with each reference at least as often as the two references agree with each other."""
    q, k, v = make_qkv(1, 32, 8192, 128, dt, "gauss", 6200)
    budgets = {"layer%02d" % layer: O.pyramid_budget(128, 8, 32, layer, 8192)[1] for layer in range(32)}
    rep = T3.window_policy(P, q, k, v, 8, budgets, dev=DEV)
    tot = {p: [0, 0, 0] for p in ("hip_vs_cpu", "hip_vs_eager", "eager_vs_cpu")}
    for label, b in rep["budgets"].items():
        for p in tot:
            tot[p][0] += b[p]["identical_set"]
            tot[p][1] += b[p]["identical_sequence"]
            tot[p][2] += b[p]["heads"]
        assert b["hip_vs_cpu"]["set_rate"] == 1.0 and b["hip_vs_eager"]["set_rate"] == 1.0, (label, b)
    _report(f"three_way/config2/{dt}", dict(totals=tot, score_ulp=rep["scores"]))
    assert tot["hip_vs_cpu"][1] >= tot["eager_vs_cpu"][1], tot
    assert tot["hip_vs_eager"][1] >= tot["eager_vs_cpu"][1], tot


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_three_way_config3_snapkv_32k(P, dt):
    """SnapKV budgets 128 and 2048 at S = 32768.  Budget 128: identical token set in every head for every pair that involves
    libpkv.  Budget 2048, where the residue of the path lives (one q.k product rounding across a model-dtype midpoint,
    tests/score_bar.py): libpkv disagrees with the CPU reference in NO MORE heads than the device reference does - the same
    measurement on the same chip (profiles/r06/parity_three_way.json: bf16 0 vs 0, fp16 1 vs 21 of 32 heads)."""
    q, k, v = make_qkv(1, 32, 32768, 128, dt, "gauss", 6300)
    rep = T3.window_policy(P, q, k, v, 8, {"budget128": 120, "budget2048": 2040}, dev=DEV)
    b128, b2048 = rep["budgets"]["budget128"], rep["budgets"]["budget2048"]
    _report(f"three_way/config3/{dt}", dict(summary=T3.summarise(rep)))
    for pair in ("hip_vs_cpu", "hip_vs_eager"):
        assert b128[pair]["set_rate"] == 1.0, (pair, b128[pair])
        assert b2048[pair]["set_rate"] == 1.0, (pair, b2048[pair])
    dis = {p: b2048[p]["heads"] - b2048[p]["identical_sequence"] for p in ("hip_vs_cpu", "hip_vs_eager", "eager_vs_cpu")}
    assert dis["hip_vs_cpu"] <= dis["eager_vs_cpu"], dis
    # every K/V row libpkv returns is the row the reference returns wherever the sequences agree (exact gather)
    for b in (b128, b2048):
        for pair in ("hip_vs_cpu", "hip_vs_eager"):
            assert b[pair]["kv_bit_identical"] >= b[pair]["identical_sequence"], (pair, b[pair])


@pytest.mark.parametrize("cap", [128, 2048])
def test_three_way_config5_adakv_gqa_32k(P, cap):
    """Ada-SnapKV on Mistral's GQA layout (32 / 8 heads, K/V un-expanded), S = 32768: libpkv's head budgets equal the CPU
    reference's; against the device reference they differ in no more heads than the two references differ from each other."""
    q, k, v = make_qkv(1, 32, 32768, 128, "bf16", "gauss", 6500)
    rep = T3.adakv(P, q, k[:, ::4].contiguous(), v[:, ::4].contiguous(), 8, cap, dev=DEV)
    _report(f"three_way/config5/budget{cap}", rep)
    assert rep["hip_vs_cpu"]["head_budgets_identical"], rep["hip_vs_cpu"]
    assert rep["hip_vs_eager"]["heads_with_other_budget"] <= max(rep["eager_vs_cpu"]["heads_with_other_budget"], 0) + 0, rep
    assert rep["hip_vs_cpu"]["kv_rate"] >= rep["eager_vs_cpu"]["kv_rate"], rep
