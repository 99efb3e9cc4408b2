"""Boundary sweep: every policy on SMALL prompts around every tiling boundary of the kernels (64 / 128 / 256-key tiles,
1008-position finalize blocks, window larger than the past, budget == everything, one selected token ...), against the
oracle.  The large-shape suites prove the BASELINE configurations; this one looks for what they cannot see (it found NaN
H2O scores for S < w + 12).

Per case: (1) scores within one unit of the last place of the oracle's on at most a few elements, (2) the selected indices
are the canonical top-k of the kernel's OWN scores - which ties the selection to the scores exactly, whatever one-unit
noise they carry - and identical to the oracle's when the scores are, (3) the compacted K/V are exact copies of the rows the
indices name, in order, followed by the window.
"""

import numpy as np
import pytest
import torch

from inputs import make_qkv
from oracle import pkv_oracle as O
from test_gpu_parity import DEV, score_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pyramidkv_amd
    return pyramidkv_amd


SHAPES = [  # (S, w)
    (9, 8), (10, 1), (17, 16), (33, 32), (65, 64), (66, 8), (72, 64), (100, 32), (127, 8), (128, 8), (129, 8), (130, 64),
    (255, 16), (256, 8), (257, 8), (300, 64), (511, 8), (513, 32), (1007, 8), (1016, 8), (1017, 8), (1025, 8), (2023, 16),
]


def _caps(S, w):
    L = S - w
    ks = sorted({1, 2, min(L, 7), min(L, 17), max(1, L // 2), max(1, L - 1), L})
    return [k for k in ks if 1 <= k <= L]


@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("pool,ks", [("maxpool", 7), ("avgpool", 5), (None, 1)])
def test_window_policies_small_shapes(P, dt, pool, ks):
    n_cases = mism = elems = 0
    for (S, w) in SHAPES:
        G = 2 if S % 2 else 1
        # lattice inputs: every q.k is exact in fp32 in any order, so the logits carry no accumulation-order noise (on
        # Gaussian inputs a logit that lands on a rounding boundary moves a LARGE probability of a short row by 2-3 units)
        q, k, v = make_qkv(2, 4, S, 128, dt, "lattice", 31 * S + w)
        kk, vv = k[:, ::G].contiguous(), v[:, ::G].contiguous()          # un-expanded K/V for G > 1
        ke, ve = kk.repeat_interleave(G, dim=1), vv.repeat_interleave(G, dim=1)
        qd, kd, vd = q.to(DEV), kk.to(DEV), vv.to(DEV)
        want_s = O.pool_scores(O.window_scores(q, ke, w), pool, ks)
        got_s = P.ops.score_window(qd, kd, w, pool, ks, kv_group=G).cpu()
        if dt == "fp32":                                         # no rounding grid: relative error of exp() and the sum orders
            err = ((got_s - want_s).abs() / want_s.abs().clamp_min(1e-30)).max().item()
            assert err <= 2e-6, (S, w, G, err)
            frac, mx = float((got_s != want_s).float().mean()), 0
        else:
            frac, mx = score_diff(got_s, want_s)
        assert mx <= 1 and (dt == "fp32" or frac <= max(5e-3, 8.0 / got_s.numel())), (S, w, G, frac, mx)     # max pooling repeats a difference 7 times
        mism += round(frac * got_s.numel())
        elems += got_s.numel()
        for kk_sel in _caps(S, w):
            kc, vc, idx = P.ops.compress(qd, kd, vd, w, kk_sel, pool, ks, kv_group=G, return_indices=True)
            idx = idx.cpu().long()
            assert torch.equal(idx, O.topk_canonical(got_s, kk_sel)), (S, w, G, kk_sel)
            if frac == 0.0:
                assert torch.equal(idx, O.topk_canonical(want_s, kk_sel))
            kr, vr = O.gather_compact(ke, ve, idx, w)
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), (S, w, G, kk_sel)
            n_cases += 1
    assert n_cases > 100 and (dt == "fp32" or mism / elems <= 2e-3)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_pyramidkv_small_prompts_all_layers(P, dt):
    """PyramidKV's three branches on prompts around the branch thresholds (q_len < cap: pass-through, q_len < (cap - w) * 2:
    uniform budget, else the pyramid), every layer of a 32- and a 2-layer model."""
    for (S, w, cap) in ((60, 8, 64), (64, 8, 64), (100, 8, 64), (111, 8, 64), (112, 8, 64), (113, 8, 64), (300, 16, 48), (40, 32, 33)):
        q, k, v = make_qkv(1, 2, S, 128, dt, "gauss", S + cap)
        qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
        for layers in (32, 2):
            for layer in range(layers):
                cl = P.PyramidKVCluster(num_hidden_layers=layers, layer_idx=layer, window_size=w, max_capacity_prompt=cap, kernel_size=5,
                                        pooling="avgpool")
                branch, kk_sel = O.pyramid_budget(cap, w, layers, layer, S, 20)
                assert cl.layer_budget(S) == (branch, kk_sel)
                kc, vc = cl.update_kv(kd, qd, vd, None, 1)
                if branch == "passthrough":
                    assert kc is kd and vc is vd
                    continue
                if kk_sel == 0:                                  # topk(0): the window alone (:271-272)
                    assert torch.equal(kc.cpu(), k[:, :, -w:]) and torch.equal(vc.cpu(), v[:, :, -w:])
                    continue
                got_s = P.ops.score_window(qd, kd, w, "avgpool", 5).cpu()
                idx = O.topk_canonical(got_s, kk_sel)
                kr, vr = O.gather_compact(k, v, idx, w)
                assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), (S, w, cap, layers, layer)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_h2o_and_streaming_small_shapes(P, dt):
    for (S, w) in SHAPES:
        q, k, v = make_qkv(1, 2, S, 128, dt, "lattice", 17 * S + w)
        qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
        got_s = P.ops.score_h2o(qd, kd, w).cpu()
        want_s = O.h2o_scores(q, k, w)
        frac, mx = score_diff(got_s, want_s)
        assert mx <= 1 and frac <= max(1e-3, 8.0 / got_s.numel()), (S, w, frac, mx)
        for kk_sel in _caps(S, w):
            cap = kk_sel + w
            kc, vc = P.H2OKVCluster(window_size=w, max_capacity_prompt=cap).update_kv(kd, qd, vd, None, 1)
            idx = O.topk_canonical(got_s, kk_sel)
            kr, vr = O.gather_compact(k, v, idx, w)
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), (S, w, kk_sel)
            kc, vc = P.StreamingLLMKVCluster(window_size=w, max_capacity_prompt=cap).update_kv(kd, qd, vd, None, 1)
            kr, vr = O.streamingllm_update_kv(k, q, v, w, cap)
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), (S, w, kk_sel)


def test_adakv_small_shapes(P):
    """Ada-SnapKV on small prompts: metadata identical to the oracle's evaluated on the kernel's own scores, flat K/V exact."""
    for (S, w, cap, H) in ((40, 8, 16, 4), (65, 8, 32, 8), (129, 16, 40, 4), (300, 8, 64, 8), (513, 32, 96, 2), (1030, 8, 128, 8)):
        for floor in (0.0, 0.2, 1.0):
            q, k, v = make_qkv(1, H, S, 128, "bf16", "gauss", S + H)
            qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
            cl = P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=floor, normalize=True)
            kf, vf = cl.update_kv(kd, qd, vd)
            got_s = P.ops.score_window(qd, kd, w, "maxpool", 7, "mean").cpu()
            _, caps = O.adakv_head_capacity(got_s, cap - w, floor, True)
            lens = [int(c) + w for c in caps[0]]
            assert cl.head_lens.cpu().tolist() == lens, (S, w, cap, H, floor)
            off = 0
            kfc, vfc = kf.cpu(), vf.cpu()
            for h in range(H):
                n = lens[h] - w
                idx = O.topk_canonical(got_s[0, h][None], n)[0] if n > 0 else torch.zeros(0, dtype=torch.int64)
                assert torch.equal(kfc[off:off + n], k[0, h, idx]) and torch.equal(vfc[off:off + n], v[0, h, idx]), (S, h, floor)
                assert torch.equal(kfc[off + n:off + n + w], k[0, h, -w:]) and torch.equal(vfc[off + n:off + n + w], v[0, h, -w:])
                off += lens[h]
            assert off == kf.shape[0] == cl.klen_sum


def test_merge_small_shapes(P):
    """LOOK-M pivot merge on small prompts: bit-identical to the oracle's merge of the same indices."""
    for (S, w, cap, H, G) in ((20, 8, 12, 2, 1), (65, 8, 24, 4, 2), (129, 16, 40, 4, 1), (300, 8, 64, 8, 4), (513, 32, 96, 2, 1)):
        q, k, v = make_qkv(2, H, S, 128, "bf16", "gauss", 3 * S + H)
        kk, vv = k[:, ::G].contiguous(), v[:, ::G].contiguous()
        ke, ve = kk.repeat_interleave(G, dim=1), vv.repeat_interleave(G, dim=1)
        qd, kd, vd = q.to(DEV), kk.to(DEV), vv.to(DEV)
        idx = P.ops.select(qd, kd, w, cap - w, "maxpool", 7, kv_group=G)
        km, vm = P.ops.merge_compact(kd, vd, idx, w, kv_group=G)
        kr, vr = O.merge_kv(ke, ve, idx.cpu().long(), w, "pivot")
        assert torch.equal(km.cpu(), kr) and torch.equal(vm.cpu(), vr), (S, w, cap, H, G)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("D", [64, 256])
def test_window_policies_small_shapes_head_sizes(P, dt, D):
    """The same sweep for head sizes 64 and 256 (the general K-scan kernel: MFMA k-steps = D / 32)."""
    n = 0
    for (S, w) in SHAPES[::2]:
        G = 2 if S % 2 else 1
        q, k, v = make_qkv(1, 4, S, D, dt, "lattice", 7 * S + w + D)
        kk, vv = k[:, ::G].contiguous(), v[:, ::G].contiguous()
        ke, ve = kk.repeat_interleave(G, dim=1), vv.repeat_interleave(G, dim=1)
        qd, kd, vd = q.to(DEV), kk.to(DEV), vv.to(DEV)
        want_s = O.pool_scores(O.window_scores(q, ke, w), "maxpool", 7)
        got_s = P.ops.score_window(qd, kd, w, "maxpool", 7, kv_group=G).cpu()
        frac, mx = score_diff(got_s, want_s)
        assert mx <= 1 and frac <= max(5e-3, 8.0 / got_s.numel()), (S, w, G, D, frac, mx)
        for kk_sel in _caps(S, w)[::2]:
            kc, vc, idx = P.ops.compress(qd, kd, vd, w, kk_sel, "maxpool", 7, kv_group=G, return_indices=True)
            idx = idx.cpu().long()
            assert torch.equal(idx, O.topk_canonical(got_s, kk_sel)), (S, w, G, D, kk_sel)
            kr, vr = O.gather_compact(ke, ve, idx, w)
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), (S, w, G, D, kk_sel)
            n += 1
        kc, vc = P.StreamingLLMKVCluster(window_size=w, max_capacity_prompt=w + 1).update_kv(kd, qd, vd, None, G)
        kr, vr = O.streamingllm_update_kv(ke, q, ve, w, w + 1)
        assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), (S, w, G, D)
    assert n > 20


def test_random_medium_shapes_window_and_h2o(P):
    """40 seeded random configurations (S 300..6000, windows 1..64, GQA 1/2/4, odd head counts, B 1..3, every pooling and
    kernel size, both 16-bit dtypes, budgets anywhere in [1, L]): window score + selection + gather and H2O, as above."""
    rng = np.random.RandomState(20260924)
    for case in range(40):
        S = int(rng.randint(300, 6001))
        w = int(rng.choice([1, 8, 16, 32, 64]))
        G = int(rng.choice([1, 2, 4]))
        H = G * int(rng.randint(1, 4))
        B = int(rng.randint(1, 4))
        dt = ("bf16", "fp16")[case % 2]
        pool, ks = [("maxpool", 7), ("avgpool", 5), ("maxpool", 17), ("avgpool", 13), (None, 1), ("maxpool", 3)][int(rng.randint(0, 6))]
        L = S - w
        kk_sel = int(rng.randint(1, L + 1)) if case % 4 else int(rng.choice([1, L, min(L, 4096), min(L, 2040)]))
        q, k, v = make_qkv(B, H, S, 128, dt, "lattice", 100 + case)
        kk, vv = k[:, ::G].contiguous(), v[:, ::G].contiguous()
        ke, ve = kk.repeat_interleave(G, dim=1), vv.repeat_interleave(G, dim=1)
        qd, kd, vd = q.to(DEV), kk.to(DEV), vv.to(DEV)
        tag = (case, B, H, G, S, w, dt, pool, ks, kk_sel)
        want_s = O.pool_scores(O.window_scores(q, ke, w), pool, ks)
        got_s = P.ops.score_window(qd, kd, w, pool, ks, kv_group=G).cpu()
        frac, mx = score_diff(got_s, want_s)
        assert mx <= 1 and frac <= 5e-3, tag + (frac, mx)
        kc, vc, idx = P.ops.compress(qd, kd, vd, w, kk_sel, pool, ks, kv_group=G, return_indices=True)
        idx = idx.cpu().long()
        assert torch.equal(idx, O.topk_canonical(got_s, kk_sel)), tag
        kr, vr = O.gather_compact(ke, ve, idx, w)
        assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), tag
        if case % 3 == 0 and S <= 3000:                           # H2O: S x S on the CPU oracle
            want_h = O.h2o_scores(q, ke, w)
            got_h = P.ops.score_h2o(qd, kd, w, kv_group=G).cpu()
            frac, mx = score_diff(got_h, want_h)
            assert mx <= 1 and frac <= max(1e-3, 8.0 / got_h.numel()), tag + (frac, mx)
            kc, vc, idx = P.ops.compress(qd, kd, vd, w, kk_sel, None, 1, kv_group=G, h2o=True, return_indices=True)
            assert torch.equal(idx.cpu().long(), O.topk_canonical(got_h, kk_sel)), tag
            kr, vr = O.gather_compact(ke, ve, idx.cpu().long(), w)
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr), tag


def test_wide_gqa_group_is_split(P):
    """8 query heads per KV head next to the reference's default window of 64 (Llama-3-70B shapes) is 512 columns per key
    row, more than the K scan carries (256): the host splits the group (K/V expanded x2, kv_group 4) instead of refusing."""
    B, H, G, S, w, cap = 1, 16, 8, 700, 64, 96
    q, k, v = make_qkv(B, H, S, 128, "bf16", "lattice", 99)
    kk, vv = k[:, ::G].contiguous(), v[:, ::G].contiguous()
    ke, ve = kk.repeat_interleave(G, dim=1), vv.repeat_interleave(G, dim=1)
    qd, kd, vd = q.to(DEV), kk.to(DEV), vv.to(DEV)
    got_s = P.ops.score_window(qd, ke.to(DEV), w, "avgpool", 5).cpu()
    idx = O.topk_canonical(got_s, cap - w)
    kr, vr = O.gather_compact(ke, ve, idx, w)
    for cl in (P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=5, pooling="avgpool"),
               P.PyramidKVCluster(num_hidden_layers=2, layer_idx=0, window_size=w, max_capacity_prompt=cap, kernel_size=5, pooling="avgpool")):
        kc, vc = cl.update_kv(kd, qd, vd, None, G)
        if isinstance(cl, P.SnapKVCluster):
            assert torch.equal(kc.cpu(), kr) and torch.equal(vc.cpu(), vr)
        else:
            assert kc.shape[1] == H and kc.shape[2] == cl.layer_budget(S)[1] + w
    km, vm = P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=5, pooling="avgpool", merge="pivot").update_kv(kd, qd, vd, None, G)
    kmr, vmr = O.merge_kv(ke, ve, idx, w, "pivot")
    assert torch.equal(km.cpu(), kmr) and torch.equal(vm.cpu(), vmr)
    ada = P.AdaKVCluster(window_size=w, kernel_size=5, pooling="avgpool", max_capacity_prompt=cap, floor=0.2, normalize=True)
    kf, vf = ada.update_kv(kd, qd, vd)
    ref = P.AdaKVCluster(window_size=w, kernel_size=5, pooling="avgpool", max_capacity_prompt=cap, floor=0.2, normalize=True)
    kf2, vf2 = ref.update_kv(ke.to(DEV), qd, ve.to(DEV))
    assert torch.equal(kf, kf2) and torch.equal(vf, vf2) and ada.head_lens.tolist() == ref.head_lens.tolist()


def test_no_writes_outside_the_caller_buffers(P):
    """Every output the C ABI writes lives between two canary regions inside one big buffer (outputs sized exactly as
    include/pkv.h states, odd lengths and budgets): after the call the canaries are intact, i.e. no kernel stores a byte
    outside [out, out + size) - torch's allocator rounds allocations up and would hide such a store."""
    N, ops = P._native, P.ops
    CAN = 4096

    def guarded(nbytes):
        buf = torch.full((CAN + ((nbytes + 255) // 256) * 256 + CAN,), 0xA5, dtype=torch.uint8, device=DEV)
        return buf, buf[CAN:CAN + nbytes]

    def intact(buf, nbytes):
        return bool((buf[:CAN] == 0xA5).all()) and bool((buf[CAN + nbytes:] == 0xA5).all())

    for (dt, es) in ((torch.bfloat16, 2), (torch.float16, 2), (torch.float32, 4)):
        for (B, H, G, S, w, k_sel, pool, ks) in ((1, 4, 2, 1003, 8, 37, "maxpool", 7), (2, 2, 1, 77, 16, 61, "avgpool", 5), (1, 2, 1, 4099, 64, 1, None, 1),
                                                (1, 8, 4, 300, 32, 268, "maxpool", 17)):
            q = torch.randn(B, H, S, 128, device=DEV).to(dt)
            k = torch.randn(B, H // G, S, 128, device=DEV).to(dt)
            v = torch.randn(B, H // G, S, 128, device=DEV).to(dt)
            L = S - w
            Lp = (L + 7) // 8 * 8
            with torch.cuda.device(q.device):
                d = ops.make_desc(q, k, v, w, pool, ks, "sum", "div", k_sel, G)
                ws = torch.empty(N.lib.pkv_workspace_bytes(d), dtype=torch.uint8, device=DEV)
                # pkv_compress: k_out, v_out [B,H,k+w,D], idx_out [B,H,k]
                nb = B * H * (k_sel + w) * 128 * es
                kb, ko = guarded(nb)
                vb, vo = guarded(nb)
                ib, io = guarded(B * H * k_sel * 4)
                N.check(N.lib.pkv_compress(d, q.data_ptr(), k.data_ptr(), v.data_ptr(), ko.data_ptr(), vo.data_ptr(), io.data_ptr(),
                                           ws.data_ptr(), ws.numel(), N.stream_ptr()), "pkv_compress")
                torch.cuda.synchronize()
                assert intact(kb, nb) and intact(vb, nb) and intact(ib, B * H * k_sel * 4), ("compress", dt, S, w, k_sel)
                idx = io.view(torch.int32).view(B, H, k_sel)
                assert int(idx.min()) >= 0 and int(idx.max()) < L
                # pkv_score_window: [B*H][stride] with stride = Lp: columns [0, L) written, nothing past the last row
                sb, so = guarded(B * H * Lp * es)
                N.check(N.lib.pkv_score_window(d, q.data_ptr(), k.data_ptr(), so.data_ptr(), Lp, ws.data_ptr(), ws.numel(), N.stream_ptr()),
                        "pkv_score_window")
                torch.cuda.synchronize()
                assert intact(sb, B * H * Lp * es), ("score_window", dt, S, w)
                # pkv_topk on those scores: [rows][k]
                tb, to = guarded(B * H * k_sel * 4)
                N.check(N.lib.pkv_topk(N.dtype_code(dt), B * H, L, k_sel, so.data_ptr(), Lp, None, to.data_ptr(), k_sel, N.stream_ptr()), "pkv_topk")
                torch.cuda.synchronize()
                assert intact(tb, B * H * k_sel * 4), ("topk", dt, S, k_sel)
                assert torch.equal(to.view(torch.int32).view(B, H, k_sel), idx)
                if dt != torch.float32:                            # H2O
                    kb, ko = guarded(nb)
                    vb, vo = guarded(nb)
                    N.check(N.lib.pkv_compress_h2o(d, q.data_ptr(), k.data_ptr(), v.data_ptr(), ko.data_ptr(), vo.data_ptr(), None,
                                                   ws.data_ptr(), ws.numel(), N.stream_ptr()), "pkv_compress_h2o")
                    torch.cuda.synchronize()
                    assert intact(kb, nb) and intact(vb, nb), ("h2o", dt, S, w, k_sel)


def test_no_writes_outside_any_buffer_the_host_layer_allocates(P):
    """The same canary idea for everything pyramidkv_amd.ops allocates (outputs, index lists, metadata, the workspace):
    torch.empty / empty_like inside ops are routed through guarded allocations for the duration of the test, every policy
    runs on odd shapes, then all canaries are checked.  Covers the flat (Ada-SnapKV / HeadKV) paths with their bound-sized
    buffers, the merge, StreamingLLM and the decode-time append."""
    import types
    CAN = 4096
    live = []

    class Proxy(types.ModuleType):
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def _guard(shape, dtype, device):
            n = 1
            for s_ in shape:
                n *= int(s_)
            nbytes = n * torch.empty(0, dtype=dtype).element_size()
            buf = torch.full((CAN + ((nbytes + 255) // 256) * 256 + CAN,), 0xA5, dtype=torch.uint8, device=device)
            live.append((buf, nbytes))
            return buf[CAN:CAN + nbytes].view(dtype).view(*shape)

        def empty(self, *size, dtype=torch.float32, device=None, pin_memory=False, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            if device is None or torch.device(device).type != "cuda":
                return torch.empty(*size, dtype=dtype, device=device, pin_memory=pin_memory, **kw)
            return self._guard(size, dtype, device)

        def empty_like(self, t, **kw):
            return self._guard(tuple(t.shape), t.dtype, t.device) if t.is_cuda and not kw else torch.empty_like(t, **kw)

    ops = P.ops
    saved_torch, saved_ws = ops.torch, dict(ops._WS) if hasattr(ops, "_WS") else None
    ops.torch = Proxy("torch_proxy")
    if hasattr(ops, "_WS"):
        ops._WS.clear()                                            # the workspace is re-allocated (guarded) on first use
    try:
        for dt in (torch.bfloat16, torch.float16):
            # the last shape has H * (cap - w) > 4096: Ada-SnapKV takes its budgets from the un-sorted rows there
            for (H, G, S, w, cap) in ((4, 2, 1003, 8, 45), (8, 4, 333, 32, 40), (2, 1, 4099, 16, 531), (8, 4, 2999, 8, 600)):
                q = torch.randn(1, H, S, 128, device=DEV).to(dt)
                k = torch.randn(1, H // G, S, 128, device=DEV).to(dt)
                v = torch.randn(1, H // G, S, 128, device=DEV).to(dt)
                P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=7, pooling="maxpool").update_kv(k, q, v, None, G)
                P.SnapKVCluster(window_size=w, max_capacity_prompt=cap, kernel_size=5, pooling="avgpool", merge="pivot").update_kv(k, q, v, None, G)
                P.PyramidKVCluster(num_hidden_layers=8, layer_idx=3, window_size=w, max_capacity_prompt=cap, kernel_size=7,
                                   pooling="maxpool").update_kv(k, q, v, None, G)
                P.H2OKVCluster(window_size=w, max_capacity_prompt=cap).update_kv(k, q, v, None, G)
                P.StreamingLLMKVCluster(window_size=w, max_capacity_prompt=cap).update_kv(k, q, v, None, G)
                for floor in (0.0, 0.2):
                    P.AdaKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, floor=floor,
                                   normalize=True).update_kv(k, q, v)
                caps = [[(7 * h + 3) % (cap - w) + 1 for h in range(H)]]
                hk = P.HeadKVCluster(window_size=w, kernel_size=7, pooling="maxpool", max_capacity_prompt=cap, layer_idx=0,
                                     num_hidden_layers=1, head_capacity=caps)
                kf, vf = hk.update_kv(k, q, v)
                state = torch.randn(H, 128, device=DEV).to(dt)
                ops.update_flatten_view(kf, state, hk.head_lens, hk.cu_klen)
                sc = ops.score_window(q, k, w, "maxpool", 7, kv_group=G)
                ops.topk(sc, 17)
                ops.sort_rows(sc[0])
        torch.cuda.synchronize()
        assert len(live) > 100
        for buf, nbytes in live:
            assert bool((buf[:CAN] == 0xA5).all()) and bool((buf[CAN + nbytes:] == 0xA5).all()), (nbytes, buf.numel())
    finally:
        ops.torch = saved_torch
        if saved_ws is not None:
            ops._WS.clear()
            ops._WS.update(saved_ws)
