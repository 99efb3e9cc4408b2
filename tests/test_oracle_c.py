"""Pin oracle/pkv_oracle.c (plain-C restatement of the integer/byte stages) against the torch
restatement oracle/pkv_oracle.py (itself pinned bit-exact to the real reference by tests/golden)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from inputs import make_qkv, bits, from_bits
from oracle import pkv_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "libpkv_oracle.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(SO):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    L = C.CDLL(SO)
    L.pkv_o_to_f32.restype = C.c_float
    L.pkv_o_to_f32.argtypes = [C.c_uint16, C.c_int]
    L.pkv_o_from_f32.restype = C.c_uint16
    L.pkv_o_from_f32.argtypes = [C.c_float, C.c_int]
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("dt,tdt", [(0, torch.bfloat16), (1, torch.float16)])
def test_conversions_match_torch(lib, dt, tdt):
    allbits = np.arange(65536, dtype=np.uint16)
    t = from_bits(allbits, tdt).float().numpy()
    mine = np.array([lib.pkv_o_to_f32(int(b), dt) for b in allbits[::7]], dtype=np.float32)
    ref = t[::7]
    assert np.array_equal(mine.view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)])
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4000, generator=g) * 10 ** torch.randint(-9, 6, (4000,), generator=g).float(),
                   torch.tensor([0.0, -0.0, 65504.0, 65519.9, 65520.0, 70000.0, 5.96e-8, 2.98e-8, 2.9802322e-8, 3e-8,
                                 6.1e-5, 6.0975552e-5, 1e-45, 3.3895314e38, 3.4e38, float("inf"), -float("inf")])])
    want = bits(x.to(tdt))
    got = np.array([lib.pkv_o_from_f32(float(v), dt) for v in x.numpy()], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_budget_grid(lib):
    br, k = C.c_int(), C.c_int()
    names = {0: "passthrough", 1: "snap", 2: "pyramid"}
    for cap in (64, 96, 128, 2048):
        for w in (8, 32):
            for S in (cap - 1, cap, 2 * (cap - w) - 1, 2 * (cap - w), 2 * (cap - w) + 5, 8192, 32768):
                if S <= w:
                    continue
                for layer in (0, 3, 31):
                    lib.pkv_o_pyramid_budget(cap, w, 32, layer, S, 20, C.byref(br), C.byref(k))
                    assert (names[br.value], k.value) == O.pyramid_budget(cap, w, 32, layer, S)


@pytest.mark.parametrize("dt,name", [(0, "bf16"), (1, "fp16")])
def test_pool_topk_gather_vs_torch_oracle(lib, dt, name):
    q, k, v = make_qkv(1, 2, 1024, 128, name, "gauss", 5)
    w = 8
    s = O.window_scores(q, k, w)
    L = s.shape[-1]
    for kind, pool, ks in ((2, "maxpool", 7), (1, "avgpool", 5), (2, "maxpool", 3), (1, "avgpool", 9)):
        want = O.pool_scores(s, pool, ks)
        for h in range(2):
            a = bits(s[0, h]).copy()
            out = np.zeros(L, dtype=np.uint16)
            lib.pkv_o_pool(ptr(a), dt, L, kind, ks, ptr(out))
            assert np.array_equal(out, bits(want[0, h])), (pool, ks)
    sp = O.pool_scores(s, "maxpool", 7)
    for kk in (1, 17, 120, L - 1, L):
        want = O.topk_canonical(sp, kk)
        for h in range(2):
            a = bits(sp[0, h]).copy()
            idx = np.zeros(kk, dtype=np.int32)
            assert lib.pkv_o_topk(ptr(a), dt, L, kk, ptr(idx)) == 0
            assert np.array_equal(idx, want[0, h].numpy().astype(np.int32))
    idx = O.topk_canonical(sp, 120)
    kc, vc = O.gather_compact(k, v, idx, w)
    for h in range(2):
        src = bits(k[0, h]).copy()
        out = np.zeros((120 + w, 128), dtype=np.uint16)
        ii = idx[0, h].numpy().astype(np.int32)
        lib.pkv_o_gather(ptr(src), C.c_int64(256), 256, 1024, w, ptr(ii), 120, ptr(out))
        assert np.array_equal(out, bits(kc[0, h]))


@pytest.mark.parametrize("floor,norm", [(0.2, True), (0.5, False), (0.0, True), (0.35, True)])
def test_ada_capacity_vs_torch_oracle(lib, floor, norm):
    H, w, base = 6, 8, 56
    q, k, v = make_qkv(1, H, 768, 128, "bf16", "gauss", 9)
    s = O.pool_scores(O.window_scores(q, k, w, "mean"), "maxpool", 7)
    sidx, cap = O.adakv_head_capacity(s, base, floor, norm, "canonical")
    sv = torch.gather(s, -1, sidx)
    a = bits(sv[0]).copy()
    out = np.zeros(H, dtype=np.int32)
    assert lib.pkv_o_ada_capacity(ptr(a), 0, H, s.shape[-1], base, C.c_double(floor), int(norm), ptr(out)) == 0
    assert out.tolist() == cap[0].tolist()
    hl, cu = np.zeros(H, np.int32), np.zeros(H + 1, np.int32)
    lib.pkv_o_ada_metadata(H, w, ptr(out), ptr(hl), ptr(cu))
    meta = O._ada_meta(H, [int(c) + w for c in cap[0]], int(cap.sum()) + H * w, 0, "cpu")
    assert hl.tolist() == meta.head_lens.tolist() and cu.tolist() == meta.cu_klen.tolist()


def test_flat_append_vs_torch_oracle(lib):
    H, D = 5, 128
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(1, 50, (H,), generator=g, dtype=torch.int32)
    cu = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(lens, 0, dtype=torch.int32)])
    cache = torch.randn(int(cu[-1]), D, generator=g).to(torch.float16)
    state = torch.randn(H, D, generator=g).to(torch.float16)
    want = O.update_flatten_view(cache, state, lens, cu)
    out = np.zeros((cache.shape[0] + H, D), dtype=np.uint16)
    lib.pkv_o_update_flatten_view(ptr(bits(cache).copy()), ptr(bits(state).copy()), ptr(lens.numpy().copy()),
                                  ptr(cu.numpy().copy()), H, 256, ptr(out))
    assert np.array_equal(out, bits(want))


def test_c_oracle_against_the_real_reference_fixtures(lib):
    """The C restatement against outputs of the REAL reference directly (tests/golden, not via the torch restatement):
    on the tie-free fixtures pool -> top-k -> gather reproduce the reference's indices and K/V bit for bit; on the Ada-SnapKV /
    HeadKV fixtures the head budgets and the var-len metadata are the reference's; the pyramid budgets are the fixtures' k."""
    import json
    gold = os.path.join(os.path.dirname(__file__), "golden")
    index = json.load(open(os.path.join(gold, "index.json")))["cases"]
    checked = {"dense": 0, "flat": 0, "budget": 0}
    for c in index:
        if c["dtype"] == "fp32" or c.get("merge"):
            continue
        z = np.load(os.path.join(gold, c["name"] + ".npz"))
        dt = 0 if c["dtype"] == "bf16" else 1
        q, k, v = make_qkv(c["B"], c["H"], c["S"], 128, c["dtype"], c["kind"], c["seed"])
        w, S = c["w"], c["S"]
        if c["policy"] == "pyramidkv" and not bool(z["passthrough"]):
            br, kk = C.c_int(), C.c_int()
            lib.pkv_o_pyramid_budget(c["cap"], w, c["layers"], c["layer"], S, 20, C.byref(br), C.byref(kk))
            assert kk.value == z["idx"].shape[-1], c["name"]
            checked["budget"] += 1
        if c["policy"] in ("snapkv", "pyramidkv") and c.get("tie_free"):
            s = O.window_scores(q, k, w)                                     # fp32 softmax / matmul stay with torch
            L, kk = S - w, z["idx"].shape[-1]
            kind = {"maxpool": 2, "avgpool": 1}[c["pool"]]
            for b in range(c["B"]):
                for h in range(c["H"]):
                    pooled = np.zeros(L, dtype=np.uint16)
                    raw = bits(s[b, h]).copy()                                # keep the array alive across the call
                    lib.pkv_o_pool(ptr(raw), dt, L, kind, c["ks"], ptr(pooled))
                    idx = np.zeros(kk, dtype=np.int32)
                    assert lib.pkv_o_topk(ptr(pooled), dt, L, kk, ptr(idx)) == 0
                    assert np.array_equal(idx, z["idx"][b, h]), c["name"]
                    for src, want in ((k, z["kc"]), (v, z["vc"])):
                        out = np.zeros((kk + w, 128), dtype=np.uint16)
                        rows = bits(src[b, h]).copy()
                        lib.pkv_o_gather(ptr(rows), C.c_int64(256), 256, S, w, ptr(idx), kk, ptr(out))
                        assert np.array_equal(out, want[b, h]), c["name"]
            checked["dense"] += 1
        if c["policy"] == "adakv" and c["cap"] - w <= S - w:        # base capacity > L: not compressed (:700)
            s = O.pool_scores(O.window_scores(q, k, w, "mean"), c["pool"], c["ks"])
            sv = torch.sort(s, dim=-1, descending=True, stable=True).values
            H = c["H"]
            cap = np.zeros(H, dtype=np.int32)
            svb = bits(sv[0]).copy()
            assert lib.pkv_o_ada_capacity(ptr(svb), dt, H, s.shape[-1], c["cap"] - w, C.c_double(c["floor"]),
                                          int(c["normalize"]), ptr(cap)) == 0
            hl, cu = np.zeros(H, np.int32), np.zeros(H + 1, np.int32)
            lib.pkv_o_ada_metadata(H, w, ptr(cap), ptr(hl), ptr(cu))
            assert np.array_equal(hl, z["head_lens"]) and np.array_equal(cu, z["cu_klen"]), c["name"]
            checked["flat"] += 1
    assert checked["dense"] >= 4 and checked["flat"] >= 3 and checked["budget"] >= 5, checked
