"""Plumbing of replace_llama / replace_mistral on transformers 5.x with a tiny random-init model on CPU.
The clusters are stood in for by the oracle here (the product clusters need a GPU); what is tested is the
adapter: cache lengths per layer, position handling after compaction, pass-through equivalence, prompt
logits untouched by compression (the prompt attends to the full K/V, reference llama_model.py:174)."""
import types

import pytest
import torch

from oracle import pkv_oracle as O

transformers = pytest.importorskip("transformers")


class _OracleCluster:
    def __init__(self, kind, cfg, layer_idx, layers):
        self.kind, self.cfg, self.layer_idx, self.layers = kind, cfg, layer_idx, layers

    def update_kv(self, k, q, v, mask, groups):
        c = self.cfg
        if self.kind == "pyramidkv":
            return O.pyramidkv_update_kv(k, q, v, c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling,
                                         self.layers, self.layer_idx)
        if self.kind == "snapkv":
            return O.snapkv_update_kv(k, q, v, c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling)
        if self.kind == "h2o":
            return O.h2o_update_kv(k, q, v, c.window_size, c.max_capacity_prompt)
        return O.streamingllm_update_kv(k, q, v, c.window_size, c.max_capacity_prompt)


class _OracleClusterUnexpanded(_OracleCluster):
    """Stand-in for the product clusters' contract: also takes K/V with H/g heads (before repeat_kv)."""
    accepts_unexpanded_kv = True

    def update_kv(self, k, q, v, mask, groups):
        g = q.shape[1] // k.shape[1]
        if g > 1:
            b, h, s, d = k.shape
            k = k[:, :, None].expand(b, h, g, s, d).reshape(b, h * g, s, d)
            v = v[:, :, None].expand(b, h, g, s, d).reshape(b, h * g, s, d)
        return super().update_kv(k, q, v, mask, groups)


def _oracle_module(cls=None):
    if cls is not None:
        m = types.SimpleNamespace()
        m.init_pyramidkv = lambda self, n: setattr(self, "kv_cluster", cls("pyramidkv", self.config, self.layer_idx, n))
        m.init_snapkv = lambda self: setattr(self, "kv_cluster", cls("snapkv", self.config, self.layer_idx, 0))
        m.init_H2O = lambda self: setattr(self, "kv_cluster", cls("h2o", self.config, self.layer_idx, 0))
        m.init_StreamingLLM = lambda self: setattr(self, "kv_cluster", cls("streamingllm", self.config, self.layer_idx, 0))
        return m
    m = types.SimpleNamespace()
    m.init_pyramidkv = lambda self, n: setattr(self, "kv_cluster", _OracleCluster("pyramidkv", self.config, self.layer_idx, n))
    m.init_snapkv = lambda self: setattr(self, "kv_cluster", _OracleCluster("snapkv", self.config, self.layer_idx, 0))
    m.init_H2O = lambda self: setattr(self, "kv_cluster", _OracleCluster("h2o", self.config, self.layer_idx, 0))
    m.init_StreamingLLM = lambda self: setattr(self, "kv_cluster", _OracleCluster("streamingllm", self.config, self.layer_idx, 0))
    return m


def _tiny(family):
    torch.manual_seed(0)
    kw = dict(vocab_size=97, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
              num_key_value_heads=1, head_dim=128, max_position_embeddings=4096)
    if family == "llama":
        from transformers import LlamaConfig, LlamaForCausalLM
        cfg = LlamaConfig(**kw)
        model = LlamaForCausalLM(cfg)
    else:
        from transformers import MistralConfig, MistralForCausalLM
        cfg = MistralConfig(sliding_window=None, **kw)
        model = MistralForCausalLM(cfg)
    return model.eval()


@pytest.fixture
def patched():
    from pyramidkv_amd import monkeypatch as mp
    saved = mp._cluster_module
    mp._cluster_module = _oracle_module()
    yield mp
    mp._cluster_module = saved
    mp.restore()


@pytest.mark.parametrize("family", ["llama", "mistral"])
@pytest.mark.parametrize("method", ["pyramidkv", "snapkv", "streamingllm", "h2o"])
def test_generate_with_compacted_cache(patched, family, method):
    model = _tiny(family)
    S, new = 200, 6
    ids = torch.randint(0, 97, (1, S), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        base = model(ids).logits
    (patched.replace_llama if family == "llama" else patched.replace_mistral)(method)
    cap, w = 64, 8
    for layer in model.model.layers:
        c = layer.self_attn.config
        c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = w, cap, 7, "maxpool", None
    from transformers import DynamicCache
    cache = DynamicCache(config=model.config)
    with torch.no_grad():
        out = model(ids, past_key_values=cache, use_cache=True)
    # the prompt itself attends to the full K/V: prompt logits are those of the unpatched model
    assert torch.allclose(out.logits, base, atol=1e-4, rtol=1e-4)
    H = model.config.num_attention_heads
    want = []
    for li in range(model.config.num_hidden_layers):
        if method == "pyramidkv":
            br, k = O.pyramid_budget(cap, w, model.config.num_hidden_layers, li, S)
            want.append(S if br == "passthrough" else k + w)
        else:
            want.append(cap)
    got = [cache.layers[li].keys.shape for li in range(model.config.num_hidden_layers)]
    assert [g[2] for g in got] == want and all(g[1] == H for g in got)      # all H heads cached, reference :158-168
    # decode: positions continue from S, the cache grows by one per step in every layer
    nxt = out.logits[:, -1:].argmax(-1)
    with torch.no_grad():
        for t in range(new):
            o = model(nxt, past_key_values=cache, use_cache=True,
                      position_ids=torch.tensor([[S + t]]), cache_position=torch.tensor([S + t]))
            nxt = o.logits[:, -1:].argmax(-1)
            assert torch.isfinite(o.logits).all()
    assert [cache.layers[li].keys.shape[2] for li in range(model.config.num_hidden_layers)] == [x + new for x in want]


def test_passthrough_equals_unpatched_generation(patched):
    model = _tiny("llama")
    ids = torch.randint(0, 97, (1, 40), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = model.generate(ids, max_new_tokens=8, do_sample=False)
    patched.replace_llama("snapkv")
    for layer in model.model.layers:
        c = layer.self_attn.config
        c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = 8, 64, 7, "maxpool", None   # 40 < 64: no eviction
    with torch.no_grad():
        got = model.generate(ids, max_new_tokens=8, do_sample=False)
    assert torch.equal(ref, got)


def test_unknown_method_raises(patched):
    with pytest.raises(ValueError):
        patched.replace_llama("cam")


@pytest.mark.parametrize("method", ["pyramidkv", "snapkv"])
def test_hf_generate_runs_on_compacted_cache(patched, method):
    """End-to-end ``model.generate`` (the reference runners' call, run_longbench.py:266-275) with eviction on."""
    model = _tiny("llama")
    S = 150
    ids = torch.randint(0, 97, (1, S), generator=torch.Generator().manual_seed(3))
    patched.replace_llama(method)
    for layer in model.model.layers:
        c = layer.self_attn.config
        c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = 8, 48, 7, "maxpool", None
    with torch.no_grad():
        out = model.generate(ids, max_new_tokens=5, do_sample=False, return_dict_in_generate=True)
    assert out.sequences.shape == (1, S + 5)
    lens = [out.past_key_values.layers[i].keys.shape[2] for i in range(model.config.num_hidden_layers)]
    # the last generated token is not fed back, so the cache holds the compacted prompt + 4 decoded tokens
    if method == "snapkv":
        assert lens == [48 + 4] * model.config.num_hidden_layers
    else:
        assert lens == [O.pyramid_budget(48, 8, model.config.num_hidden_layers, i, S)[1] + 8 + 4 for i in range(model.config.num_hidden_layers)]


@pytest.mark.parametrize("method", ["pyramidkv", "streamingllm"])
def test_unexpanded_kv_branch_gives_the_same_cache(patched, method):
    """Clusters that accept K/V before repeat_kv get them un-expanded; cache contents and logits equal the
    reference order (expanded first)."""
    from transformers import DynamicCache
    model = _tiny("llama")
    S = 160
    ids = torch.randint(0, 97, (1, S), generator=torch.Generator().manual_seed(5))
    res = {}
    for cls in (_OracleCluster, _OracleClusterUnexpanded):
        patched._cluster_module = _oracle_module(cls)
        patched.replace_llama(method)
        for layer in model.model.layers:
            if hasattr(layer.self_attn, "kv_cluster"):
                del layer.self_attn.kv_cluster
            c = layer.self_attn.config
            c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = 8, 64, 7, "maxpool", None
        cache = DynamicCache(config=model.config)
        with torch.no_grad():
            out = model(ids, past_key_values=cache, use_cache=True)
        res[cls.__name__] = (out.logits, [(l.keys.clone(), l.values.clone()) for l in cache.layers])
        assert isinstance(model.model.layers[0].self_attn.kv_cluster, cls)
    a, b = res["_OracleCluster"], res["_OracleClusterUnexpanded"]
    assert torch.allclose(a[0], b[0], atol=1e-5, rtol=1e-5)
    for (ka, va), (kb, vb) in zip(a[1], b[1]):
        assert torch.equal(ka, kb) and torch.equal(va, vb)


# ------------------------------------------------------------------------------------------ adakv / headkv (flat cache)
class _OracleFlatCluster:
    """Stand-in for AdaKVCluster / HeadKVCluster on CPU: oracle update_kv + the metadata attributes the forward bumps."""

    def __init__(self, kind, cfg, layer_idx):
        self.kind, self.cfg, self.layer_idx = kind, cfg, layer_idx

    def update_kv(self, k, q, v):
        c = self.cfg
        if self.kind == "adakv":
            kf, vf, m = O.adakv_update_kv(k, q, v, c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling,
                                          c.floor, c.normalize)
        else:
            kf, vf, m = O.headkv_update_kv(k, q, v, c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling,
                                           c.head_capacity, self.layer_idx)
        self.head_lens, self.cu_klen, self.cu_qlen, self.cu_offset = m.head_lens, m.cu_klen, m.cu_qlen, m.cu_offset
        self.max_seqlen_k, self.klen_sum = m.max_seqlen_k, m.klen_sum
        return kf, vf


def _flat_module():
    m = types.SimpleNamespace()

    def init(kind):
        def f(self):
            if not hasattr(self, "kv_cluster"):
                self.kv_cluster = _OracleFlatCluster(kind, self.config, self.layer_idx)
        return f
    m.init_adakv, m.init_headkv = init("adakv"), init("headkv")
    return m


def test_varlen_decode_attention_matches_per_head_loop():
    from pyramidkv_amd.monkeypatch import varlen_decode_attention
    g = torch.Generator().manual_seed(0)
    lens = [5, 1, 17, 9]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    q = torch.randn(4, 16, generator=g)
    kf, vf = torch.randn(sum(lens), 16, generator=g), torch.randn(sum(lens), 16, generator=g)
    got = varlen_decode_attention(q, kf, vf, cu, max(lens), 0.25)
    for h, n in enumerate(lens):
        a, b = int(cu[h]), int(cu[h + 1])
        p = torch.softmax((kf[a:b] @ q[h]) * 0.25, 0)
        assert torch.allclose(got[h], p @ vf[a:b], atol=1e-6)


@pytest.mark.parametrize("family", ["llama", "mistral"])
@pytest.mark.parametrize("method", ["adakv", "headkv"])
def test_flat_cache_methods_generate(patched, method, family):
    """replace_llama('adakv'|'headkv') with a DynamicCacheSplitHeadFlatten: prompt logits untouched, per-layer flat
    lengths = sum_h(cap_h + w), decode appends one row per head and keeps positions running."""
    import pyramidkv_amd as P
    model = _tiny(family)
    S, new, cap, w = 200, 5, 64, 8
    ids = torch.randint(0, 97, (1, S), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        base = model(ids).logits
    patched._cluster_module = _flat_module()
    (patched.replace_llama if family == "llama" else patched.replace_mistral)(method)
    H, Lyr = model.config.num_attention_heads, model.config.num_hidden_layers
    for layer in model.model.layers:
        c = layer.self_attn.config
        c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = w, cap, 7, "maxpool", None
        c.floor, c.normalize = 0.2, True
        c.head_capacity = torch.tensor([[40, 72]] * Lyr, dtype=torch.int32)           # HeadKV: per (layer, head) budgets
    saved = P.DynamicCacheSplitHeadFlatten._append
    P.DynamicCacheSplitHeadFlatten._append = staticmethod(O.update_flatten_view)     # CPU: the oracle's flat append
    try:
        cache = P.DynamicCacheSplitHeadFlatten()
        with torch.no_grad():
            out = model(ids, past_key_values=cache, use_cache=True)
        assert torch.allclose(out.logits, base, atol=1e-4, rtol=1e-4)
        assert cache.get_seq_length() == S and len(cache) == Lyr
        want = [int(l.self_attn.kv_cluster.klen_sum) for l in model.model.layers]   # sum_h (cap_h + w), per layer
        if method == "headkv":
            assert want == [(40 + 72) + H * w] * Lyr
        else:       # budgets are rounded per head (:719): the total may be off the nominal H * cap by a token or two
            assert all(abs(x - H * cap) <= H for x in want)
        assert [cache.key_cache[i].shape[0] for i in range(Lyr)] == want
        nxt = out.logits[:, -1:].argmax(-1)
        with torch.no_grad():
            for t in range(new):
                o = model(nxt, past_key_values=cache, use_cache=True)               # positions come from the cache
                nxt = o.logits[:, -1:].argmax(-1)
                assert torch.isfinite(o.logits).all()
        assert cache.get_seq_length() == S + new
        assert [cache.key_cache[i].shape[0] for i in range(Lyr)] == [x + H * new for x in want]
        cl = model.model.layers[0].self_attn.kv_cluster
        assert int(cl.klen_sum) == want[0] + H * new and int(cl.cu_klen[-1]) == want[0] + H * new
    finally:
        P.DynamicCacheSplitHeadFlatten._append = saved
        for layer in model.model.layers:
            if hasattr(layer.self_attn, "kv_cluster"):
                del layer.self_attn.kv_cluster


# ----------------------------------------------------------------------------------------- round-2 advisor findings
def test_sliding_window_model_keeps_the_compacted_rows(patched):
    """Mistral-7B-v0.1 style config (sliding_window set): transformers 5 builds DynamicSlidingWindowLayer entries that crop
    to the LAST sliding_window - 1 rows on update - the compacted cache stores its top-scored rows FIRST, so the crop would
    evict exactly them.  The adapter holds compacted layers in plain DynamicLayer entries (the reference's 4.44 cache never
    crops): cap + new tokens rows survive."""
    from transformers import MistralConfig, MistralForCausalLM, DynamicCache
    torch.manual_seed(0)
    cfg = MistralConfig(vocab_size=97, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                        num_key_value_heads=1, head_dim=128, max_position_embeddings=4096, sliding_window=64)
    model = MistralForCausalLM(cfg).eval()
    S, cap, w, new = 150, 48, 8, 30
    ids = torch.randint(0, 97, (1, S), generator=torch.Generator().manual_seed(1))
    patched.replace_mistral("snapkv")
    for layer in model.model.layers:
        c = layer.self_attn.config
        c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = w, cap, 7, "maxpool", None
    with torch.no_grad():
        out = model.generate(ids, max_new_tokens=new, do_sample=False, return_dict_in_generate=True,
                             past_key_values=DynamicCache(config=cfg))
    for lay in out.past_key_values.layers:
        assert lay.keys.shape[2] == cap + new - 1          # nothing cropped (the stock sliding layer would hold 63 rows)


@pytest.mark.parametrize("family", ["llama", "mistral_sliding"])
def test_positions_follow_the_true_length_without_explicit_positions(patched, family):
    """After compaction the stock cache reports the compressed length; a decode loop that does not pass position_ids /
    cache_position would rotate the next token as position `cap` instead of S.  The adapter tracks the true token count
    (reference: self.kv_seq_len, llama_model.py:139-145,166,170,172): decoding with and without explicit positions gives
    the same logits - also on a Mistral config with ``sliding_window`` set (transformers 5 builds sliding cache layers and
    sliding masks there; round-2 advisor finding)."""
    from transformers import DynamicCache
    S, cap, w = 150, 48, 8
    ids = torch.randint(0, 97, (1, S), generator=torch.Generator().manual_seed(2))
    if family == "llama":
        model = _tiny("llama")
        patched.replace_llama("snapkv")
    else:
        from transformers import MistralConfig, MistralForCausalLM
        torch.manual_seed(0)
        model = MistralForCausalLM(MistralConfig(
            vocab_size=97, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
            num_key_value_heads=1, head_dim=128, max_position_embeddings=4096, sliding_window=64)).eval()
        patched.replace_mistral("snapkv")
    for layer in model.model.layers:
        c = layer.self_attn.config
        c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = w, cap, 7, "maxpool", None
    outs = []
    for explicit in (True, False):
        cache = DynamicCache(config=model.config)
        with torch.no_grad():
            o = model(ids, past_key_values=cache, use_cache=True)
            assert cache.get_seq_length() == S and cache.layers[0].keys.shape[2] == cap
            tok = o.logits[:, -1:].argmax(-1)
            step = []
            for t in range(3):
                kw = dict(position_ids=torch.tensor([[S + t]]), cache_position=torch.tensor([S + t])) if explicit else {}
                o = model(tok, past_key_values=cache, use_cache=True, **kw)
                step.append(o.logits.clone())
                tok = o.logits[:, -1:].argmax(-1)
            assert cache.get_seq_length() == S + 3 and cache.layers[0].keys.shape[2] == cap + 3
        outs.append(torch.cat(step, 1))
    assert torch.allclose(outs[0], outs[1], atol=1e-5, rtol=1e-5)


def test_module_is_a_drop_in_for_the_reference_imports():
    """INTEGRATION.md's one-line swap (sys.modules['pyramidkv.pyramidkv_utils'] = pyramidkv_amd.pyramidkv_utils) only works if
    every name the reference's model files import from that module exists here: llama_model.py:16,20,
    llama_model_think.py:16,20 and mistral_model.py:19-20."""
    import os
    import re
    import pyramidkv_amd.pyramidkv_utils as U
    names = {"init_pyramidkv", "init_snapkv", "init_CAM", "init_H2O", "init_StreamingLLM", "init_l2norm", "init_adakv",
             "init_headkv", "DynamicCacheSplitHeadFlatten"}
    ref = "/root/reference/pyramidkv"
    if os.path.isdir(ref):                      # in the build container: read the import lines themselves
        for f in ("llama_model.py", "llama_model_think.py", "mistral_model.py"):
            for line in open(os.path.join(ref, f)):
                m = re.match(r"from pyramidkv\.pyramidkv_utils import (.+)", line)
                if m:
                    names |= {n.strip() for n in m.group(1).split(",") if n.strip()}
    missing = [n for n in sorted(names) if not hasattr(U, n)]
    assert not missing, missing
    for n in ("init_CAM", "init_l2norm", "init_think"):        # outside the hot path: present, and loud when called
        with pytest.raises(NotImplementedError):
            getattr(U, n)(object())
    x = torch.zeros(1, 2, 3, 4)
    assert U.repeat_kv(x, 1) is x and U.repeat_kv(x, 2).shape == (1, 4, 3, 4)
