"""``replace_llama(method)`` / ``replace_mistral(method)`` for transformers 5.x.

The reference's ``pyramidkv/monkeypatch.py`` (:19-87, :92-145) assigns 20 copy-pasted forwards written
against transformers==4.44.2 class names that no longer exist (``LlamaSdpaAttention`` ...), and cannot
even be imported at this commit.  This module keeps the two entry points and
their ``method`` strings and installs ONE attention forward per model family that reproduces the call
contract of ``llama_model.py:157-172`` on top of the transformers-5 attention module:

    q/k/v proj -> RoPE -> repeat_kv (:158-159) -> prefill: init_<method>(self); kv_cluster.update_kv(K, Q, V,
    mask, groups) -> past_key_values.update(K_c, V_c) (:167-168) | decode: past_key_values.update(k, v) (:171)
    -> attention of the current tokens over the FULL prompt K/V at prefill (:174), over the compacted
    cache at decode.

As in the reference the cache stores all H (query) heads, because K/V are expanded before ``update_kv``.
``pyramidkv``, ``snapkv``, ``h2o``, ``streamingllm`` use the model's own cache object.  ``adakv`` / ``headkv``
(reference llama_model.py:2236-2391 / :2393-2540) keep a FLAT per-head var-len cache: pass a
``pyramidkv_amd.DynamicCacheSplitHeadFlatten()`` as ``past_key_values`` (batch 1, as the reference); the decode step
appends through ``pkv_update_flatten_view`` and attends per head over its own rows - the reference calls
``flash_attn_varlen_func`` there, this adapter uses a padded-batch softmax written in plain PyTorch ops (caller-side
glue, like the rest of this file).
"""
from __future__ import annotations

from typing import Callable, Dict

import torch
import torch.nn.functional as F

from . import pyramidkv_utils as _utils

# method -> (init function name in pyramidkv_utils, takes num_hidden_layers)
_METHODS: Dict[str, tuple] = {
    "pyramidkv": ("init_pyramidkv", True),
    "snapkv": ("init_snapkv", False),
    "h2o": ("init_H2O", False),
    "streamingllm": ("init_StreamingLLM", False),
}
_FLAT_METHODS: Dict[str, str] = {"adakv": "init_adakv", "headkv": "init_headkv"}
# the module whose init_* functions build the clusters; tests swap in an oracle-backed stand-in on CPU
_cluster_module = _utils
# hand K/V to update_kv BEFORE repeat_kv when the cluster accepts it (pyramidkv_amd's do); False = the reference's order
skip_repeat_kv = True


def _repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    b, h, s, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


def _unslide_layer(cache, layer_idx: int) -> None:
    """The reference runs on transformers 4.44.2, whose DynamicCache never crops.  transformers 5 gives models with a
    ``sliding_window`` (Mistral-7B-v0.1) ``DynamicSlidingWindowLayer`` entries, which silently keep only the LAST
    sliding_window - 1 rows on update - and the compacted cache stores its top-scored rows FIRST, so the crop would evict
    the highest-scored tokens.  A compacted layer is therefore held in a plain ``DynamicLayer`` (as in the reference the
    window limit then lives in the attention mask only)."""
    layers = getattr(cache, "layers", None)
    if layers is None:
        return
    from transformers.cache_utils import DynamicLayer
    if getattr(cache, "layer_class_to_replicate", None) is not None and cache.layer_class_to_replicate is not DynamicLayer \
            and getattr(cache.layer_class_to_replicate, "is_sliding", False):
        cache.layer_class_to_replicate = DynamicLayer
    if layer_idx < len(layers) and getattr(layers[layer_idx], "is_sliding", False):
        layers[layer_idx] = DynamicLayer()


def _track_tokens(cache, layer_idx: int, n_new: int, prefill: bool) -> None:
    """True sequence length next to the compacted cache (the reference keeps ``self.kv_seq_len`` on the attention module,
    llama_model.py:139-145,166,170,172).  After compaction the stock cache reports the COMPRESSED length, and a decode
    loop that does not pass positions explicitly would derive RoPE positions from it (prompt ends at 149, next token
    rotated as 48).  The cache instance gets a ``get_seq_length`` that reports the tokens seen so far for a layer that
    holds content (0 for an empty layer, which is how the forward recognises the prefill call); mask sizes still come
    from the layers, i.e. from the compacted length."""
    if not hasattr(cache, "_pkv_seen_tokens"):
        import types
        inner = cache.get_seq_length
        cache._pkv_seen_tokens = 0
        cache._pkv_count_layer = layer_idx      # the first patched layer that runs (layer 0 need not be a patched attention)

        def get_seq_length(self, layer_idx=0):
            return 0 if inner(layer_idx) == 0 else self._pkv_seen_tokens

        cache.get_seq_length = types.MethodType(get_seq_length, cache)
    # one forward visits the layers in ascending order: the counter advances on the lowest patched layer index seen
    if layer_idx <= cache._pkv_count_layer:
        cache._pkv_count_layer = layer_idx
        cache._pkv_seen_tokens = n_new if prefill else cache._pkv_seen_tokens + n_new


def _attend(module, q, k, v, attention_mask, is_prefill):
    """Attention over K/V with H heads (no second repeat_kv) or H/g heads (un-expanded: SDPA's grouped-query
    path, expanded views for eager).  eager = the reference's matmul / fp32 softmax / matmul
    (llama_model.py:174-183); otherwise PyTorch SDPA."""
    q_len, kv_len = q.shape[-2], k.shape[-2]
    gqa = k.shape[1] != q.shape[1]
    if gqa and getattr(module.config, "_attn_implementation", "sdpa") == "eager":
        k, v = _repeat_kv(k, q.shape[1] // k.shape[1]), _repeat_kv(v, q.shape[1] // v.shape[1])
        gqa = False
    mask = None
    if attention_mask is not None and q_len > 1:
        mask = attention_mask[:, :, :, :kv_len]          # reference :176-178
    causal = mask is None and q_len > 1
    if getattr(module.config, "_attn_implementation", "sdpa") == "eager":
        w = torch.matmul(q, k.transpose(2, 3)) * module.scaling
        if mask is not None:
            w = w + mask
        elif causal:
            w = w + torch.full((q_len, kv_len), torch.finfo(w.dtype).min, device=w.device, dtype=w.dtype).triu(1 + kv_len - q_len)
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        out = torch.matmul(w, v)
    else:
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=causal, scale=module.scaling, enable_gqa=gqa)
    return out.transpose(1, 2).contiguous()


def _make_forward(method: str, apply_rotary_pos_emb: Callable):
    init_name, takes_layers = _METHODS[method]

    def forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
        input_shape = hidden_states.shape[:-1]
        hidden_shape = (*input_shape, -1, self.head_dim)
        q = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)
        k = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)
        v = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)
        cos, sin = position_embeddings
        q, k = apply_rotary_pos_emb(q, k, cos, sin)
        is_prefill = True
        if past_key_values is not None:
            is_prefill = past_key_values.get_seq_length(self.layer_idx) == 0   # == (key_len == kv_seq_len), :165
            if is_prefill:
                init = getattr(_cluster_module, init_name)
                if takes_layers:
                    init(self, self.config.num_hidden_layers)             # llama_model.py:101
                else:
                    init(self)
                # The reference materialises repeat_kv first (:158-159) - g copies of K and V, more bytes than the
                # whole eviction step moves.  Clusters that say so take the H/g heads as they are (same result).
                if not (skip_repeat_kv and getattr(self.kv_cluster, "accepts_unexpanded_kv", False)):
                    k = _repeat_kv(k, self.num_key_value_groups)          # llama_model.py:158
                    v = _repeat_kv(v, self.num_key_value_groups)          # llama_model.py:159
                kc, vc = self.kv_cluster.update_kv(k, q, v, attention_mask, self.num_key_value_groups)   # :167
                _unslide_layer(past_key_values, self.layer_idx)
                past_key_values.update(kc, vc, self.layer_idx)            # :168 (the prompt attends to the full K/V)
                _track_tokens(past_key_values, self.layer_idx, q.shape[-2], True)     # kv_seq_len, :166
            else:
                k = _repeat_kv(k, self.num_key_value_groups)              # one token: the cache holds all H heads
                v = _repeat_kv(v, self.num_key_value_groups)
                k, v = past_key_values.update(k, v, self.layer_idx)       # :171
                _track_tokens(past_key_values, self.layer_idx, q.shape[-2], False)    # kv_seq_len += q_len, :172
        attn = _attend(self, q, k, v, attention_mask, is_prefill)
        attn = attn.reshape(*input_shape, -1).contiguous()
        return self.o_proj(attn), None

    forward.__name__ = f"attn_forward_{method}"
    return forward


def varlen_decode_attention(q, k_flat, v_flat, cu_klen, max_seqlen_k, scaling):
    """One query token per head over a flat var-len cache: q [H, D], k_flat/v_flat [sum_h len_h, D], head h owns rows
    cu_klen[h]..cu_klen[h+1].  What flash_attn_varlen_func(..., cu_seqlens_q=arange(H+1), cu_seqlens_k=cu_klen,
    max_seqlen_q=1, causal=True) computes in the reference (llama_model.py:2377-2383): fp32 softmax per head."""
    H, D = q.shape
    cu = cu_klen.to(torch.long)
    lens = cu[1:] - cu[:-1]
    j = torch.arange(int(max_seqlen_k), device=q.device)
    valid = j[None, :] < lens[:, None]                                      # [H, max]
    rows = torch.where(valid, cu[:-1, None] + j[None, :], torch.zeros((), dtype=torch.long, device=q.device))
    kp = k_flat[rows]                                                       # [H, max, D]
    vp = v_flat[rows]
    s = torch.einsum("hd,hjd->hj", q.float(), kp.float()) * scaling
    s = s.masked_fill(~valid, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.einsum("hj,hjd->hd", p, vp.float()).to(q.dtype)


def _make_flat_forward(method: str, apply_rotary_pos_emb: Callable):
    init_name = _FLAT_METHODS[method]

    def forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
        input_shape = hidden_states.shape[:-1]
        hidden_shape = (*input_shape, -1, self.head_dim)
        q = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)
        k = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)
        v = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)
        cos, sin = position_embeddings
        q, k = apply_rotary_pos_emb(q, k, cos, sin)
        k = _repeat_kv(k, self.num_key_value_groups)                      # llama_model.py:2285-2286
        v = _repeat_kv(v, self.num_key_value_groups)
        q_len = q.shape[-2]
        flat = past_key_values is not None and hasattr(past_key_values, "key_cache")
        if past_key_values is not None and not flat:
            raise TypeError(f"{method} keeps a flat per-head cache: pass past_key_values=pyramidkv_amd.DynamicCacheSplitHeadFlatten()")
        if not flat or len(past_key_values.key_cache) <= self.layer_idx:  # prefill (reference :2300-2360)
            if flat:
                assert q.shape[0] == 1, "flat var-len cache: batch 1 (reference pyramidkv_utils.py:724)"
                getattr(_cluster_module, init_name)(self)                 # built once per module (:1049, :1076)
                kf, vf = self.kv_cluster.update_kv(k, q, v)               # :2310
                past_key_values.update(kf, vf, self.layer_idx)
                if self.layer_idx == 0:
                    past_key_values._seen_tokens += q_len                 # :2386
            attn = _attend(self, q, k, v, attention_mask, True)           # the prompt attends to the full K/V
        else:                                                             # decode (reference :2362-2384)
            assert q.shape[0] == 1 and q_len == 1
            cl = self.kv_cluster
            kf, vf = past_key_values.update(k, v, self.layer_idx, {"head_lens": cl.head_lens, "cu_klen": cl.cu_klen})
            cl.klen_sum += q.shape[1]
            cl.max_seqlen_k += 1
            cl.cu_klen += cl.cu_offset
            cl.head_lens += 1
            if self.layer_idx == 0:
                past_key_values._seen_tokens += 1
            out = varlen_decode_attention(q[0, :, 0], kf, vf, cl.cu_klen, cl.max_seqlen_k, self.scaling)   # [H, D]
            attn = out[None, None]                                        # [1, 1, H, D]
        attn = attn.reshape(*input_shape, -1).contiguous()
        return self.o_proj(attn), None

    forward.__name__ = f"attn_forward_{method}"
    return forward


def _replace(module_path: str, class_name: str, method: str):
    if method not in _METHODS and method not in _FLAT_METHODS:
        raise ValueError(f"method {method!r} is not wired in pyramidkv_amd.monkeypatch "
                         f"(supported: {sorted(list(_METHODS) + list(_FLAT_METHODS))})")
    import importlib
    mod = importlib.import_module(module_path)
    cls = getattr(mod, class_name)
    if not hasattr(cls, "_pkv_original_forward"):
        cls._pkv_original_forward = cls.forward
    make = _make_flat_forward if method in _FLAT_METHODS else _make_forward
    cls.forward = make(method, mod.apply_rotary_pos_emb)


def replace_llama(method, model_name=None):
    """reference monkeypatch.py:19-87."""
    _replace("transformers.models.llama.modeling_llama", "LlamaAttention", method)


def replace_mistral(method):
    """reference monkeypatch.py:92-145."""
    _replace("transformers.models.mistral.modeling_mistral", "MistralAttention", method)


def restore():
    """Undo replace_llama / replace_mistral (not in the reference; handy for tests)."""
    import importlib
    for module_path, class_name in (("transformers.models.llama.modeling_llama", "LlamaAttention"),
                                    ("transformers.models.mistral.modeling_mistral", "MistralAttention")):
        cls = getattr(importlib.import_module(module_path), class_name)
        if hasattr(cls, "_pkv_original_forward"):
            cls.forward = cls._pkv_original_forward
            del cls._pkv_original_forward
