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
``adakv`` / ``headkv`` need the var-len flash-attention decode path of the reference (flash_attn_varlen_func)
and are not wired here; ``pyramidkv``, ``snapkv``, ``h2o``, ``streamingllm`` are.
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
# the module whose init_* functions build the clusters; tests swap in an oracle-backed stand-in on CPU
_cluster_module = _utils
# hand K/V to update_kv BEFORE repeat_kv when the cluster accepts it (pyramidkv_amd's do); False = the reference's order
skip_repeat_kv = True


def _repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    b, h, s, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None, :, :].expand(b, h, n_rep, s, d).reshape(b, h * n_rep, s, d)


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
                past_key_values.update(kc, vc, self.layer_idx)            # :168 (the prompt attends to the full K/V)
            else:
                k = _repeat_kv(k, self.num_key_value_groups)              # one token: the cache holds all H heads
                v = _repeat_kv(v, self.num_key_value_groups)
                k, v = past_key_values.update(k, v, self.layer_idx)       # :171
        attn = _attend(self, q, k, v, attention_mask, is_prefill)
        attn = attn.reshape(*input_shape, -1).contiguous()
        return self.o_proj(attn), None

    forward.__name__ = f"attn_forward_{method}"
    return forward


def _replace(module_path: str, class_name: str, method: str):
    if method not in _METHODS:
        raise ValueError(f"method {method!r} is not wired in pyramidkv_amd.monkeypatch (supported: {sorted(_METHODS)})")
    import importlib
    mod = importlib.import_module(module_path)
    cls = getattr(mod, class_name)
    if not hasattr(cls, "_pkv_original_forward"):
        cls._pkv_original_forward = cls.forward
    cls.forward = _make_forward(method, mod.apply_rotary_pos_emb)


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
