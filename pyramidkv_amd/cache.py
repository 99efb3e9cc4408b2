"""Flat per-head var-len KV cache for Ada-SnapKV / HeadKV decode (reference pyramidkv_utils.py:28-102,
``DynamicCacheSplitHeadFlatten``, adapted there from FFY0/AdaKV).

Layer entries are 2-D ``[sum_h len_h, D]`` tensors: head h owns rows ``cu_klen[h] .. cu_klen[h] + head_lens[h]``.
A decode step appends one row per head through libpkv's ``pkv_update_flatten_view`` (the HIP replacement of the
reference's only CUDA kernel, csrc/csrc/cuda_api.cu:11-85).  The reference subclasses transformers' ``Cache``;
transformers 5.x made that constructor require per-layer objects (SURVEY.md section 7 hard part 6), so this class
is a plain container with the same methods the reference's forwards call (``update``, ``get_seq_length``,
``__len__``, ``__getitem__``, ``to_legacy_cache``, ``from_legacy_cache``) plus the three the transformers-5 model
forward asks every cache for (``is_sliding``, ``get_query_offset``, ``get_mask_sizes``).  No nvtx ranges
(reference :63-69).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import ops


class DynamicCacheSplitHeadFlatten:
    # the flat append; tests on CPU swap in the oracle's restatement (the product path is the HIP kernel only)
    _append = staticmethod(ops.update_flatten_view)
    is_sliding = [False]          # transformers 5 masking_utils: no sliding-window layers in this cache

    def __init__(self) -> None:
        self.key_cache: List[torch.Tensor] = []
        self.value_cache: List[torch.Tensor] = []
        self._seen_tokens = 0     # tokens of the sequence so far; the patched forward sets it (reference llama_model.py:2386)

    def __len__(self):
        return len(self.key_cache)

    def __iter__(self):
        for layer_idx in range(len(self)):
            yield (tuple(self.key_cache[layer_idx]), tuple(self.value_cache[layer_idx]))

    def __getitem__(self, layer_idx: int) -> Tuple[Tuple[torch.Tensor], Tuple[torch.Tensor]]:
        if layer_idx < len(self):
            return (tuple(self.key_cache[layer_idx]), tuple(self.value_cache[layer_idx]))
        raise KeyError(f"Cache only has {len(self)} layers, attempted to access layer with index {layer_idx}")

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        if len(self.key_cache) <= layer_idx:                                     # prefill: store the flat tensors (:53-55)
            self.key_cache.append(key_states)
            self.value_cache.append(value_states)
        else:                                                                    # decode: append one row per head (:57-72)
            assert self.key_cache[layer_idx].dim() == 2
            bs, head, seqlen, dim = key_states.shape
            assert bs == 1 and seqlen == 1
            head_lens = cache_kwargs["head_lens"]
            cu_klen = cache_kwargs["cu_klen"]
            self.key_cache[layer_idx] = self._append(self.key_cache[layer_idx].view(-1, dim),
                                                     key_states.view(-1, dim), head_lens, cu_klen)
            self.value_cache[layer_idx] = self._append(self.value_cache[layer_idx].view(-1, dim),
                                                       value_states.view(-1, dim), head_lens, cu_klen)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    def get_seq_length(self, layer_idx: Optional[int] = 0) -> int:
        if len(self.key_cache) <= layer_idx:
            return 0
        # the reference returns 1 = "has content" (:80-81); once a forward keeps _seen_tokens (reference
        # llama_model.py:2386) the true sequence length is reported, which is what transformers 5 derives positions from
        return self._seen_tokens if self._seen_tokens > 0 else 1

    def get_query_offset(self, layer_idx: int = 0) -> int:
        return self._seen_tokens if len(self.key_cache) > layer_idx else 0

    def get_mask_sizes(self, query_length: int, layer_idx: int = 0):
        return self.get_query_offset(layer_idx) + query_length, 0

    def get_max_length(self) -> Optional[int]:
        return None

    def to_legacy_cache(self):
        return tuple((self.key_cache[i], self.value_cache[i]) for i in range(len(self)))

    @classmethod
    def from_legacy_cache(cls, past_key_values=None) -> "DynamicCacheSplitHeadFlatten":
        cache = cls()
        if past_key_values is not None:
            for layer_idx in range(len(past_key_values)):
                k, v = past_key_values[layer_idx]
                cache.update(k, v, layer_idx)
        return cache
