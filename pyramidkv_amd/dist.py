"""Head-sharded multi-GPU wrapper (one process per GPU, torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm, "gloo" on CPU for tests).

The path shards by attention head: every (batch, head) row is independent from score to gather
(reference pyramidkv_utils.py:334-346 operate along dim -1 / 2 only), and the per-layer k depends on
the layer only, so all ranks select the same count.  Rank r owns heads [r*H/N, (r+1)*H/N) - with
Llama-3-8B / Mistral-7B (8 KV heads) and N=8 that is exactly one KV head per GPU.  There is ONE
exchange step per layer: an all-gather of the selected int32 indices (KBs; latency-bound, nowhere
near the 7 x ~153 GB/s xGMI links).  Compacted K/V stay head-sharded, as tensor-parallel attention
consumes them.  The reference has no distributed code at all.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_heads(num_heads: int, rank: int, world: int) -> Tuple[int, int]:
    if num_heads % world:
        raise ValueError(f"num_heads={num_heads} is not divisible by world_size={world}")
    per = num_heads // world
    return rank * per, (rank + 1) * per


def allgather_indices(idx_local: torch.Tensor, group: Optional[dist.ProcessGroup] = None, force: bool = False) -> torch.Tensor:
    """idx_local int32 [B, H/N, k] on every rank -> int32 [B, H, k] on every rank (head-major order =
    rank order).  One collective.  ``force`` issues it even at world size 1 (RCCL smoke with nranks = 1)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return idx_local
    B, Hl, k = idx_local.shape
    out = torch.empty(world * B, Hl, k, dtype=idx_local.dtype, device=idx_local.device)   # rank-major concat
    dist.all_gather_into_tensor(out, idx_local.contiguous(), group=group)
    return out.view(world, B, Hl, k).permute(1, 0, 2, 3).reshape(B, world * Hl, k)


class PendingIndices:
    """Handle of an all-gather in flight (``allgather_indices_async``): ``wait()`` makes the CURRENT stream wait for it
    (no host block with the nccl/RCCL backend) and returns the [B, H, k] tensor."""

    def __init__(self, out, work, shape):
        self._out, self._work, self._shape = out, work, shape

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        world, B, Hl, k = self._shape
        return self._out.view(world, B, Hl, k).permute(1, 0, 2, 3).reshape(B, world * Hl, k)


def allgather_indices_async(idx_local: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                            force: bool = False) -> PendingIndices:
    """Same exchange as ``allgather_indices`` but not waited for: the next layer's update_kv does not depend on this
    layer's gathered indices, so the KB-sized, latency-bound collective overlaps with it (RCCL runs it on its own
    stream).  Call ``.wait()`` before the indices are consumed."""
    world = dist.get_world_size(group)
    B, Hl, k = idx_local.shape
    if world == 1 and not force:
        return PendingIndices(idx_local.contiguous(), None, (1, B, Hl, k))
    out = torch.empty(world * B, Hl, k, dtype=idx_local.dtype, device=idx_local.device)
    work = dist.all_gather_into_tensor(out, idx_local.contiguous(), group=group, async_op=True)
    return PendingIndices(out, work, (world, B, Hl, k))


class PrefillIndexExchange:
    """ONE all-gather per prefill instead of one per layer.  The gathered indices of layer i are not an input of layer i+1
    (every rank compresses its own heads from its own K/V), so the ranks can write the selections of all layers into one
    buffer - ``slot(layer)`` hands ``ops.compress(..., idx_out=...)`` its int32 [B, H/N, k_layer] slice - and exchange the
    whole prefill's worth (sum_l k_l indices per head, ~16 KB per head at budget 128) in a single collective:
    32x fewer latency-bound launches than the per-layer form.  ``gather()`` returns the per-layer [B, H, k_layer] tensors."""

    def __init__(self, ks, B: int, H_local: int, device, group: Optional[dist.ProcessGroup] = None, force: bool = False):
        self.ks, self.B, self.Hl, self.group, self.force = list(ks), B, H_local, group, force
        self.offsets = [0]
        for k in self.ks:
            self.offsets.append(self.offsets[-1] + B * H_local * k)
        self.local = torch.empty(self.offsets[-1], dtype=torch.int32, device=device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.all = torch.empty(self.world * self.offsets[-1], dtype=torch.int32, device=device) if (self.world > 1 or force) else None

    def slot(self, layer: int) -> torch.Tensor:
        return self.local[self.offsets[layer]:self.offsets[layer + 1]].view(self.B, self.Hl, self.ks[layer])

    def gather_async(self):
        """Issue the collective (RCCL's own stream); returns a work handle or None."""
        if self.all is None:
            return None
        return dist.all_gather_into_tensor(self.all, self.local, group=self.group, async_op=True)

    def views(self, work=None):
        if work is not None:
            work.wait()
        if self.all is None:
            return [self.slot(i) for i in range(len(self.ks))]
        per_rank = self.all.view(self.world, -1)
        out = []
        for i, k in enumerate(self.ks):
            seg = per_rank[:, self.offsets[i]:self.offsets[i + 1]].reshape(self.world, self.B, self.Hl, k)
            out.append(seg.permute(1, 0, 2, 3).reshape(self.B, self.world * self.Hl, k))
        return out


class HeadShardedCluster:
    """Wraps a SnapKV/PyramidKV/H2O-style cluster: ``update_kv`` runs the local heads through the HIP
    path and all-gathers the selected indices.  ``select_fn(q,k,v) -> (kc, vc, idx)`` is the local
    compress returning indices (``ops.compress(..., return_indices=True)``)."""

    def __init__(self, select_fn, group: Optional[dist.ProcessGroup] = None):
        self.select_fn = select_fn
        self.group = group

    def update_kv(self, key_states, query_states, value_states):
        kc, vc, idx = self.select_fn(query_states, key_states, value_states)
        return kc, vc, allgather_indices(idx, self.group)


class HeadShardedAdaKV:
    """Ada-SnapKV with heads sharded over ranks.  The per-layer budget couples ALL heads
    (flattened top-(H*base), reference pyramidkv_utils.py:712-717), so there is one exchange step: an
    all-gather of every rank's per-head descending LISTS - [H/N, M] 16-bit values, whatever ``score_sort_fn`` returns as
    its second result.  The HIP wiring below (``hip_head_sharded_adakv``) hands over each head's ADAPTIVE top-M list
    (:709-711 already applied, M = min(L, H_total * base) <= 4096 entries = 8 KB per head; complete rows only beyond that);
    the CPU tests pass complete sorted rows.  Every rank then evaluates the budgets of all H heads from the gathered
    lists and keeps the capacities of its own.  The flat K/V output and its var-len metadata stay local to the rank (its
    heads only).

    ``score_sort_fn(q, k) -> (sorted_idx [Hl, M] int32, sorted_val [Hl, M])`` and
    ``budget_fn(sorted_val_all [H, M]) -> capacities int32 [H]`` and
    ``gather_fn(k, v, sorted_idx, cap_local) -> (K_flat, V_flat, head_lens, cu_klen)`` are the local stages
    (HIP ops on a GPU; tests pass oracle stand-ins on CPU)."""

    def __init__(self, score_sort_fn, budget_fn, gather_fn, group: Optional[dist.ProcessGroup] = None):
        self.score_sort_fn, self.budget_fn, self.gather_fn, self.group = score_sort_fn, budget_fn, gather_fn, group

    def update_kv(self, key_states, query_states, value_states):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        sorted_idx, sorted_val = self.score_sort_fn(query_states, key_states)
        Hl, L = sorted_val.shape
        if world > 1:
            allv = torch.empty(world * Hl, L, dtype=sorted_val.dtype, device=sorted_val.device)
            # 16-bit payloads travel as raw bytes so that every backend (gloo included) accepts them
            dist.all_gather_into_tensor(allv.view(torch.uint8), sorted_val.contiguous().view(torch.uint8), group=self.group)
        else:
            allv = sorted_val
        cap_all = self.budget_fn(allv)                                   # [H] on every rank, identical
        cap_local = cap_all[rank * Hl:(rank + 1) * Hl].contiguous()
        kf, vf, head_lens, cu_klen = self.gather_fn(key_states, value_states, sorted_idx, cap_local)
        return kf, vf, head_lens, cu_klen, cap_all


def hip_head_sharded_adakv(num_heads_total: int, window_size: int, kernel_size: int, pooling: str, max_capacity_prompt: int,
                           floor: float, normalize: bool, group: Optional[dist.ProcessGroup] = None) -> HeadShardedAdaKV:
    """``HeadShardedAdaKV`` wired to the HIP stages (what a tensor-parallel host runs on every rank): local window scores ->
    top-M indices of the local heads (M = min(L, H_total * base): the most one head can receive, so truncated lists decide
    the global budget exactly) -> local ADAPTIVE lists (:709-711; the ratio needs the rank's own rows only) -> ONE
    all-gather of those lists -> the budget of all heads on every rank (normalisation already applied) -> flat gather of
    the local heads."""
    from . import ops, config as _cfg
    from .pyramidkv_utils import _ADA_TOPM_MAX
    base = max_capacity_prompt - window_size

    def score_sort(q, k):
        s = ops.score_window(q, k, window_size, pooling, kernel_size, reduce="mean", scale_mode=_cfg.scale_mode,
                             kv_group=q.shape[1] // k.shape[1])[0]
        M = min(s.shape[-1], num_heads_total * base)
        if M <= _ADA_TOPM_MAX:
            top_idx = ops.topk(s, M)
        else:                      # the single-GPU rule (pyramidkv_utils._ADA_TOPM_MAX): long lists come from the complete sort
            top_idx, _ = ops.sort_rows(s, want_values=False)
            top_idx = top_idx[:, :M].contiguous()
        return top_idx, ops.ada_adaptive_lists(s, top_idx, base, bool(normalize))

    def budget(all_lists):
        return ops.ada_budget(all_lists, base, floor, False)

    def gather(k, v, top_idx, cap_local):
        head_lens, cu = ops.ada_metadata(cap_local, window_size)
        caps = cap_local.tolist()
        kf, vf = ops.gather_flat(k, v, top_idx, cap_local, cu, window_size, sum(caps) + len(caps) * window_size,
                                 max_cap=max(caps), kv_group=top_idx.shape[0] // k.shape[1])
        return kf, vf, head_lens, cu

    return HeadShardedAdaKV(score_sort, budget, gather, group)
