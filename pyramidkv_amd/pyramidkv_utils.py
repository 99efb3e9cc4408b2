"""Drop-in replacement for the reference's ``pyramidkv/pyramidkv_utils.py`` KV-cluster classes.

Same class names, constructor keywords, ``update_kv`` signatures, return conventions and ``init_*``
factories as the reference (file:line cited per symbol), so the reference's patched attention
forwards (``llama_model.py`` / ``mistral_model.py``) work unchanged when this module is swapped in
for ``pyramidkv.pyramidkv_utils``.  The body of every ``update_kv`` is one or a few calls into
libpkv's HIP kernels through the C ABI (``include/pkv.h``); nothing is computed in PyTorch.

Differences from the reference, all deliberate:
  * no per-call ``print`` (pyramidkv_utils.py:217,312,539,601 write to stdout on every layer);
  * ``merge="pivot"`` (LOOK-M pivot merge, :119-170) runs in HIP kernels (libpkv ``pkv_merge_compact``) with the
    reference's own output orders (keys [window, selected], values [selected, window]); any other value raises
    ``ValueError('Merge method not supported')`` as :164 does;
  * top-k tie order is pinned to (value desc, index asc) - see DESIGN.md;
  * tensors must live on a HIP device (no CPU path).
"""
from __future__ import annotations

import enum
import threading
import time

import torch

from . import ops
from . import config as _cfg
from .cache import DynamicCacheSplitHeadFlatten  # noqa: F401  (llama_model.py:20 / mistral_model.py:20 import it from here)


def _check_merge(merge):
    if merge is not None and merge != "pivot":
        raise ValueError('Merge method not supported')                               # :164


def _kv_group(num_key_value_groups, num_heads) -> int:
    """GQA de-duplication is opt-in: the reference hands over K/V already expanded by repeat_kv
    (llama_model.py:158-159) and ignores num_key_value_groups.  With config.gqa_dedup the kernels read
    only the first head of every group of the (materialised) expanded tensor."""
    g = int(num_key_value_groups or 1)
    if _cfg.gqa_dedup and g > 1 and num_heads % g == 0:
        return g
    return 1


def _dedup_view(t: torch.Tensor, g: int) -> torch.Tensor:
    return t if g == 1 else t[:, ::g]


def _repeat_kv(t: torch.Tensor, g: int) -> torch.Tensor:
    """reference pyramidkv_utils.py:109-117."""
    if g == 1:
        return t
    b, h, s, d = t.shape
    return t[:, :, None, :, :].expand(b, h, g, s, d).reshape(b, h * g, s, d)


def _unexpanded_group(key_states, query_states) -> int:
    """K/V may be handed over BEFORE repeat_kv ([B, H/g, S, D] next to Q [B, H, S, D]); the
    kernels then read every KV head once per group instead of g materialised copies.  Returns g (1 = already expanded)."""
    hq, hk = query_states.shape[1], key_states.shape[1]
    if hk == hq:
        return 1
    if hk == 0 or hq % hk:
        raise ValueError(f"{hk} key/value heads do not divide {hq} query heads")
    return hq // hk


_MAX_COLS = 256      # the K scan carries kv_group * window columns per key row (include/pkv.h: PKV_ERR_UNSUPPORTED beyond)


def _fit_group(key_states, value_states, g: int, window: int):
    """A GQA group of g query heads x `window` rows must fit the K scan's 256 columns per key row.  Wider groups (g = 8
    next to window 64: Llama-3-70B shapes with the reference's default window) are split: K/V are expanded by g / g2,
    g2 = the largest divisor of g with g2 * window <= 256, and the kernels run with kv_group = g2."""
    if g <= 1 or g * window <= _MAX_COLS:
        return key_states, value_states, g
    if window > _MAX_COLS:
        raise ValueError(f"window_size {window} is beyond the {_MAX_COLS} columns per key row of the K scan")
    g2 = max(d for d in range(1, g + 1) if g % d == 0 and d * window <= _MAX_COLS)
    r = g // g2
    return _repeat_kv(key_states, r), _repeat_kv(value_states, r), g2


class _WindowPolicy:
    """Shared body of SnapKV / PyramidKV / H2O: score -> top-k -> gather in one C call."""

    window_size: int
    kernel_size: int
    pooling: str

    accepts_unexpanded_kv = True      # update_kv also takes K/V with H/g heads (before repeat_kv)
    # optional contiguous int32 [B, H, k] tensor: update_kv ALSO writes the selected indices there (a head-sharded host
    # all-gathers them, pyramidkv_amd/dist.py; bench.py reads them back for its parity block).  Not a reference attribute.
    # Written on every path that SELECTS (plain gather and merge="pivot"); a call that selects nothing - the pass-through
    # below max_capacity_prompt (:219,:315) or a pyramid layer with a budget of 0 past tokens - leaves it untouched, and a
    # buffer whose size is not B*H*k raises.
    index_out = None

    def _compress(self, key_states, query_states, value_states, k, num_key_value_groups, h2o=False):
        # fast path: the call prepared by an earlier update_kv of this cluster with the same layouts and knobs (ops.PreparedCompress)
        sig = (k, h2o, self.window_size, self.pooling, self.kernel_size, getattr(self, "merge", None), num_key_value_groups)
        prep = self.__dict__.get("_prep")
        if prep is not None and prep[0] == sig and prep[1].hit(query_states, key_states, value_states):
            out = prep[1].run(query_states, key_states, value_states, self.index_out)
            if out is not None:
                return out
        gu = _unexpanded_group(key_states, query_states)
        if k == 0 and getattr(self, "merge", None) is None:
            # a pyramid layer whose budget came out as 0 past tokens (tiny max_capacity_prompt - window_size, :205-215):
            # the reference's topk(0) selects nothing and the cat (:271-272) returns the observation window alone
            w = self.window_size
            return (_repeat_kv(key_states[:, :, -w:, :], gu).contiguous(), _repeat_kv(value_states[:, :, -w:, :], gu).contiguous())
        if k == 0:
            raise ValueError("merge='pivot' with a layer budget of 0 past tokens (max_capacity_prompt - window_size too small "
                             "for this layer count, :205-215) is not supported: there is no selected row to merge into")
        k_in, v_in = key_states, value_states
        key_states, value_states, gu = _fit_group(key_states, value_states, gu, self.window_size)
        if getattr(self, "merge", None) is not None:                                 # :336-339: merge_kv instead of the gather
            g = gu if gu > 1 else _kv_group(num_key_value_groups, query_states.shape[1])
            if g * self.window_size > _MAX_COLS:
                g = 1
            ks, vs = (key_states, value_states) if gu > 1 else (_dedup_view(key_states, g), _dedup_view(value_states, g))
            idx = ops.select(query_states, ks, self.window_size, k, self.pooling, self.kernel_size,
                             scale_mode=_cfg.scale_mode, kv_group=g, h2o=h2o)
            if self.index_out is not None:
                if self.index_out.dtype != torch.int32 or not self.index_out.is_contiguous() or self.index_out.numel() != idx.numel():
                    raise ValueError("index_out must be a contiguous int32 tensor of B*H*k elements")
                self.index_out.view_as(idx).copy_(idx)
            return ops.merge_compact(ks, vs, idx, self.window_size, kv_group=g)
        step = 1
        if gu > 1:
            g = gu
        else:
            g = _kv_group(num_key_value_groups, query_states.shape[1])
            if g * self.window_size > _MAX_COLS:
                g = 1
            step = g
        if key_states is k_in and value_states is v_in:           # the operands go to the kernels as they are: remember the call
            pc = ops.prepare_compress(query_states, key_states, value_states, self.window_size, k, self.pooling, self.kernel_size,
                                      _cfg.scale_mode, g, h2o, k_head_step=step)
            self._prep = (sig, pc) if pc is not None else None
        return ops.compress(query_states, _dedup_view(key_states, step), _dedup_view(value_states, step),
                            self.window_size, k, self.pooling, self.kernel_size,
                            scale_mode=_cfg.scale_mode, kv_group=g, h2o=h2o, idx_out=self.index_out)[:2]

    @staticmethod
    def _passthrough(key_states, query_states, value_states):
        """S < max_capacity_prompt: the reference returns its inputs (:219,:315); un-expanded inputs come back as the
        repeat_kv tensors the reference would have been handed."""
        g = _unexpanded_group(key_states, query_states)
        return _repeat_kv(key_states, g), _repeat_kv(value_states, g)


class PyramidKVCluster(_WindowPolicy):
    """reference pyramidkv_utils.py:173-283."""

    def __init__(self, num_hidden_layers=32, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5,
                 pooling='avgpool', beta=20, num_layers=80, layer_idx=None, merge=None):
        self.layer_idx = layer_idx
        self.num_hidden_layers = num_hidden_layers
        self.steps = -1
        self.beta = beta
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0                      # :184
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.merge = merge

    def reset(self, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None):
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.merge = merge

    def layer_budget(self, q_len: int):
        """Integer budget arithmetic of :205-215 and the branch thresholds of :218,:220,:252."""
        min_num = (self.max_capacity_prompt - self.window_size) // self.beta
        max_num = (self.max_capacity_prompt - self.window_size) * 2 - min_num
        if max_num >= q_len - self.window_size:
            max_num = q_len - self.window_size
            min_num = (self.max_capacity_prompt - self.window_size) * 2 - max_num
        steps = (max_num - min_num) // (self.num_hidden_layers - 1)
        max_capacity_prompt = max_num - self.layer_idx * steps
        if q_len < self.max_capacity_prompt:
            return "passthrough", 0
        if q_len < (self.max_capacity_prompt - self.window_size) * 2:
            return "snap", self.max_capacity_prompt - self.window_size
        return "pyramid", max_capacity_prompt

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        assert key_states.shape[-2] == query_states.shape[-2]                       # :200
        q_len = query_states.shape[-2]
        branch, k = self.layer_budget(q_len)
        if branch == "passthrough":
            return self._passthrough(key_states, query_states, value_states)        # :219 (same objects)
        if self.pooling not in ('avgpool', 'maxpool'):
            raise ValueError('Pooling method not supported')                         # :237
        _check_merge(self.merge)
        return self._compress(key_states, query_states, value_states, k, num_key_value_groups)


class SnapKVCluster(_WindowPolicy):
    """reference pyramidkv_utils.py:285-347."""

    def __init__(self, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None,
                 recent_size=32, ratio=0.4):
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0                      # :289
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.merge = merge
        self.recent_size = recent_size
        self.ratio = ratio

    def reset(self, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None):
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.merge = merge

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        assert key_states.shape[-2] == query_states.shape[-2]                       # :309
        q_len = query_states.shape[-2]
        if q_len < self.max_capacity_prompt:                                        # :314
            return self._passthrough(key_states, query_states, value_states)
        if self.pooling not in ('avgpool', 'maxpool'):
            raise ValueError('Pooling method not supported')                         # :333
        _check_merge(self.merge)
        return self._compress(key_states, query_states, value_states,
                              self.max_capacity_prompt - self.window_size, num_key_value_groups)


class H2OKVCluster(_WindowPolicy):
    """reference pyramidkv_utils.py:516-575."""

    def __init__(self, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None):
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.merge = merge

    reset = __init__

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        assert key_states.shape[-2] == query_states.shape[-2]                       # :536
        q_len = query_states.shape[-2]
        if q_len < self.max_capacity_prompt:                                        # :541
            return self._passthrough(key_states, query_states, value_states)
        _check_merge(self.merge)
        return self._compress(key_states, query_states, value_states,
                              self.max_capacity_prompt - self.window_size, num_key_value_groups, h2o=True)


class StreamingLLMKVCluster:
    """reference pyramidkv_utils.py:578-620: attention sinks 0..cap-w-1 + the last w tokens."""

    accepts_unexpanded_kv = True

    def __init__(self, window_size=64, max_capacity_prompt=256 + 64, kernel_size=5, pooling='avgpool', merge=None):
        self.window_size = window_size
        self.max_capacity_prompt = max_capacity_prompt
        assert self.max_capacity_prompt - self.window_size > 0
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.merge = merge

    reset = __init__

    def update_kv(self, key_states, query_states, value_states, attention_mask, num_key_value_groups):
        assert key_states.shape[-2] == query_states.shape[-2]                       # :598
        q_len = query_states.shape[-2]
        g = _unexpanded_group(key_states, query_states)
        if q_len < self.max_capacity_prompt:                                        # :603
            return _repeat_kv(key_states, g), _repeat_kv(value_states, g)
        _check_merge(self.merge)
        if self.merge is not None:                                                   # :609-613
            bsz, num_heads = query_states.shape[0], query_states.shape[1]
            n = self.max_capacity_prompt - self.window_size
            idx = torch.arange(n, dtype=torch.int32, device=key_states.device)[None, None, :].expand(bsz, num_heads, n)
            return ops.merge_compact(key_states, value_states, idx.contiguous(), self.window_size, kv_group=g)
        kc, vc = ops.gather_streaming(key_states, value_states, self.max_capacity_prompt - self.window_size,
                                      self.window_size)
        return _repeat_kv(kc, g), _repeat_kv(vc, g)       # every head of a group keeps the same tokens


_PINNED = threading.local()      # per host thread: two threads reading capacities back never share a staging buffer


def _read_back(t: torch.Tensor):
    """Small device int32 vector -> Python list through a cached pinned buffer (the one host sync of Ada-SnapKV, as the
    reference's ``.item()`` at :718): an asynchronous copy + a stream synchronise instead of a pageable blocking copy."""
    n = t.numel()
    cache = _PINNED.__dict__.setdefault("bufs", {})
    buf = cache.get(n)
    if buf is None:
        buf = torch.empty(n, dtype=t.dtype, pin_memory=True)
        cache[n] = buf
    buf.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return buf.tolist()


class _HostMirror:
    """Pinned host uint64 [H] the budget kernel writes the capacities into as self-validating words (word h = sequence number
    << 32 | "a list ran out" << 31 | cap_h): the host polls until every word carries the current sequence number instead of
    enqueueing a copy and synchronising the stream (~35 us of the ~130 us call at S = 32k in round 3).  Round 5: whole 64-bit
    words instead of int32 capacities + a system fence + a flag word - the kernel no longer waits for two PCIe round trips
    before it retires, and the host sees the capacities one round trip earlier.

    One mirror per CLUSTER INSTANCE (the reference builds one Ada-SnapKV cluster per attention layer, :1049): a
    process-wide buffer keyed by (H, device) would let two threads / streams running Ada-SnapKV on the same device
    overwrite each other's capacities between the poll and the read.  A cluster instance itself is single-caller state
    (``head_lens``, ``cu_klen`` ... are attributes the decode step mutates), so nothing is shared beyond it."""

    def __init__(self, H):
        self.t = torch.zeros(H, dtype=torch.int64, pin_memory=True)
        self.np = self.t.numpy()                 # shares the pinned memory
        self.ptr = self.t.data_ptr()
        self.H, self.seq = H, 0

    def next_seq(self):
        self.seq = self.seq % 0x3fffffff + 1
        return self.seq

    def arrived(self):
        """One look at the last word: has the budget kernel of the current sequence number written the capacities?"""
        return (int(self.np[self.H - 1]) >> 32) == self.seq

    def wait_words(self, device):
        """-> the H raw words once every one of them carries the current sequence number (word h = seq << 32 | ran_out << 31 |
        cap_h, so "all carry seq" is one min and one max over the list); ``self.exhausted`` = bit 31."""
        a, last, seq = self.np, self.H - 1, self.seq
        lo, hi = seq << 32, (seq + 1) << 32
        t_end = time.perf_counter() + 0.5
        spins = 0
        while True:
            if lo <= int(a[last]) < hi:          # the heads' stores leave together: look at one word, then check all of them
                vals = a.tolist()                # copied out before the next call of this instance can reuse the buffer
                mx = max(vals)
                if lo <= min(vals) and mx < hi:
                    break
            spins += 1
            if spins & 63 == 0:
                time.sleep(0)                    # hand the GIL to other host threads while the kernel runs
                if time.perf_counter() > t_end:  # a lost signal must not hang the host: fall back
                    torch.cuda.current_stream(device).synchronize()
                    vals = a.tolist()
                    mx = max(vals)
                    if not (lo <= min(vals) and mx < hi):
                        raise RuntimeError("pyramidkv_amd: the budget kernel did not report its head capacities")
                    break
        self.exhausted = bool(vals[0] & 0x80000000)
        self.max_word = mx               # the largest word (same upper half everywhere: the largest capacity)
        return vals

    def wait(self, device):
        """-> the H capacities; ``self.exhausted`` = bit 31 of the words (short lists: some head's list ran out)."""
        return [v & 0x7fffffff for v in self.wait_words(device)]


_ADA_TOPM_MAX = 4096      # longest per-head list taken from the top-k kernel; beyond it the rows are sorted completely


class _FlatPolicy:
    """Shared var-len metadata of AdaKV / HeadKV (reference :682-698)."""

    def _init_state(self):
        self.head_lens = None
        self.max_seqlen_k = 0
        self.klen_sum = 0
        self.cu_klen = 0
        self.cu_offset = None
        self.cu_headlens = None

    _CONST = {}     # (num_heads, device) -> the four read-only index vectors (the decode step only adds cu_offset to cu_klen)

    @property
    def head_capacity_last(self):
        """The head capacities of the last update_kv as Python ints (not a reference attribute; tests and the short-list
        protocol read it).  The fast path keeps the mirror's raw words and decodes them here, on demand."""
        w = self.__dict__.get("_cap_words")
        return [v & 0x7fffffff for v in w] if w is not None else self.__dict__.get("_caps")

    @head_capacity_last.setter
    def head_capacity_last(self, caps):
        self.__dict__["_cap_words"] = None
        self.__dict__["_caps"] = caps

    def _init_metadata(self, num_heads, head_lens, cu_klen, klen_sum, max_seqlen_k, device, cu_headlens=None):
        self.head_lens = head_lens                                                   # int32 [H]          :684
        self.klen_sum = klen_sum                                                     #                    :685
        self.max_seqlen_k = max_seqlen_k                                             #                    :686
        # inclusive prefix :687 (a separate tensor: the decode steps shift cu_klen in place, llama_model.py:2374)
        self.cu_headlens = cu_headlens if cu_headlens is not None else cu_klen[1:].clone()
        self.cu_klen = cu_klen                                                       # int32 [H+1]        :689-691
        key = (num_heads, str(device))
        const = _FlatPolicy._CONST.get(key)
        if const is None:            # built once per geometry: four tiny launches less on every prefill call
            ar = torch.arange(0, num_heads + 1, dtype=torch.int32, device=device)
            const = (torch.ones(num_heads, dtype=torch.int32, device=device), ar, ar.clone(), ar[1:].clone())
            _FlatPolicy._CONST[key] = const
        self.layer_qlens, self.cu_qlen, self.cu_offset, self.cu_head_offset = const   # :692, :694-698
        self.qlen_sum = num_heads

    def _scores(self, key_states, query_states):
        """calcul_attn_sore (:647-672): mean over the window rows, then pooling."""
        if self.pooling not in ('avgpool', 'maxpool'):
            raise ValueError('Pooling method not supported')
        return ops.score_window(query_states, key_states, self.window_size, self.pooling, self.kernel_size,
                                reduce="mean", scale_mode=_cfg.scale_mode,
                                kv_group=_unexpanded_group(key_states, query_states))

    accepts_unexpanded_kv = True      # K/V may arrive with H/g heads (before repeat_kv): every KV head is read once per group

    def _passthrough(self, key_states, value_states, num_heads, q_len, head_dim):
        g = num_heads // key_states.shape[1]
        key_states, value_states = _repeat_kv(key_states, g), _repeat_kv(value_states, g)
        dev = key_states.device
        head_lens = torch.full((num_heads,), q_len, dtype=torch.int32, device=dev)
        cu = torch.arange(0, num_heads + 1, dtype=torch.int32, device=dev) * q_len
        self._init_metadata(num_heads, head_lens, cu, q_len * num_heads, q_len, dev)  # :701
        return key_states.reshape(-1, head_dim), value_states.reshape(-1, head_dim)   # :703

    def _flat_from_capacity(self, key_states, value_states, sorted_idx, cap_dev, num_heads, caps_host=None, meta=None,
                            rows_bound=None, mirror=None):
        """Flat gather + metadata.  The boundary exposes klen_sum / max_seqlen_k as Python ints (:685-686), which needs the
        capacities on the host: HeadKV knows them already (``caps_host``), Ada-SnapKV reads them back (one host sync, as the
        reference's ``.item()`` at :718)."""
        head_lens, cu = meta[:2] if meta is not None else ops.ada_metadata(cap_dev, self.window_size)
        g = num_heads // key_states.shape[1]
        if caps_host is None and rows_bound is not None:
            # Ada-SnapKV on the fused path: the flat gather runs on the device-resident capacities BEFORE the one host
            # sync, into buffers sized by the bound sum_h cap_h <= H*base + H/2 (:719 rounds every head by < 1/2); the
            # read-back then only narrows the views.  Nothing but the sync latency is left between the kernels and the return.
            kf, vf = ops.gather_flat(key_states, value_states, sorted_idx, cap_dev, cu, self.window_size, rows_bound,
                                     max_cap=sorted_idx.shape[1], kv_group=g)
            caps = mirror.wait(key_states.device) if mirror is not None else _read_back(cap_dev)
            klen_sum = sum(caps) + num_heads * self.window_size
            max_cap = max(caps)
            kf, vf = kf[:klen_sum], vf[:klen_sum]
        else:
            caps = caps_host if caps_host is not None else _read_back(cap_dev)
            klen_sum = sum(caps) + num_heads * self.window_size
            max_cap = max(caps)
            kf, vf = ops.gather_flat(key_states, value_states, sorted_idx, cap_dev, cu, self.window_size, klen_sum,
                                     max_cap=max_cap, kv_group=g)
        self._init_metadata(num_heads, head_lens, cu, klen_sum, max_cap + self.window_size, key_states.device,
                            cu_headlens=meta[2] if meta is not None and len(meta) > 2 else None)
        self.head_capacity_last = caps
        return kf, vf


class _AdaRoute(enum.Enum):
    """How an Ada-SnapKV cluster (one per attention layer, :1049) gets from the scores to the head budgets."""
    LISTS = "lists"   # per-head candidate lists from the top-k kernel: short first, the call repeated with the full length when the
    #                   budget kernel reports a list that ran out at the threshold (pkv_ada_select)
    ROWS = "rows"     # budgets from selections / counts over the un-sorted rows (pkv_ada_budget_rows): what a run-out leaves
    #                   behind when H * base > 4096 (no longer list exists); also fp32 tensors and config.host_poll = 0


class _AdaState:
    """Everything an AdaKVCluster remembers between calls (round 6: one object instead of five ad-hoc attributes).

      route      LISTS until a short list of a LARGE budget (H * base > 4096) runs out, ROWS from then on - never back
      list_len   0, or the list length a run-out taught this layer (twice the largest share it saw): short lists start there
      cap_seen   ROWS route: the largest head capacity seen, which sizes the guess the selection is issued with before the sync
      prepared   the two prepared C calls of the LISTS route for the last operand layouts (_AdaPrepared) or None
      repeats    how often a call had to be repeated (a list ran out / a guess was too small) - read by tests and bench.py
      klen_last  `klen_sum` of the last prepared call: the flat outputs are narrowed to it BEFORE the host sync (the totals of one
                 layer differ between prompts only by the roundings of :719) and the views are kept when the capacities agree

    A call is: prepared hit -> done, or a run-out -> the general path with ``retry_full``; the general path picks the route,
    runs it, and leaves `prepared` behind for the next call.  Nothing else is kept on the cluster."""
    __slots__ = ("route", "list_len", "cap_seen", "prepared", "repeats", "klen_last")

    def __init__(self):
        self.route = _AdaRoute.LISTS
        self.list_len = 0
        self.cap_seen = 0
        self.prepared = None
        self.repeats = 0
        self.klen_last = 0


class _AdaPrepared:
    __slots__ = ("sig", "pa", "mirror", "m_use", "M")

    def __init__(self, sig, pa, mirror, m_use, M):
        self.sig, self.pa, self.mirror, self.m_use, self.M = sig, pa, mirror, m_use, M


class AdaKVCluster(_FlatPolicy):
    """reference pyramidkv_utils.py:622-757 (adapted there from FFY0/AdaKV)."""

    def __init__(self, window_size=32, kernel_size=7, pooling='maxpool', max_capacity_prompt=None, floor=None,
                 normalize=None, layer_idx=None, num_hidden_layers=None):
        self.window_size = window_size
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.base_capacity = max_capacity_prompt - window_size                       # :630
        self.floor_ratio = floor
        self.floor_capacity = int(self.base_capacity * self.floor_ratio)             # :632
        self.adaptive_capacity = self.base_capacity - self.floor_capacity
        self.num_hidden_layers = num_hidden_layers
        self.normalize = normalize
        self.layer_idx = layer_idx
        self.ada = _AdaState()
        self._init_state()

    def _fast_sig(self):
        return (self.window_size, self.pooling, self.kernel_size, self.base_capacity, self.floor_ratio, self.normalize,
                _cfg.host_poll, _cfg.ada_short_lists, self.ada.list_len)

    def update_kv(self, key_states, query_states, value_states):
        # fast path (round 5): the two C calls prepared by an earlier update_kv of this cluster with the same layouts, list length
        # and knobs (ops.PreparedAda) - between the capacities arriving on the host and the first kernel of the NEXT call lies
        # nothing but this check, one allocation and one foreign call (the host sync of the policy, :718, makes every
        # microsecond of host work here a microsecond of idle GPU)
        st = self.ada
        retry_full = False               # this call only: the prepared call below saw a list run out
        fast = st.prepared
        if fast is not None and fast.sig == self._fast_sig() and fast.pa.hit(query_states, key_states, value_states):
            pa, mirror = fast.pa, fast.mirror
            out = pa.run(query_states, key_states, value_states, mirror.ptr, mirror.next_seq())
            if out is not None:
                head_lens, cu, cuh, kf, vf = out
                num_heads, w = pa.H, self.window_size
                self._init_metadata(num_heads, head_lens, cu, 0, 0, pa.device, cu_headlens=cuh)    # everything but the two host ints
                # Round 6: host work that does not need the capacities happens HERE, while the kernels run - after the wait every
                # microsecond of it is idle GPU.  (a) The outputs narrowed to the last call's total (two view ops, ~2.4 us).
                # (b) The NEXT call's metadata buffer (its torch.empty sits in front of that call's first launch, ~1.6 us).
                # Only when the capacities have not arrived yet (a short prompt's kernels can be done before the host has issued
                # them all: then this work would sit between the arrival and the return).
                guess = 0
                if _cfg.ada_prewait and not mirror.arrived():
                    guess = st.klen_last
                    kg, vg = (kf[:guess], vf[:guess]) if guess else (None, None)
                    pa.spare()
                words = mirror.wait_words(pa.device)
                if not (mirror.exhausted and fast.m_use < fast.M):
                    # the words share their upper halves (sequence number, ran-out bit): sum and maximum of the capacities come
                    # from sum and maximum of the words (wait_words leaves the maximum behind: it validated the list with it)
                    flags = words[0] & ~0x7fffffff
                    klen_sum = sum(words) - num_heads * flags + num_heads * w
                    self.klen_sum = klen_sum                                             # :685
                    self.max_seqlen_k = (mirror.max_word & 0x7fffffff) + w               # :686
                    self._cap_words = words
                    if klen_sum == guess:
                        return kg, vg
                    st.klen_last = klen_sum
                    return kf[:klen_sum], vf[:klen_sum]
                st.prepared = None                                   # a list ran out: the general path repeats with the full length
                st.repeats += 1
                retry_full = True
        bsz, num_heads, q_len, head_dim = query_states.shape
        L = q_len - self.window_size
        if self.base_capacity > L:                                                   # :700
            return self._passthrough(key_states, value_states, num_heads, q_len, head_dim)
        assert bsz == 1                                                              # :724
        key_states, value_states, _ = _fit_group(key_states, value_states, _unexpanded_group(key_states, query_states), self.window_size)
        # :706 sorts every row completely; what :709-757 consume of that order is bounded: one head can receive at most
        # H*base entries of the global top-(H*base) (:712-717), so the first M = min(L, H*base) entries per head decide the
        # budgets AND hold every index the gather takes (cap_h <= M).  They come from the top-k kernel (no full sort);
        # score -> top-M -> budgets -> metadata is ONE C call.
        M = min(L, num_heads * self.base_capacity)
        # The list path serves lists up to _ADA_TOPM_MAX entries.  M itself fits for small budgets (H*base <= 4096); for large ones
        # (budget 2048: M is the whole row) it still applies with SHORT lists of 2 x base entries per head (round 5) - a head of a
        # real prompt stays below twice its base budget - and a list that runs out sends this layer to the un-sorted-rows
        # path below for good (there is no longer list to repeat with).
        full_ok = M <= _ADA_TOPM_MAX
        short_big = (not full_ok and _cfg.host_poll and _cfg.ada_short_lists > 0 and 2 * self.base_capacity <= _ADA_TOPM_MAX
                     and st.route is _AdaRoute.LISTS)
        ran_out = retry_full
        if ran_out and not full_ok:
            st.route, short_big = _AdaRoute.ROWS, False
        if (full_ok or short_big) and key_states.dtype != torch.float32:      # fp32 tensors take the un-sorted-rows path below
            if self.pooling not in ('avgpool', 'maxpool'):
                raise ValueError('Pooling method not supported')
            mirror = None
            if _cfg.host_poll:
                mirror = getattr(self, "_mirror", None)
                if mirror is None or mirror.H != num_heads:
                    mirror = self._mirror = _HostMirror(num_heads)
            # The lists start SHORT (round 4; round 5: config.ada_short_lists x base, at least 512 entries - what the top-k
            # kernel's small-k path takes - instead of M: no head of a real prompt comes near H x its base budget).  The budgets
            # are exact unless the kernel reports a head whose list ran out (the threshold sits below the list's last entry);
            # then the call is repeated with the full M, and the cluster - one per layer, :1049 - remembers twice the largest
            # share it has seen from then on.
            def short_len():
                if not full_ok:          # large budgets: 2.5 x base, what the list path can hold at most
                    return min(_ADA_TOPM_MAX, max(2 * self.base_capacity, (5 * self.base_capacity) // 2, 512))
                return min(M, max(st.list_len, _cfg.ada_short_lists * self.base_capacity, 512))
            m_use = M
            if mirror is not None and _cfg.ada_short_lists > 0:
                m_use = short_len()
            bound = num_heads * self.base_capacity + num_heads + num_heads * self.window_size
            gq = _unexpanded_group(key_states, query_states)
            if ran_out:
                m_use = M
            while True:
                sorted_idx, cap, head_lens, cu, cuh = ops.ada_select(
                    query_states, key_states, self.window_size, self.pooling, self.kernel_size, m_use, self.base_capacity,
                    self.floor_ratio, bool(self.normalize), scale_mode=_cfg.scale_mode, kv_group=gq,   # :647-672, :706-719, :682-691
                    host_mirror=mirror.t if mirror is not None else None, host_seq=mirror.next_seq() if mirror is not None else 0)
                out = self._flat_from_capacity(key_states, value_states, sorted_idx, cap, num_heads, meta=(head_lens, cu, cuh),
                                               rows_bound=bound, mirror=mirror)
                if m_use == M or not mirror.exhausted:
                    if ran_out:      # remembered from now on: this layer's heads want longer lists than the default
                        st.list_len = max(st.list_len, min(M, 2 * max(self.head_capacity_last)))
                    if mirror is not None:                           # the next call with these layouts takes the fast path above
                        m_next = short_len() if _cfg.ada_short_lists > 0 else M
                        pa = ops.prepare_ada(query_states, key_states, value_states, self.window_size, self.pooling, self.kernel_size,
                                             m_next, self.base_capacity, self.floor_ratio, bool(self.normalize), _cfg.scale_mode, gq, bound)
                        st.prepared = _AdaPrepared(self._fast_sig(), pa, mirror, m_next, M) if pa is not None else None
                    return out
                st.repeats += 1
                if not full_ok:                                      # large budget: no longer list exists - the un-sorted rows from now on
                    st.route = _AdaRoute.ROWS
                    break
                ran_out, m_use = True, M
        # H*base > 4096 (budget 2048: M is the whole row, a top-M list would be a full sort).  What :706-719 consume of the
        # order are selections and counts: the budgets come from histograms over the un-sorted rows (pkv_ada_budget_rows),
        # then every head's first cap_h entries of the canonical order from one top-k launch with per-head k.  Rows or
        # capacities beyond one top-k workgroup keep the complete sort.
        attn_score = self._scores(key_states, query_states)[0]                       # [H, L]   :647-672
        if key_states.dtype == torch.float32 and L > 32768:
            raise ValueError("Ada-SnapKV on fp32 tensors: rows up to 32768 past tokens")
        if L <= 65536:
            mirror = None
            if _cfg.host_poll:
                mirror = getattr(self, "_mirror", None)
                if mirror is None or mirror.H != num_heads:
                    mirror = self._mirror = _HostMirror(num_heads)
            cap, head_lens, cu, cuh = ops.ada_budget_rows(
                attn_score, self.base_capacity, self.floor_ratio, bool(self.normalize), self.window_size,   # :709-719, :682-691
                host_mirror=mirror.t if mirror is not None else None, host_seq=mirror.next_seq() if mirror is not None else 0)
            # Round 5: selection and flat gather are issued BEFORE the host sync, on the device-resident capacities, with a
            # guessed upper bound of the largest head capacity (twice the base budget, or twice the largest this layer has seen)
            # and outputs sized by the bound sum_h cap_h <= H*base + H/2; the read-back then only narrows the views.  Round 4
            # waited for the capacities first: the GPU idled for the host's round trip in the middle of every call (~10 us of
            # ~120 at budget 2048).  A capacity beyond the guess (checked after the wait) repeats selection + gather the old way.
            guess = min(L, max(2 * self.base_capacity, 2 * st.cap_seen))
            if mirror is not None and key_states.dtype != torch.float32 and ops.topk_fits(num_heads, L, guess):
                g = num_heads // key_states.shape[1]
                bound = num_heads * self.base_capacity + num_heads + num_heads * self.window_size
                top_idx = ops.topk(attn_score, guess, k_per_row=cap)
                kf, vf = ops.gather_flat(key_states, value_states, top_idx, cap, cu, self.window_size, bound, max_cap=guess, kv_group=g)
                caps = mirror.wait(key_states.device)                                    # the one host sync (:718)
                if max(caps) <= guess:
                    klen_sum = sum(caps) + num_heads * self.window_size
                    self._init_metadata(num_heads, head_lens, cu, klen_sum, max(caps) + self.window_size, key_states.device, cu_headlens=cuh)
                    self.head_capacity_last = caps
                    return kf[:klen_sum], vf[:klen_sum]
                st.cap_seen = max(caps)                                                  # the guess was too small: the exact path below
                st.repeats += 1
            else:
                caps = mirror.wait(key_states.device) if mirror is not None else _read_back(cap)   # the one host sync (:718)
            kmax = max(1, max(caps))
            if key_states.dtype == torch.float32 or ops.topk_fits(num_heads, L, kmax):
                try:
                    top_idx = ops.topk(attn_score, kmax, k_per_row=cap)
                except ValueError:
                    if key_states.dtype != torch.float32:
                        raise
                    raise ValueError("Ada-SnapKV on fp32 tensors: head capacities up to 4096 past tokens") from None
            else:
                # rows / capacities beyond one top-k workgroup: the complete order once, the capacities already computed above
                # (no second budget pass, no long-row top-k scratch for a call that cannot take k_per_row)
                top_idx, _ = ops.sort_rows(attn_score, want_values=False)
            return self._flat_from_capacity(key_states, value_states, top_idx, cap, num_heads, caps_host=caps,
                                            meta=(head_lens, cu, cuh))
        sorted_idx, sorted_val = ops.sort_rows(attn_score)                           # :706
        cap = ops.ada_budget(sorted_val, self.base_capacity, self.floor_ratio, bool(self.normalize))  # :709-719
        return self._flat_from_capacity(key_states, value_states, sorted_idx, cap, num_heads)


class HeadKVCluster(_FlatPolicy):
    """reference pyramidkv_utils.py:760-878: AdaKV gather with precomputed head_capacity[layer][head]."""

    def __init__(self, window_size=32, kernel_size=7, pooling='maxpool', max_capacity_prompt=None, layer_idx=None,
                 num_hidden_layers=None, head_capacity=None):
        self.window_size = window_size
        self.kernel_size = kernel_size
        self.pooling = pooling
        self.base_capacity = max_capacity_prompt - window_size
        self.head_adaptive_capacity = head_capacity
        self.num_hidden_layers = num_hidden_layers
        self.layer_idx = layer_idx
        self._init_state()

    def update_kv(self, key_states, query_states, value_states):
        bsz, num_heads, q_len, head_dim = query_states.shape
        L = q_len - self.window_size
        if self.base_capacity > L:                                                   # :834
            return self._passthrough(key_states, value_states, num_heads, q_len, head_dim)
        assert bsz == 1                                                              # :845
        key_states, value_states, _ = _fit_group(key_states, value_states, _unexpanded_group(key_states, query_states), self.window_size)
        caps = [min(int(self.head_adaptive_capacity[self.layer_idx][h]), L) for h in range(num_heads)]  # :855 slice
        key = (tuple(caps), str(key_states.device))
        if getattr(self, "_cap_key", None) != key:               # the per-layer capacities are constants: upload them once
            self._cap_key, self._cap_dev = key, torch.tensor(caps, dtype=torch.int32, device=key_states.device)
        cap = self._cap_dev
        if key_states.dtype == torch.float32:
            # fp32 tensors (round 3): fp32 window scores -> per-head top-cap_h (32-bit radix select, k <= 4096) -> flat gather
            attn_score = self._scores(key_states, query_states)[0]
            sorted_idx = ops.topk(attn_score, max(1, max(caps)), k_per_row=cap)      # ValueError beyond 4096 entries per head
            return self._flat_from_capacity(key_states, value_states, sorted_idx, cap, num_heads, caps_host=caps)
        # :840 sorts every row completely and :855 keeps the first cap_h entries: a top-k with k = max_h cap_h holds them all
        if max(caps) <= _ADA_TOPM_MAX:
            if self.pooling not in ('avgpool', 'maxpool'):
                raise ValueError('Pooling method not supported')
            kmax = max(1, max(caps))
            klen_sum = sum(caps) + num_heads * self.window_size
            gq = _unexpanded_group(key_states, query_states)
            # round 5: both C calls prepared per (layouts, capacities, knobs) - HeadKV has no host sync, so its call is bound by
            # whichever of host issue and device is slower (ops.PreparedAda with the capacities given)
            sig = (self._cap_key, self.window_size, self.pooling, self.kernel_size)
            fast = self.__dict__.get("_fast")
            if fast is not None and fast[0] == sig and fast[1].hit(query_states, key_states, value_states):
                out = fast[1].run(query_states, key_states, value_states, None, 0, given_ptr=cap.data_ptr())
                if out is not None:
                    head_lens, cu, cuh, kf, vf = out
                    self._init_metadata(num_heads, head_lens, cu, klen_sum, kmax + self.window_size, key_states.device, cu_headlens=cuh)
                    self.head_capacity_last = caps
                    return kf, vf
            else:
                pa = ops.prepare_ada(query_states, key_states, value_states, self.window_size, self.pooling, self.kernel_size, kmax,
                                     0, 0.0, False, _cfg.scale_mode, gq, klen_sum)
                self._fast = (sig, pa) if pa is not None else None
            sorted_idx, _, head_lens, cu, cuh = ops.ada_select(
                query_states, key_states, self.window_size, self.pooling, self.kernel_size, kmax,
                given_capacity=cap, scale_mode=_cfg.scale_mode, kv_group=gq)
            return self._flat_from_capacity(key_states, value_states, sorted_idx, cap, num_heads, caps_host=caps,
                                            meta=(head_lens, cu, cuh))
        attn_score = self._scores(key_states, query_states)[0]
        try:                                                                         # per-head top-cap_h instead of the sort of :840
            sorted_idx = ops.topk(attn_score, max(caps), k_per_row=cap)
        except ValueError:                                                           # beyond one top-k workgroup's LDS
            sorted_idx, _ = ops.sort_rows(attn_score, want_values=False)             # :840
        return self._flat_from_capacity(key_states, value_states, sorted_idx, cap, num_heads, caps_host=caps)


# ------------------------------------------------------------------------------------------------
# init_* factories: the plugin seam read by the patched attention forwards (reference :880-1085)
# ------------------------------------------------------------------------------------------------
def _default(config, name, value):
    if not hasattr(config, name):
        setattr(config, name, value)


def _rebuilt(self, cluster):
    """The reference rebuilds the dense clusters on every forward (:894,:918,:1003,:1025).  The prepared call of the cluster
    being replaced (ops.PreparedCompress: descriptor + scratch size for the layouts it saw) moves to the new one - it is
    keyed by every knob that could have changed, so a stale one is simply never hit."""
    old = getattr(self, "kv_cluster", None)
    if old is not None and type(old) is type(cluster):
        prep = old.__dict__.get("_prep")
        if prep is not None:
            cluster._prep = prep
    self.kv_cluster = cluster


def init_pyramidkv(self, num_hidden_layers):
    """reference :880-902 (cluster rebuilt on every call, as there)."""
    if not hasattr(self, "kv_cluster"):
        _default(self.config, 'window_size', 32)
        _default(self.config, 'max_capacity_prompt', 2048)
        _default(self.config, 'kernel_size', 5)
        _default(self.config, 'pooling', 'avgpool')
        _default(self.config, 'merge', None)
    _rebuilt(self, PyramidKVCluster(
        num_hidden_layers=num_hidden_layers, layer_idx=self.layer_idx, window_size=self.config.window_size,
        max_capacity_prompt=self.config.max_capacity_prompt, kernel_size=self.config.kernel_size,
        pooling=self.config.pooling, merge=self.config.merge))


def init_snapkv(self):
    """reference :904-924."""
    if not hasattr(self, "kv_cluster"):
        _default(self.config, 'window_size', 32)
        _default(self.config, 'max_capacity_prompt', 4096)
        _default(self.config, 'kernel_size', 5)
        _default(self.config, 'pooling', 'avgpool')
        _default(self.config, 'merge', None)
    _rebuilt(self, SnapKVCluster(
        window_size=self.config.window_size, max_capacity_prompt=self.config.max_capacity_prompt,
        kernel_size=self.config.kernel_size, pooling=self.config.pooling, merge=self.config.merge))


def init_H2O(self):
    """reference :990-1009."""
    if not hasattr(self, "kv_cluster"):
        _default(self.config, 'window_size', 32)
        _default(self.config, 'max_capacity_prompt', 2048)
        _default(self.config, 'kernel_size', 5)
        _default(self.config, 'pooling', 'avgpool')
        _default(self.config, 'merge', None)
    _rebuilt(self, H2OKVCluster(
        window_size=self.config.window_size, max_capacity_prompt=self.config.max_capacity_prompt,
        kernel_size=self.config.kernel_size, pooling=self.config.pooling, merge=self.config.merge))


def init_StreamingLLM(self):
    """reference :1011-1031."""
    if not hasattr(self, "kv_cluster"):
        _default(self.config, 'window_size', 32)
        _default(self.config, 'max_capacity_prompt', 2048)
        _default(self.config, 'kernel_size', 5)
        _default(self.config, 'pooling', 'avgpool')
        _default(self.config, 'merge', None)
    _rebuilt(self, StreamingLLMKVCluster(
        window_size=self.config.window_size, max_capacity_prompt=self.config.max_capacity_prompt,
        kernel_size=self.config.kernel_size, pooling=self.config.pooling, merge=self.config.merge))


def init_adakv(self):
    """reference :1033-1059 (built once; reads config.floor like the reference does at :1057)."""
    if not hasattr(self, "kv_cluster"):
        _default(self.config, 'window_size', 32)
        _default(self.config, 'max_capacity_prompt', 2048)
        _default(self.config, 'kernel_size', 5)
        _default(self.config, 'pooling', 'maxpool')
        _default(self.config, 'floor_ratio', 0.2)
        _default(self.config, 'normalize', True)
    if not hasattr(self, "kv_cluster"):
        self.kv_cluster = AdaKVCluster(
            num_hidden_layers=self.config.num_hidden_layers, layer_idx=self.layer_idx,
            window_size=self.config.window_size, max_capacity_prompt=self.config.max_capacity_prompt,
            kernel_size=self.config.kernel_size, pooling=self.config.pooling, floor=self.config.floor,
            normalize=self.config.normalize)


def init_headkv(self):
    """reference :1062-1085."""
    if not hasattr(self, "kv_cluster"):
        _default(self.config, 'window_size', 32)
        _default(self.config, 'max_capacity_prompt', 2048)
        _default(self.config, 'kernel_size', 5)
        _default(self.config, 'pooling', 'maxpool')
        if not hasattr(self.config, 'head_capacity'):
            raise ValueError("Must have head_capacity")                              # :1073
    if not hasattr(self, "kv_cluster"):
        self.kv_cluster = HeadKVCluster(
            num_hidden_layers=self.config.num_hidden_layers, layer_idx=self.layer_idx,
            window_size=self.config.window_size, max_capacity_prompt=self.config.max_capacity_prompt,
            kernel_size=self.config.kernel_size, pooling=self.config.pooling,
            head_capacity=self.config.head_capacity)


def repeat_kv(hidden_states: torch.Tensor, n_rep: int) -> torch.Tensor:
    """reference pyramidkv_utils.py:108-117 (same name and behaviour: n_rep == 1 returns the input object)."""
    return _repeat_kv(hidden_states, n_rep)


def _out_of_scope(name, where):
    def init(self, *args, **kwargs):
        raise NotImplementedError(
            f"{name} (reference pyramidkv_utils.py:{where}) is outside the scope of pyramidkv_amd (SURVEY.md section 2: "
            "CAM / L2Norm / ThinK are not on the update_kv hot path); the name exists so that the reference's "
            "llama_model.py / mistral_model.py import this module unchanged")
    init.__name__ = name
    init.__doc__ = f"Placeholder for the reference's {name} (:{where}); raises NotImplementedError when called."
    return init


# llama_model.py:16, llama_model_think.py:16 and mistral_model.py:19 import these three from this module.
init_CAM = _out_of_scope("init_CAM", "970-988")
init_l2norm = _out_of_scope("init_l2norm", "954-968")
init_think = _out_of_scope("init_think", "926-952")


def headkv_head_capacity(head_scores, num_hidden_layers, num_attention_heads, max_capacity_prompts, head_beta=1.01):
    """Per-(layer, head) HeadKV budgets from a retrieval/reasoning head-score table (reference runner,
    run_longbench.py:225-234; the score files are data/heads_score/*.json: ``{"layer-head": [scores...]}``).

    ``head_scores`` is that mapping (insertion order = layer-major, as the runner iterates ``.items()``) or a
    sequence of per-head score lists.  Arithmetic is IEEE double exactly as the runner's numpy/torch mix:
    mean per head -> / sequential sum -> x pool + min_num -> round half-to-even -> int32
    ``[num_hidden_layers, num_attention_heads]`` (what ``config.head_capacity`` holds, :234)."""
    import numpy as np
    rows = list(head_scores.values()) if hasattr(head_scores, "values") else list(head_scores)
    if len(rows) != num_hidden_layers * num_attention_heads:
        raise ValueError(f"head score table has {len(rows)} heads, model has {num_hidden_layers}x{num_attention_heads}")
    means = [np.mean(r) for r in rows]                                           # :228
    share = np.asarray(means, dtype=np.float64) / sum(means)                      # :229 (builtin sum: sequential)
    pool = (max_capacity_prompts // head_beta) * num_hidden_layers * num_attention_heads   # :231
    min_num = max_capacity_prompts - max_capacity_prompts // head_beta           # :232
    cap = np.round(share * pool + min_num).astype(np.int32)                      # :233
    return torch.from_numpy(cap.reshape(num_hidden_layers, num_attention_heads))
