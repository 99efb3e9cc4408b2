"""Stage-level host wrappers over the C ABI (include/pkv.h).

PyTorch is used for device memory and the current stream only; all arithmetic happens in libpkv's
HIP kernels.  Every function raises if the tensors are not on a HIP device - there is no CPU path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import _native as N
from . import config as _cfg

_WS: Dict[Tuple[int, int], torch.Tensor] = {}
# Superseded workspaces stay alive while a HIP graph captured earlier may still have their address baked in.  Growth is
# geometric (every new buffer is >= 2x the one it supersedes), so per (device, stream) the retired buffers sum to less than
# the live one and number ~log2(largest / 1 MB); they are never released behind the caller's back (release_workspaces()
# drops everything, e.g. between serving sessions once no captured graph is replayed any more).
_WS_RETIRED: Dict[Tuple[int, int], list] = {}


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pyramidkv_amd runs on MI355X (HIP) tensors only; got a CPU tensor. "
                               "There is no CPU fallback - use the reference (or oracle/) on CPU.")


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    """[B,H,S,D] with D contiguous and 16-byte aligned rows; anything else is materialised."""
    if t.is_contiguous() and not (t.shape[-1] & 7) and not (t.data_ptr() & 15):      # the common case, one cheap test
        return t
    if t.stride(-1) != 1 or any(s % 8 for s in t.stride()[:-1]) or t.data_ptr() % 16:
        t = t.contiguous()
    return t


class _DeviceGuard:
    """``with torch.cuda.device(d)`` without its cost when ``d`` is already current (the usual case: ~4 us per call saved;
    at S <= 8192 one update_kv is bound by the host's time to issue it, bench.py `sweep[].host_us`)."""
    __slots__ = ("idx", "prev")

    def __init__(self, device: torch.device):
        self.idx = device.index
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if self.idx is not None and cur != self.idx:
            torch.cuda.set_device(self.idx)
            self.prev = cur
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
            self.prev = -1
        return False


def _workspace_and_stream(nbytes: int, device: torch.device):
    """(scratch, raw stream handle) of the CURRENT stream of ``device`` - one ``current_stream`` query for both."""
    st = torch.cuda.current_stream(device).cuda_stream
    key = (device.index if device.index is not None else torch.cuda.current_device(), st)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _WS_RETIRED.setdefault(key, []).append(ws)     # never trimmed: a captured graph may still replay with this address
        ws = torch.empty(max(nbytes, 2 * ws.numel() if ws is not None else 0, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws, st


def workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    """Scratch per (device, stream), grown geometrically and NEVER freed or shrunk: the buffer a larger request
    supersedes is kept alive (``_WS_RETIRED``), because a HIP graph captured while it was current replays with its
    address.  Capture a graph only after one eager call of the same shape on the same stream (the workspace then exists
    outside the graph's private pool); see INTEGRATION.md."""
    return _workspace_and_stream(nbytes, device)[0]


def release_workspaces() -> None:
    """Drop every cached and retired workspace (all devices, all streams).  Only safe when no HIP graph captured around
    ``update_kv`` will be replayed again: a graph replays with the workspace address it was captured with."""
    _WS.clear()
    _WS_RETIRED.clear()


def make_desc(q: Optional[torch.Tensor], k: torch.Tensor, v: Optional[torch.Tensor], window: int,
              pooling=None, kernel_size: int = 1, reduce: str = "sum", scale_mode: str = "div",
              topk: int = 0, kv_group: int = 1, num_heads: Optional[int] = None) -> N.PkvDesc:
    B, Hk, S, D = k.shape
    H = num_heads if num_heads is not None else (q.shape[1] if q is not None else Hk * kv_group)
    d = N.PkvDesc()
    d.dtype = N.dtype_code(k.dtype)
    d.B, d.H, d.S, d.D = B, H, S, D
    d.kv_group = kv_group
    for name, t in (("q_stride", q), ("k_stride", k), ("v_stride", v)):
        arr = getattr(d, name)
        st = t.stride()[:3] if t is not None else k.stride()[:3]
        for i in range(3):
            arr[i] = st[i]
    d.window = window
    d.pool_kind = N.POOL[pooling]
    d.pool_kernel = kernel_size if pooling not in (None, "none") else 1
    d.reduce = N.REDUCE[reduce]
    d.scale_mode = N.SCALE[scale_mode]
    d.topk = topk
    d.tie_order = N.TIE[_cfg.tie_order]
    return d


_DESC_CACHE: dict = {}


def _scoring_desc(q, k, v, window, pooling, kernel_size, reduce, scale_mode, topk, kv_group):
    """(descriptor, pkv_workspace_bytes(descriptor)) for a scoring call, cached by everything the descriptor is made of:
    filling the ctypes struct and asking the library for the scratch size are ~5 us of host time per call otherwise."""
    key = (k.dtype, q.shape, k.shape, q.stride(), k.stride(), v.stride() if v is not None else None, window, pooling, kernel_size,
           reduce, scale_mode, topk, kv_group, _cfg.tie_order)
    hit = _DESC_CACHE.get(key)
    if hit is None:
        if len(_DESC_CACHE) > 4096:
            _DESC_CACHE.clear()
        d = make_desc(q, k, v, window, pooling, kernel_size, reduce, scale_mode, topk, kv_group)
        hit = _DESC_CACHE[key] = (d, N.lib.pkv_workspace_bytes(d))
    return hit


def _lp(L: int) -> int:
    return (L + 7) // 8 * 8


def score_window(q, k, window: int, pooling=None, kernel_size: int = 1, reduce: str = "sum",
                 scale_mode: str = "div", kv_group: int = 1) -> torch.Tensor:
    """pyramidkv_utils.py:317-333.  Returns [B,H,S-w] (a view of a [B,H,Lp] buffer)."""
    _require_gpu(q, k)
    q, k = _rowmajor(q), _rowmajor(k)
    B, H, S, _ = q.shape
    L = S - window
    with torch.cuda.device(k.device):
        d = make_desc(q, k, None, window, pooling, kernel_size, reduce, scale_mode, 0, kv_group)
        nb = N.lib.pkv_workspace_bytes(d)
        ws = workspace(nb, k.device)
        out = torch.empty(B, H, _lp(L), dtype=k.dtype, device=k.device)
        N.check(N.lib.pkv_score_window(d, q.data_ptr(), k.data_ptr(), out.data_ptr(), _lp(L), ws.data_ptr(),
                                       ws.numel(), N.stream_ptr()), "pkv_score_window")
    return out[..., :L]


def score_h2o(q, k, window: int, scale_mode: str = "div", kv_group: int = 1) -> torch.Tensor:
    """pyramidkv_utils.py:544-554."""
    _require_gpu(q, k)
    q, k = _rowmajor(q), _rowmajor(k)
    B, H, S, _ = q.shape
    L = S - window
    with torch.cuda.device(k.device):
        d = make_desc(q, k, None, window, None, 1, "sum", scale_mode, 0, kv_group)
        ws = workspace(N.lib.pkv_workspace_bytes(d), k.device)
        out = torch.empty(B, H, _lp(L), dtype=k.dtype, device=k.device)
        N.check(N.lib.pkv_score_h2o(d, q.data_ptr(), k.data_ptr(), out.data_ptr(), _lp(L), ws.data_ptr(),
                                    ws.numel(), N.stream_ptr()), "pkv_score_h2o")
    return out[..., :L]


def topk_fits(rows: int, L: int, k: int) -> bool:
    """True when a row of L scores and k winners fits ONE top-k workgroup (no long-row scratch): the only form that takes
    ``k_per_row``."""
    return N.lib.pkv_topk_workspace_bytes(rows, L, k) == 0


def _one_topk_workgroup(L: int, k: int) -> bool:
    """The LDS budget of topk_kernel (csrc/pkv_topk.hip topk_lds_bytes): the row as 16-bit keys + the
    selection list (rounded up to a power of two beyond 4096 entries) in 160 KB.  k > 16 384 never fits."""
    lw = max(512, -(-(-(-L // 16)) // 512) * 512)
    kpad = (k + 15) & ~15 if k <= 4096 else 1 << (k - 1).bit_length()
    return 16 * lw <= 65536 and 2 * 16 * lw + 4 * max(kpad, 8192) + 4 * 256 + 4 * 64 <= 160 * 1024


def _needs_full_sort(L: int, k: int, dtype) -> bool:
    """Budgets beyond one top-k workgroup on rows the sort kernel holds (k in the tens of thousands at S <= 32k: nothing the
    runners use, but the reference takes any k <= L): the selection is the first k entries of the complete canonical order."""
    return k > 4096 and dtype != torch.float32 and L <= 32768 and not _one_topk_workgroup(L, k)


def topk(scores: torch.Tensor, k: int, k_per_row: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pyramidkv_utils.py:334.  scores [..., L] (16-bit) -> int32 [..., k], (value desc, index asc).  ``k_per_row`` (device
    int32, one entry per row, each <= k): row r gets only its first k_per_row[r] entries (the rest of the row is unspecified) -
    the per-head capacities of Ada-SnapKV / HeadKV."""
    _require_gpu(scores)
    if scores.stride(-1) != 1:
        scores = scores.contiguous()
    L = scores.shape[-1]
    lead = scores.shape[:-1]
    rows = int(math.prod(lead)) if len(lead) else 1
    if scores.dim() > 1:
        stride = scores.stride(-2)
        flat_ok = all(scores.stride(i) == scores.stride(i + 1) * scores.shape[i + 1] for i in range(scores.dim() - 2))
        if not flat_ok:
            scores = scores.contiguous()
            stride = L
    else:
        stride = L
    out = torch.empty(*lead, k, dtype=torch.int32, device=scores.device)
    with torch.cuda.device(scores.device):
        nb = N.lib.pkv_topk_workspace_bytes(rows, L, k)          # > 0 only for rows beyond one workgroup's LDS
        kpr = k_per_row.data_ptr() if k_per_row is not None else None
        if nb:
            ws = workspace(nb, scores.device)
            N.check(N.lib.pkv_topk_ws(N.dtype_code(scores.dtype), rows, L, k, scores.data_ptr(), stride, kpr,
                                      out.data_ptr(), k, ws.data_ptr(), ws.numel(), N.stream_ptr()), "pkv_topk_ws")
        else:
            N.check(N.lib.pkv_topk(N.dtype_code(scores.dtype), rows, L, k, scores.data_ptr(), stride, kpr,
                                   out.data_ptr(), k, N.stream_ptr()), "pkv_topk")
    return out


def gather_compact(k, v, idx: torch.Tensor, window: int, kv_group: int = 1):
    """pyramidkv_utils.py:335,341-346.  idx int32 [B,H,n] -> (K_c, V_c) [B,H,n+w,D]."""
    _require_gpu(k, v, idx)
    k, v = _rowmajor(k), _rowmajor(v)
    B, H, n = idx.shape
    idx = idx.to(torch.int32).contiguous()
    D = k.shape[-1]
    with torch.cuda.device(k.device):
        d = make_desc(None, k, v, window, topk=n, kv_group=kv_group, num_heads=H)
        ko = torch.empty(B, H, n + window, D, dtype=k.dtype, device=k.device)
        vo = torch.empty_like(ko)
        N.check(N.lib.pkv_gather_compact(d, k.data_ptr(), v.data_ptr(), idx.data_ptr(), n, ko.data_ptr(),
                                         vo.data_ptr(), N.stream_ptr()), "pkv_gather_compact")
    return ko, vo


def gather_streaming(k, v, n_sink: int, window: int):
    """pyramidkv_utils.py:607-620."""
    _require_gpu(k, v)
    k, v = _rowmajor(k), _rowmajor(v)
    B, H, S, D = k.shape
    with torch.cuda.device(k.device):
        d = make_desc(None, k, v, window, topk=n_sink, num_heads=H)
        ko = torch.empty(B, H, n_sink + window, D, dtype=k.dtype, device=k.device)
        vo = torch.empty_like(ko)
        N.check(N.lib.pkv_gather_streaming(d, k.data_ptr(), v.data_ptr(), ko.data_ptr(), vo.data_ptr(),
                                           N.stream_ptr()), "pkv_gather_streaming")
    return ko, vo


def compress(q, k, v, window: int, topk_k: int, pooling, kernel_size: int, scale_mode: str = "div",
             kv_group: int = 1, h2o: bool = False, return_indices: bool = False, idx_out: Optional[torch.Tensor] = None):
    """Fused score -> top-k -> gather (pyramidkv_utils.py:317-346 / :544-575): one C call.  ``idx_out`` (contiguous int32
    [B,H,k], e.g. a slice of a per-prefill buffer that is all-gathered once) receives the indices instead of a new tensor."""
    _require_gpu(q, k, v)
    q, k, v = _rowmajor(q), _rowmajor(k), _rowmajor(v)
    B, H, S, D = q.shape
    if _needs_full_sort(S - window, topk_k, q.dtype):
        idx = select(q, k, window, topk_k, pooling, kernel_size, scale_mode, kv_group, h2o)
        ko, vo = gather_compact(k, v, idx, window, kv_group)
        if idx_out is not None:
            idx_out.view(B, H, topk_k).copy_(idx)
            return ko, vo, idx_out
        return (ko, vo, idx) if return_indices else (ko, vo)
    with _DeviceGuard(k.device):
        d, nb = _scoring_desc(q, k, v, window, None if h2o else pooling, kernel_size, "sum", scale_mode, topk_k, kv_group)
        ws, st = _workspace_and_stream(nb, k.device)
        ko = torch.empty(B, H, topk_k + window, D, dtype=k.dtype, device=k.device)
        vo = torch.empty_like(ko)
        if idx_out is not None:
            if idx_out.dtype != torch.int32 or not idx_out.is_contiguous() or idx_out.numel() != B * H * topk_k:
                raise ValueError("idx_out must be a contiguous int32 tensor of B*H*k elements")
            idx, return_indices = idx_out, True
        else:
            idx = torch.empty(B, H, topk_k, dtype=torch.int32, device=k.device) if return_indices else None
        fn = N.lib.pkv_compress_h2o if h2o else N.lib.pkv_compress
        rc = fn(d, q.data_ptr(), k.data_ptr(), v.data_ptr(), ko.data_ptr(), vo.data_ptr(),
                idx.data_ptr() if idx is not None else None, ws.data_ptr(), ws.numel(), st)
        if rc:
            N.check(rc, "pkv_compress")
    return (ko, vo, idx) if return_indices else (ko, vo)


# ---- prepared calls (round 5): everything of one update_kv that does not change from call to call ----------------------
# At S <= 8192 one dense update_kv is bound by the HOST's time to issue it (bench.py `sweep[].host_us`).  A cluster therefore
# keeps, per (shapes, strides, dtype, device, budget, knobs), the filled descriptor, its byref object, the scratch size and the
# bound entry point; a call is then: one signature comparison, the raw current stream, two torch.empty, five data_ptr() and
# ONE foreign call.  Anything unusual (operands that need a copy, budgets beyond one top-k workgroup, another current
# device) is not prepared and takes the general path above.
# Two private torch entry points make the prepared calls cheap (the raw handle of the current stream, the current device index -
# ~1 us each instead of ~4 through the public objects).  They are resolved ONCE here; a torch build without them gets the public
# API instead (same results, a few microseconds more per call) - never an AttributeError inside update_kv.
_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)
if _get_raw_stream is None:
    def _get_raw_stream(dev_index: int) -> int:
        return torch.cuda.current_stream(dev_index).cuda_stream
if _get_device is None:
    _get_device = torch.cuda.current_device


def _raw_stream(dev_index: int) -> int:
    return _get_raw_stream(dev_index)


class PreparedCompress:
    """pkv_compress / pkv_compress_h2o for fixed shapes: built by ``prepare_compress`` from the tensors of a first call."""
    __slots__ = ("qs", "ks", "vs", "qst", "kst", "vst", "dtype", "device", "dev_index", "desc", "dref", "nb", "fn", "what",
                 "out_shape", "nidx", "knobs")

    def hit(self, q, k, v) -> bool:
        return (q.shape == self.qs and k.shape == self.ks and v.shape == self.vs and q.stride() == self.qst
                and k.stride() == self.kst and v.stride() == self.vst and q.dtype is self.dtype and k.dtype is self.dtype
                and v.dtype is self.dtype and q.device == self.device
                and self.knobs == (_cfg.scale_mode, _cfg.tie_order, _cfg.gqa_dedup)
                and _get_device() == self.dev_index)

    def run(self, q, k, v, idx_out=None):
        qp, kp, vp = q.data_ptr(), k.data_ptr(), v.data_ptr()
        if (qp | kp | vp) & 15:
            return None                                   # a view that starts off a 16-byte boundary: the general path copies it
        if idx_out is not None:
            if idx_out.dtype != torch.int32 or not idx_out.is_contiguous() or idx_out.numel() != self.nidx:
                raise ValueError("idx_out must be a contiguous int32 tensor of B*H*k elements")
            ip = idx_out.data_ptr()
        else:
            ip = None
        st = _raw_stream(self.dev_index)
        ws = _WS.get((self.dev_index, st))
        if ws is None or ws.numel() < self.nb:
            ws = _workspace_and_stream(self.nb, self.device)[0]
        ko = torch.empty(self.out_shape, dtype=self.dtype, device=self.device)
        vo = torch.empty(self.out_shape, dtype=self.dtype, device=self.device)
        rc = self.fn(self.dref, qp, kp, vp, ko.data_ptr(), vo.data_ptr(), ip, ws.data_ptr(), ws.numel(), st)
        if rc:
            N.check(rc, self.what)
        return ko, vo


def prepare_compress(q, k, v, window: int, topk_k: int, pooling, kernel_size: int, scale_mode: str, kv_group: int, h2o: bool,
                     k_head_step: int = 1):
    """-> PreparedCompress for exactly these operand layouts, or None when the call needs the general path.
    ``k_head_step`` = g reads every g-th K/V head of expanded tensors (config.gqa_dedup: the view ``t[:, ::g]`` without
    building it: same base pointer, head stride x g)."""
    if not (q.is_cuda and k.is_cuda and v.is_cuda) or _needs_full_sort(q.shape[2] - window, topk_k, q.dtype):
        return None
    if any(_rowmajor(t) is not t for t in (q, k, v)) or q.device != k.device or q.device != v.device:
        return None
    pc = PreparedCompress()
    pc.qs, pc.ks, pc.vs = q.shape, k.shape, v.shape
    pc.qst, pc.kst, pc.vst = q.stride(), k.stride(), v.stride()
    pc.dtype, pc.device = q.dtype, q.device
    pc.dev_index = q.device.index if q.device.index is not None else torch.cuda.current_device()
    kk, vv = (k, v) if k_head_step == 1 else (k[:, ::k_head_step], v[:, ::k_head_step])
    pc.desc = make_desc(q, kk, vv, window, None if h2o else pooling, kernel_size, "sum", scale_mode, topk_k, kv_group)
    pc.dref = N.C.byref(pc.desc)
    pc.nb = N.lib.pkv_workspace_bytes(pc.desc)
    pc.fn = N.lib.pkv_compress_h2o if h2o else N.lib.pkv_compress
    pc.what = "pkv_compress_h2o" if h2o else "pkv_compress"
    B, H, _, D = q.shape
    pc.out_shape = (B, H, topk_k + window, D)
    pc.nidx = B * H * topk_k
    pc.knobs = (_cfg.scale_mode, _cfg.tie_order, _cfg.gqa_dedup)
    return pc


class PreparedAda:
    """The two C calls of AdaKVCluster.update_kv on the list path (pkv_ada_select, then pkv_gather_flat on the device-resident
    capacities) for fixed shapes and list length M.  ``run`` returns (head_lens, cu_klen, cu_headlens, K_flat, V_flat) with the
    flat outputs sized by ``rows_bound``; the caller narrows them once the capacities are on the host."""
    __slots__ = ("qs", "ks", "vs", "qst", "kst", "vst", "dtype", "device", "dev_index", "H", "M", "D", "dsel", "dsel_ref", "nb",
                 "dgat", "dgat_ref", "rows_bound", "sizes", "base", "floor", "normalize", "knobs", "f_sel", "f_gat", "_spare")

    hit = PreparedCompress.hit

    def run(self, q, k, v, mirror_ptr, seq, given_ptr=None):
        """``given_ptr`` (HeadKV): device int32 [H] of host-derived capacities - no budgets are computed, ``rows_bound`` is then the
        exact number of output rows."""
        qp, kp, vp = q.data_ptr(), k.data_ptr(), v.data_ptr()
        if (qp | kp | vp) & 15:
            return None
        H, M = self.H, self.M
        st = _raw_stream(self.dev_index)
        ws = _WS.get((self.dev_index, st))
        if ws is None or ws.numel() < self.nb:
            ws = _workspace_and_stream(self.nb, self.device)[0]
        buf = self._spare                                                               # cap | head_lens | cu_klen | cu_headlens | lists
        if buf is None:
            buf = torch.empty(4 * H + 1 + H * M, dtype=torch.int32, device=self.device)
        else:
            self._spare = None       # a fresh tensor per call: the metadata views of this call's result own it
        p0 = buf.data_ptr()
        p_hl, p_cu, p_cuh, p_top = p0 + 4 * H, p0 + 8 * H, p0 + 4 * (3 * H + 1), p0 + 4 * (4 * H + 1)
        rc = self.f_sel(self.dsel_ref, qp, kp, self.base, self.floor, self.normalize, given_ptr, p_top, p0, p_hl, p_cu, p_cuh,
                        mirror_ptr, seq, ws.data_ptr(), ws.numel(), st)
        if rc:
            N.check(rc, "pkv_ada_select")
        kf = torch.empty((self.rows_bound, self.D), dtype=self.dtype, device=self.device)
        vf = torch.empty((self.rows_bound, self.D), dtype=self.dtype, device=self.device)
        rc = self.f_gat(self.dgat_ref, kp, vp, p_top, M, given_ptr if given_ptr is not None else p0, p_cu, kf.data_ptr(), vf.data_ptr(),
                        self.rows_bound, st)
        if rc:
            N.check(rc, "pkv_gather_flat")
        _, head_lens, cu, cuh, _ = buf.split(self.sizes)          # views for the metadata attributes: after both calls are issued
        return head_lens, cu, cuh, kf, vf

    def spare(self):
        """Allocate the NEXT run's metadata buffer now (the caller is about to wait for this run's kernels anyway): the
        allocation then no longer sits between that call's entry and its first launch.  66 KB per cluster at H = 32, M = 512;
        given back to the allocator when the cluster is."""
        if self._spare is None:
            self._spare = torch.empty(4 * self.H + 1 + self.H * self.M, dtype=torch.int32, device=self.device)


def prepare_ada(q, k, v, window: int, pooling, kernel_size: int, M: int, base_capacity: int, floor_ratio: float, normalize: bool,
                scale_mode: str, kv_group: int, rows_bound: int):
    """-> PreparedAda for exactly these operand layouts and list length, or None when the call needs the general path."""
    if not (q.is_cuda and k.is_cuda and v.is_cuda) or q.dtype == torch.float32:
        return None
    if any(_rowmajor(t) is not t for t in (q, k, v)) or q.device != k.device or q.device != v.device:
        return None
    pa = PreparedAda()
    pa.qs, pa.ks, pa.vs = q.shape, k.shape, v.shape
    pa.qst, pa.kst, pa.vst = q.stride(), k.stride(), v.stride()
    pa.dtype, pa.device = q.dtype, q.device
    pa.dev_index = q.device.index if q.device.index is not None else torch.cuda.current_device()
    H = q.shape[1]
    pa.H, pa.M, pa.D = H, M, q.shape[3]
    pa.dsel = make_desc(q, k, None, window, pooling, kernel_size, "mean", scale_mode, M, kv_group)
    pa.dsel_ref = N.C.byref(pa.dsel)
    pa.nb = N.lib.pkv_workspace_bytes(pa.dsel)
    pa.dgat = make_desc(None, k, v, window, topk=M, kv_group=kv_group, num_heads=H)
    pa.dgat_ref = N.C.byref(pa.dgat)
    pa.rows_bound = rows_bound
    pa.sizes = [H, H, H + 1, H, H * M]
    pa.base, pa.floor, pa.normalize = int(base_capacity), float(floor_ratio), 1 if normalize else 0
    pa.knobs = (_cfg.scale_mode, _cfg.tie_order, _cfg.gqa_dedup)
    pa.f_sel, pa.f_gat = N.lib.pkv_ada_select, N.lib.pkv_gather_flat
    pa._spare = None
    return pa


def select(q, k, window: int, topk_k: int, pooling, kernel_size: int, scale_mode: str = "div", kv_group: int = 1,
           h2o: bool = False) -> torch.Tensor:
    """Score -> top-k only (the front half of ``compress``): int32 indices [B,H,k] in (value desc, index asc) order."""
    _require_gpu(q, k)
    q, k = _rowmajor(q), _rowmajor(k)
    B, H = q.shape[0], q.shape[1]
    L = q.shape[2] - window
    if _needs_full_sort(L, topk_k, q.dtype):
        scores = score_h2o(q, k, window, scale_mode=scale_mode, kv_group=kv_group) if h2o else \
            score_window(q, k, window, pooling, kernel_size, "sum", scale_mode, kv_group=kv_group)
        order, _ = sort_rows(scores.reshape(B * H, L), want_values=False)
        return order[:, :topk_k].reshape(B, H, topk_k).contiguous()
    with torch.cuda.device(k.device):
        d = make_desc(q, k, None, window, None if h2o else pooling, kernel_size, "sum", scale_mode, topk_k, kv_group)
        ws = workspace(N.lib.pkv_workspace_bytes(d), k.device)
        idx = torch.empty(B, H, topk_k, dtype=torch.int32, device=k.device)
        N.check(N.lib.pkv_select(d, q.data_ptr(), k.data_ptr(), 1 if h2o else 0, idx.data_ptr(), ws.data_ptr(), ws.numel(),
                                 N.stream_ptr()), "pkv_select")
    return idx


def merge_compact(k, v, idx: torch.Tensor, window: int, kv_group: int = 1):
    """pyramidkv_utils.py:119-170 merge_kv(..., "pivot"): idx int32 [B,H,n] -> (K_m [B,H,n+w,D] ordered [window, selected],
    V_m [B,H,n+w,D] ordered [selected, window]) - the reference's own orders."""
    _require_gpu(k, v, idx)
    k, v = _rowmajor(k), _rowmajor(v)
    B, H, n = idx.shape
    idx = idx.to(torch.int32).contiguous()
    D = k.shape[-1]
    with torch.cuda.device(k.device):
        d = make_desc(None, k, v, window, topk=n, kv_group=kv_group, num_heads=H)
        ko = torch.empty(B, H, n + window, D, dtype=k.dtype, device=k.device)
        vo = torch.empty_like(ko)
        nb = N.lib.pkv_merge_workspace_bytes(d)
        ws = workspace(nb, k.device)            # the cached per-(device, stream) scratch: the selection that produced idx is done with it
        N.check(N.lib.pkv_merge_compact(d, k.data_ptr(), v.data_ptr(), idx.data_ptr(), n, ko.data_ptr(), vo.data_ptr(),
                                        ws.data_ptr(), ws.numel(), N.stream_ptr()), "pkv_merge_compact")
    return ko, vo


def sort_rows(scores: torch.Tensor, want_values: bool = True):
    """pyramidkv_utils.py:706.  scores [rows, L] -> (sorted_idx int32 [rows,L], sorted_val or None)."""
    _require_gpu(scores)
    assert scores.dim() == 2
    if scores.stride(-1) != 1:
        scores = scores.contiguous()
    rows, L = scores.shape
    si = torch.empty(rows, L, dtype=torch.int32, device=scores.device)
    sv = torch.empty(rows, L, dtype=scores.dtype, device=scores.device) if want_values else None
    with torch.cuda.device(scores.device):
        N.check(N.lib.pkv_sort_rows(N.dtype_code(scores.dtype), rows, L, scores.data_ptr(), scores.stride(0),
                                    si.data_ptr(), sv.data_ptr() if sv is not None else None, N.stream_ptr()),
                "pkv_sort_rows")
    return si, sv


def ada_budget(sorted_val: torch.Tensor, base_capacity: int, floor_ratio: float, normalize: bool) -> torch.Tensor:
    """pyramidkv_utils.py:709-719.  sorted_val [H,L] descending -> int32 head_capacity [H]."""
    _require_gpu(sorted_val)
    sorted_val = sorted_val.contiguous()
    H, L = sorted_val.shape
    cap = torch.empty(H, dtype=torch.int32, device=sorted_val.device)
    nb = 1024 + 2 * H * 256 * 4
    ws = torch.empty(nb, dtype=torch.uint8, device=sorted_val.device)
    with torch.cuda.device(sorted_val.device):
        N.check(N.lib.pkv_ada_budget(N.dtype_code(sorted_val.dtype), H, L, sorted_val.data_ptr(), base_capacity,
                                     float(floor_ratio), 1 if normalize else 0, cap.data_ptr(), ws.data_ptr(), nb,
                                     N.stream_ptr()), "pkv_ada_budget")
    return cap


def ada_budget_topm(scores: torch.Tensor, top_idx: torch.Tensor, base_capacity: int, floor_ratio: float,
                    normalize: bool, window: Optional[int] = None):
    """pyramidkv_utils.py:709-719 without the full sort of :706: scores [H,L] (un-sorted), top_idx int32 [H,M] = the first
    M >= min(L, H*base) entries of every head's descending order (``topk(scores, M)``) -> int32 head_capacity [H].
    With ``window`` the var-len metadata of :682-691 comes out of the same launch: (cap, head_lens [H], cu_klen [H+1])."""
    _require_gpu(scores, top_idx)
    assert scores.dim() == 2 and top_idx.dim() == 2 and scores.stride(-1) == 1 and top_idx.stride(-1) == 1
    H, L = scores.shape
    M = top_idx.shape[1]
    dev = scores.device
    out = torch.empty(2 * H + 1 + H, dtype=torch.int32, device=dev)          # cap | head_lens | cu_klen (one allocation)
    cap, head_lens, cu = out[:H], out[H:2 * H], out[2 * H:]
    nb = 1024 + 2 * H * 256 * 4
    with torch.cuda.device(dev):
        ws = workspace(nb, dev)
        N.check(N.lib.pkv_ada_budget_topm(N.dtype_code(scores.dtype), H, L, M, scores.data_ptr(), scores.stride(0),
                                          top_idx.data_ptr(), top_idx.stride(0), base_capacity, float(floor_ratio),
                                          1 if normalize else 0, int(window or 0), cap.data_ptr(),
                                          head_lens.data_ptr() if window is not None else None,
                                          cu.data_ptr() if window is not None else None,
                                          ws.data_ptr(), ws.numel(), N.stream_ptr()), "pkv_ada_budget_topm")
    return (cap, head_lens, cu) if window is not None else cap


def ada_budget_rows(scores: torch.Tensor, base_capacity: int, floor_ratio: float, normalize: bool, window: int,
                    host_mirror: Optional[torch.Tensor] = None, host_seq: int = 0):
    """pyramidkv_utils.py:706-719 from the un-sorted rows alone (pkv_ada_budget_rows): scores [H,L] -> (head_capacity,
    head_lens, cu_klen, cu_headlens), device int32.  No sort and no top-M list: selections and counts over the whole row."""
    _require_gpu(scores)
    assert scores.dim() == 2 and scores.stride(-1) == 1
    H, L = scores.shape
    dev = scores.device
    meta = torch.empty(4 * H + 1, dtype=torch.int32, device=dev)
    cap, head_lens, cu, cuh = meta[:H], meta[H:2 * H], meta[2 * H:3 * H + 1], meta[3 * H + 1:]
    nb = 1024 + (4 if scores.dtype == torch.float32 else 2) * H * 256 * 4 + 4 * H * 4
    with torch.cuda.device(dev):
        ws = workspace(nb, dev)
        N.check(N.lib.pkv_ada_budget_rows(N.dtype_code(scores.dtype), H, L, scores.data_ptr(), scores.stride(0), base_capacity,
                                          float(floor_ratio), 1 if normalize else 0, int(window), cap.data_ptr(),
                                          head_lens.data_ptr(), cu.data_ptr(), cuh.data_ptr(),
                                          host_mirror.data_ptr() if host_mirror is not None else None, int(host_seq),
                                          ws.data_ptr(), ws.numel(), N.stream_ptr()), "pkv_ada_budget_rows")
    return cap, head_lens, cu, cuh


def ada_adaptive_lists(scores: torch.Tensor, top_idx: torch.Tensor, base_capacity: int, normalize: bool) -> torch.Tensor:
    """pyramidkv_utils.py:709-711 for a head shard: scores [Hl,L], top_idx int32 [Hl,M] -> the adaptive lists [Hl,M] (model
    dtype) the ranks of the head-sharded Ada-SnapKV exchange."""
    _require_gpu(scores, top_idx)
    Hl, L = scores.shape
    M = top_idx.shape[1]
    out = torch.empty(Hl, M, dtype=scores.dtype, device=scores.device)
    nb = 1024 + 2 * Hl * 256 * 4
    with torch.cuda.device(scores.device):
        ws = workspace(nb, scores.device)
        N.check(N.lib.pkv_ada_adaptive_lists(N.dtype_code(scores.dtype), Hl, L, M, scores.data_ptr(), scores.stride(0),
                                             top_idx.data_ptr(), top_idx.stride(0), base_capacity, 1 if normalize else 0,
                                             out.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()),
                "pkv_ada_adaptive_lists")
    return out


def ada_select(q, k, window: int, pooling, kernel_size: int, M: int, base_capacity: int = 0, floor_ratio: float = 0.0,
               normalize: bool = False, given_capacity: Optional[torch.Tensor] = None, scale_mode: str = "div",
               kv_group: int = 1, host_mirror: Optional[torch.Tensor] = None, host_seq: int = 0):
    """Front half of AdaKVCluster / HeadKVCluster.update_kv in one C call (pkv_ada_select): mean-reduced window score ->
    top-M indices per head -> head budgets (or the given ones) + var-len metadata.
    Returns (top_idx int32 [H,M], head_capacity int32 [H], head_lens int32 [H], cu_klen int32 [H+1], cu_headlens int32 [H])."""
    _require_gpu(q, k)
    q, k = _rowmajor(q), _rowmajor(k)
    H = q.shape[1]
    dev = k.device
    with torch.cuda.device(dev):
        d = make_desc(q, k, None, window, pooling, kernel_size, "mean", scale_mode, M, kv_group)
        ws = workspace(N.lib.pkv_workspace_bytes(d), dev)
        top_idx = torch.empty(H, M, dtype=torch.int32, device=dev)
        meta = torch.empty(4 * H + 1, dtype=torch.int32, device=dev)       # cap | head_lens | cu_klen | cu_headlens (one allocation)
        cap, head_lens, cu, cuh = meta[:H], meta[H:2 * H], meta[2 * H:3 * H + 1], meta[3 * H + 1:]
        N.check(N.lib.pkv_ada_select(d, q.data_ptr(), k.data_ptr(), base_capacity, float(floor_ratio), 1 if normalize else 0,
                                     given_capacity.data_ptr() if given_capacity is not None else None, top_idx.data_ptr(),
                                     cap.data_ptr(), head_lens.data_ptr(), cu.data_ptr(), cuh.data_ptr(),
                                     host_mirror.data_ptr() if host_mirror is not None else None, int(host_seq),
                                     ws.data_ptr(), ws.numel(), N.stream_ptr()), "pkv_ada_select")
    return top_idx, (given_capacity if given_capacity is not None else cap), head_lens, cu, cuh


def ada_metadata(head_capacity: torch.Tensor, window: int):
    """pyramidkv_utils.py:682-691.  -> (head_lens int32 [H], cu_klen int32 [H+1])."""
    _require_gpu(head_capacity)
    H = head_capacity.numel()
    head_lens = torch.empty(H, dtype=torch.int32, device=head_capacity.device)
    cu = torch.empty(H + 1, dtype=torch.int32, device=head_capacity.device)
    with torch.cuda.device(head_capacity.device):
        N.check(N.lib.pkv_ada_metadata(H, window, head_capacity.data_ptr(), head_lens.data_ptr(), cu.data_ptr(),
                                       N.stream_ptr()), "pkv_ada_metadata")
    return head_lens, cu


def gather_flat(k, v, sorted_idx: torch.Tensor, head_capacity: torch.Tensor, cu_klen: torch.Tensor,
                window: int, total_rows: int, max_cap: int = 0, kv_group: int = 1):
    """pyramidkv_utils.py:733-757.  -> flat (K, V) [total_rows, D]."""
    _require_gpu(k, v, sorted_idx)
    k, v = _rowmajor(k), _rowmajor(v)
    H = sorted_idx.shape[0]
    D = k.shape[-1]
    with torch.cuda.device(k.device):
        d = make_desc(None, k, v, window, topk=max_cap, kv_group=kv_group, num_heads=H)
        ko = torch.empty(total_rows, D, dtype=k.dtype, device=k.device)
        vo = torch.empty_like(ko)
        N.check(N.lib.pkv_gather_flat(d, k.data_ptr(), v.data_ptr(), sorted_idx.data_ptr(), sorted_idx.stride(0),
                                      head_capacity.data_ptr(), cu_klen.data_ptr(), ko.data_ptr(), vo.data_ptr(), total_rows,
                                      N.stream_ptr()), "pkv_gather_flat")
    return ko, vo


def update_flatten_view(cache: torch.Tensor, state: torch.Tensor, head_lens: torch.Tensor,
                        cu_klen: torch.Tensor) -> torch.Tensor:
    """Replacement of tiny_api_cuda.update_flatten_view (csrc/csrc/cuda_api.cu:55-85)."""
    _require_gpu(cache, state, head_lens, cu_klen)
    if head_lens.dtype != torch.int32:
        raise ValueError("expected headlens to be int32")          # cuda_api.cu:56
    if cu_klen.dtype != torch.int32:
        raise ValueError("expected cu_dst_pos to be int32")        # cuda_api.cu:57
    cache, state = cache.contiguous(), state.contiguous()
    H = head_lens.numel()
    dim = cache.shape[1]
    out = torch.empty(cache.shape[0] + H, dim, dtype=cache.dtype, device=cache.device)
    with torch.cuda.device(cache.device):
        N.check(N.lib.pkv_update_flatten_view(N.dtype_code(cache.dtype), H, dim, cache.data_ptr(), state.data_ptr(),
                                              head_lens.data_ptr(), cu_klen.data_ptr(), out.data_ptr(),
                                              N.stream_ptr()), "pkv_update_flatten_view")
    return out
