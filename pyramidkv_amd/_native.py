"""ctypes binding of libpkv.so (the C ABI declared in include/pkv.h).

The HIP extension is the product: if it is missing this module raises, there is no eager/PyTorch
fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # imported first so that libamdhip64.so.7 resolves to the runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# PKV_LIB selects another build of the same ABI (tools/ load libpkv_debug.so, the -DPKV_DEBUG build with trace hooks)
LIB_PATH = os.environ.get("PKV_LIB") or os.path.join(_HERE, "libpkv.so")

PKV_BF16, PKV_F16, PKV_F32 = 0, 1, 2
POOL = {None: 0, "none": 0, "avgpool": 1, "maxpool": 2}
REDUCE = {"sum": 0, "mean": 1}
SCALE = {"div": 0, "rcp": 1}
TIE = {"canonical": 0, "aten_rocm": 1}
KERNEL_NAMES = ["logits", "finalize", "topk", "gather", "h2o_stats", "h2o_colsum", "sort", "budget"]


PKV_VERSION = 201            # include/pkv.h PKV_VERSION this binding was written against (checked when the library loads)


class PkvDesc(C.Structure):
    """struct pkv_desc of include/pkv.h (tests/test_abi_and_host.py compares the two field by field)."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("S", C.c_int32), ("D", C.c_int32),
        ("kv_group", C.c_int32), ("reserved0", C.c_int32),
        ("q_stride", C.c_int64 * 3), ("k_stride", C.c_int64 * 3), ("v_stride", C.c_int64 * 3),
        ("window", C.c_int32), ("pool_kind", C.c_int32), ("pool_kernel", C.c_int32),
        ("reduce", C.c_int32), ("scale_mode", C.c_int32), ("topk", C.c_int32), ("tie_order", C.c_int32),
        ("reserved1", C.c_int32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if not self.struct_size:                 # the size of THIS layout: what the library is allowed to read
            self.struct_size = C.sizeof(PkvDesc)


class PkvError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the gfx950 HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C pyramidkv_amd/csrc`). "
            "pyramidkv_amd has no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.pkv_version.restype = C.c_int
    have = lib.pkv_version()
    if have != PKV_VERSION:          # libpkv.so is built in-tree and git-ignored: a stale build must not be driven by newer Python
        raise ImportError(
            f"{LIB_PATH} reports PKV_VERSION {have}, this package binds version {PKV_VERSION} (include/pkv.h): rebuild it with "
            "`make -C pyramidkv_amd/csrc clean all` or `python -c 'import __graft_entry__ as g; g.build()'`")
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    dp = C.POINTER(PkvDesc)
    sig = {
        "pkv_version": (C.c_int, []),
        "pkv_strerror": (C.c_char_p, [C.c_int]),
        "pkv_last_hip_error": (C.c_int, []),
        "pkv_runtime_reset": (C.c_int, []),
        "pkv_workspace_bytes": (sz, [dp]),
        "pkv_score_window": (C.c_int, [dp, vp, vp, vp, i64, vp, sz, vp]),
        "pkv_score_h2o": (C.c_int, [dp, vp, vp, vp, i64, vp, sz, vp]),
        "pkv_topk": (C.c_int, [i32, i32, i32, i32, vp, i64, vp, vp, i64, vp]),
        "pkv_topk_workspace_bytes": (sz, [i32, i32, i32]),
        "pkv_topk_ws": (C.c_int, [i32, i32, i32, i32, vp, i64, vp, vp, i64, vp, sz, vp]),
        "pkv_gather_compact": (C.c_int, [dp, vp, vp, vp, i64, vp, vp, vp]),
        "pkv_gather_streaming": (C.c_int, [dp, vp, vp, vp, vp, vp]),
        "pkv_compress": (C.c_int, [dp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
        "pkv_compress_h2o": (C.c_int, [dp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
        "pkv_select": (C.c_int, [dp, vp, vp, i32, vp, vp, sz, vp]),
        "pkv_merge_workspace_bytes": (sz, [dp]),
        "pkv_merge_compact": (C.c_int, [dp, vp, vp, vp, i64, vp, vp, vp, sz, vp]),
        "pkv_sort_rows": (C.c_int, [i32, i32, i32, vp, i64, vp, vp, vp]),
        "pkv_ada_budget": (C.c_int, [i32, i32, i32, vp, i32, C.c_double, i32, vp, vp, sz, vp]),
        "pkv_ada_budget_topm": (C.c_int, [i32, i32, i32, i32, vp, i64, vp, i64, i32, C.c_double, i32, i32, vp, vp, vp, vp, sz, vp]),
        "pkv_ada_budget_rows": (C.c_int, [i32, i32, i32, vp, i64, i32, C.c_double, i32, i32, vp, vp, vp, vp, vp, i32, vp, sz, vp]),
        "pkv_ada_adaptive_lists": (C.c_int, [i32, i32, i32, i32, vp, i64, vp, i64, i32, i32, vp, vp, sz, vp]),
        "pkv_ada_select": (C.c_int, [dp, vp, vp, i32, C.c_double, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, sz, vp]),
        "pkv_ada_metadata": (C.c_int, [i32, i32, vp, vp, vp, vp]),
        "pkv_gather_flat": (C.c_int, [dp, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp]),
        "pkv_update_flatten_view": (C.c_int, [i32, i32, i32, vp, vp, vp, vp, vp, vp]),
        "pkv_allgather_indices": (C.c_int, [vp, vp, vp, i32, i32, i32, vp, sz, vp]),
        "pkv_last_nccl_error": (C.c_int, []),
        "pkv_debug_build": (C.c_int, []),
        "pkv_debug_topk_trace": (C.c_int, [vp]),
        "pkv_debug_wg_trace": (C.c_int, [vp]),
        "pkv_debug_exp": (C.c_int, [vp, vp, i64, vp]),
        "pkv_debug_round": (C.c_int, [i32, vp, vp, i64, vp]),
        "pkv_debug_scale_multiplier": (C.c_float, [i32, i32, i32]),
        "pkv_prof_enable": (C.c_int, [C.c_int]),
        "pkv_prof_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTED = _load()


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = lib.pkv_strerror(rc).decode()
    if rc == -6:
        msg += f" (hipError {lib.pkv_last_hip_error()})"
    if rc == -8:
        msg += f" (ncclResult {lib.pkv_last_nccl_error()})"
    if rc in (-1, -2, -3, -5, -9):
        raise ValueError(f"libpkv {what}: {msg}")
    raise PkvError(f"libpkv {what}: {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return PKV_BF16
    if dt == torch.float16:
        return PKV_F16
    if dt == torch.float32:      # window policies + StreamingLLM only (include/pkv.h: PKV_F32); the rest raises ValueError
        return PKV_F32
    raise ValueError(f"pyramidkv_amd supports bf16/fp16/fp32 tensors, got {dt}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def prof_enable(on: bool) -> bool:
    return bool(lib.pkv_prof_enable(1 if on else 0))


def prof_read(reset: bool = True):
    ms = (C.c_double * len(KERNEL_NAMES))()
    n = (C.c_int64 * len(KERNEL_NAMES))()
    check(lib.pkv_prof_read(ms, n, 1 if reset else 0), "prof_read")
    return {KERNEL_NAMES[i]: (ms[i], n[i]) for i in range(len(KERNEL_NAMES))}
