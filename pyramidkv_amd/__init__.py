"""pyramidkv_amd - MI355X (gfx950) native prefill-time KV-cache eviction (PyramidKV / SnapKV / H2O /
StreamingLLM / Ada-SnapKV / HeadKV ``update_kv``), behind the reference's own plugin surface.

Importing this package loads ``libpkv.so`` (hand-written HIP kernels behind a C ABI, include/pkv.h)
and fails loudly if it has not been built.
"""
from . import _native  # noqa: F401  (raises ImportError when the HIP extension is missing)
from . import config, ops  # noqa: F401
from .cache import DynamicCacheSplitHeadFlatten  # noqa: F401
from .pyramidkv_utils import (  # noqa: F401
    AdaKVCluster, H2OKVCluster, HeadKVCluster, PyramidKVCluster, SnapKVCluster, StreamingLLMKVCluster,
    headkv_head_capacity, init_adakv, init_H2O, init_headkv, init_pyramidkv, init_snapkv, init_StreamingLLM,
)

__version__ = "0.1.0"
