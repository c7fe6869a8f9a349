"""MI355X-native frequency-aware cached EmbeddingBag (drop-in for the CachedEmbedding
hot path of hpcaitech/CachedEmbedding).

Importing this package loads libce_hip.so (HIP kernels behind the C ABI in
include/ce_api.h).  There is no CPU fallback: the import fails if the library is absent.
"""
from ._lib import CeError, LIB_PATH  # noqa: F401
from .cache_mgr import CachedParamMgr, EvictionStrategy, HostTable  # noqa: F401
from .cached_embedding import CachedEmbeddingBag  # noqa: F401
from .functional import FusedSGD, SrcKeys, embedding_bag  # noqa: F401

__all__ = ["CachedEmbeddingBag", "CachedParamMgr", "EvictionStrategy", "HostTable", "embedding_bag",
           "FusedSGD", "SrcKeys", "CeError", "LIB_PATH"]
