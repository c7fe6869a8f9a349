"""ctypes binding of libce_hip.so (the C ABI in include/ce_api.h).

There is deliberately no CPU fallback: if the HIP library is missing or cannot be loaded
the import fails loudly.  torch is imported first so the library binds to the HIP runtime
(libamdhip64.so.7) that torch already mapped -- device pointers of torch tensors and the
current torch stream are then valid arguments for every entry point.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p
from pathlib import Path

import torch  # noqa: F401  (must precede the CDLL below)

_PKG = Path(__file__).resolve().parent
# CE_LIBRARY: another build of the same sources (tests load libce_hip_testhooks.so, the -DCE_TEST_HOOKS twin, this way)
LIB_PATH = Path(os.environ["CE_LIBRARY"]) if os.environ.get("CE_LIBRARY") else _PKG / "libce_hip.so"

CE_OK = 0
CE_ERR_INVALID = 1
CE_ERR_HIP = 2
CE_ERR_CAPACITY = 3
CE_ERR_NOMEM = 4
CE_ERR_UNSUPPORTED = 5
CE_ERR_RANGE = 6
CE_EVICT_DATASET = 0
CE_EVICT_LFU = 1
CE_MODE_SUM = 0
CE_MODE_MEAN = 1
CE_TRANSPORT_ZEROCOPY = 0
CE_TRANSPORT_STAGED = 1
CE_TRANSPORT_WORKER = 2
CE_CALL_PREPARE = 0
CE_CALL_PRELOAD = 1
CE_CALL_FLUSH = 2


class CeCacheConfig(Structure):
    _fields_ = [
        ("num_embeddings", c_int64),
        ("cuda_row_num", c_int64),
        ("embedding_dim", c_int32),
        ("evict_strategy", c_int32),
        ("transport", c_int32),
        ("protect_depth", c_int32),
        ("max_ids_per_call", c_int64),
        ("host_weight", c_void_p),
        ("host_weight_dev", c_void_p),
        ("cache_weight", c_void_p),
        ("idx_map", c_void_p),
        ("inverted_cached_idx", c_void_p),
        ("cached_idx_map", c_void_p),
        ("freq_cnter", c_void_p),
        ("workspace", c_void_p),
        ("workspace_bytes", c_size_t),
    ]


class CeCallStats(Structure):
    _fields_ = [
        ("seq", c_int64),
        ("n_ids", c_int64),
        ("n_unique", c_int64),
        ("n_miss", c_int64),
        ("n_evict", c_int64),
        ("miss_lookups", c_int64),
        ("n_free_after", c_int64),
        ("status", c_int32),
        ("kind", c_int32),
    ]


# name -> (restype, argtypes); mirrors include/ce_api.h declaration by declaration
_BAG_COMMON = [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_int32, c_int64]
SIGNATURES = {
    "ce_version": (c_int, []),
    "ce_cpu_budget": (c_int32, []),
    "ce_last_error": (c_char_p, []),
    "ce_stream_create_cu_mask": (c_int, [c_void_p, c_int32, POINTER(c_void_p)]),
    "ce_stream_destroy": (c_int, [c_void_p]),
    "ce_host_alloc": (c_int, [c_size_t, c_int, POINTER(c_void_p), POINTER(c_void_p)]),
    "ce_host_free": (c_int, [c_void_p]),
    "ce_host_register": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "ce_host_unregister": (c_int, [c_void_p]),
    "ce_host_fill_uniform": (c_int, [c_void_p, c_int64, c_float, c_float, c_uint64, c_int]),
    "ce_host_fill_uniform_rows": (c_int, [c_void_p, c_int64, c_int32, c_float, c_float, c_uint64, c_void_p, c_void_p]),
    "ce_host_rows_gather": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p]),
    "ce_box_probe": (c_int, [c_void_p, c_size_t, c_int32, POINTER(c_double), POINTER(c_double), c_void_p]),
    "ce_probe_rows": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_int32, POINTER(c_double), POINTER(c_double), c_void_p]),
    "ce_bag_forward": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32, c_int64,
                               c_int32, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "ce_bag_backward_dense": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32, c_int64,
                                      c_int32, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "ce_bag_backward_dense_presorted": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32, c_int64,
                                      c_int32, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p]),
    "ce_bag_backward_rows": (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_int32, c_int64, c_int32, c_void_p,
                                     c_int32, c_int64, c_void_p, c_void_p]),
    "ce_bag_backward_sgd": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32, c_int64,
                                    c_int32, c_void_p, c_int32, c_int64, c_void_p, c_float, c_void_p]),
    "ce_bag_presort_len": (c_int64, [c_int64]),
    "ce_bag_presort": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "ce_bag_backward_sgd_presorted": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32,
                                              c_int64, c_int32, c_void_p, c_int32, c_int64, c_void_p, c_float,
                                              c_void_p, c_void_p]),
    "ce_bag_presort_window": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "ce_bag_presort_window_src": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_int64, c_int64,
                                          c_int32, c_int64, c_void_p, c_void_p]),
    "ce_bag_forward_src_keys": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_void_p]),
    "ce_bag_backward_sgd_presorted_src": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_float, c_void_p,
                                                  c_void_p]),
    "ce_bag_presort_window_src_excl": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_int64,
                                               c_int64, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ce_bag_backward_sgd_presorted_src_excl": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_float,
                                                       c_void_p, c_void_p, c_void_p]),
    "ce_bag_backward_dense_presorted_src": (c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p,
                                                    c_void_p]),
    "ce_bag_backward_sgd_sorted_workspace": (c_size_t, [c_int64, c_int64]),
    "ce_bag_backward_sgd_sorted": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32,
                                           c_int64, c_int32, c_void_p, c_int32, c_int64, c_void_p, c_float,
                                           c_void_p, c_size_t, c_void_p]),
    "ce_cache_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int32]),
    "ce_cache_create": (c_int, [POINTER(CeCacheConfig), c_void_p, POINTER(c_void_p)]),
    "ce_cache_destroy": (c_int, [c_void_p]),
    "ce_cache_preload": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "ce_cache_set_freq_bound": (c_int, [c_void_p, c_int64]),
    "ce_cache_prepare_ids": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "ce_cache_prepare_ids_keys": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int32,
                                          c_int64, c_int64, c_int32, c_int64, c_void_p, c_void_p]),
    "ce_cache_prepare_ids_begin": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int32,
                                           c_int64, c_int64, c_int32, c_int64, c_void_p, c_void_p]),
    "ce_cache_prepare_ids_finish": (c_int, [c_void_p, c_void_p]),
    "ce_cache_prepare_ids_padded": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "ce_cache_prepare_ids_begin_padded": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "ce_cache_set_deferred_rows": (c_int, [c_void_p, c_int32]),
    "ce_cache_rows_ticket": (c_int64, [c_void_p]),
    "ce_cache_wait_rows": (c_int, [c_void_p, c_int64, c_void_p]),
    "ce_cache_last_stats": (c_int, [c_void_p, POINTER(CeCallStats)]),
    "ce_cache_totals": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                                POINTER(c_int64), POINTER(c_int64)]),
    "ce_cache_history": (c_int64, [c_void_p, c_int64, POINTER(CeCallStats), c_int64]),
    "ce_cache_lookup_slots": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "ce_cache_flush": (c_int, [c_void_p, c_void_p]),
    "ce_cache_set_protect_depth": (c_int, [c_void_p, c_int32]),
    "ce_cache_set_transport": (c_int, [c_void_p, c_int32]),
    "ce_cache_get_transport": (c_int32, [c_void_p]),
    "ce_cache_set_cache_weight": (c_int, [c_void_p, c_void_p]),
    "ce_cache_graph_replayed": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "ce_cache_set_buffer_rows": (c_int, [c_void_p, c_int64]),
    "ce_cache_set_profiling": (c_int, [c_void_p, c_int32]),
    "ce_cache_phase_count": (c_int32, []),
    "ce_cache_phase_name": (c_char_p, [c_int32]),
    "ce_cache_phase_times": (c_int, [c_void_p, POINTER(c_double), c_int32, POINTER(c_int64), c_int32]),
    "ce_cache_writeback_wait": (c_int, [c_void_p]),
    "ce_cache_swap_stats": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
    "ce_cache_failures": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int32), POINTER(c_int64)]),
    "ce_cache_free_rows": (c_int, [c_void_p, POINTER(c_int64)]),
    "ce_bucketize_workspace": (c_size_t, [c_int64, c_int32]),
    "ce_bucketize_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "ce_dedupe_bucket_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ce_dedupe_bucket_rows_padded": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ce_dedupe_bucket_rows_padded_window": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int32, c_int64,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_void_p]),
    "ce_exchange_local_index": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                        c_void_p, c_void_p]),
    "ce_split_classify": (c_int, [c_void_p, c_int32, c_int32, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                  c_void_p]),
    "ce_split_places": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    "ce_exchange_local_index_split": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                              c_int64, c_int64, c_int64, c_void_p, c_int32, c_int64, c_int64, c_int64,
                                              c_void_p, c_void_p, c_void_p]),
    "ce_bag_forward_max": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32, c_int64, c_int32,
                                   c_int64, c_void_p, c_void_p, c_void_p]),
    "ce_bag_backward_max": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                    c_float, c_void_p]),
    "ce_bag_backward_psw": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int32, c_int64, c_int32,
                                    c_int64, c_void_p, c_void_p, c_void_p]),
    "ce_rows_renorm_workspace": (c_size_t, [c_int64]),
    "ce_rows_renorm": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_float, c_float, c_void_p, c_size_t,
                               c_void_p]),
    "ce_rows_axpy": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_float, c_void_p]),
}


class CeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libce_hip error {code}: {msg}")
        self.code = code


def _load() -> ctypes.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
    lib = ctypes.CDLL(str(LIB_PATH), mode=os.RTLD_NOW | os.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError (loud) if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error() -> str:
    m = lib.ce_last_error()
    return m.decode("utf-8", "replace") if m else ""


def check(rc: int) -> None:
    if rc != CE_OK:
        raise CeError(rc, last_error())


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("cachedembedding_amd needs a HIP device (MI355X); no GPU is visible and there is "
                           "no CPU fallback for the product path")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device (the raw getter is ~20x cheaper than
    constructing a torch.cuda.Stream object, and this is called for every kernel launch)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
