"""CachedParamMgr: the Python mirror of ColossalAI's software-cache manager over the C ABI.

Same names, argument meaning and error behaviour as the class the reference drives
(`embed.cache_weight_mgr.prepare_ids(...)` recsys/dlrm_main.py:259, `print_comm_stats`
benchmark/benchmark_cache.py:75, hit/miss histories recsys/dlrm_main.py:286-289); semantics
per SURVEY.md Appendix A.1-A.6.  All cache work is HIP kernels in libce_hip.so
(csrc/ce_cache.hip); torch only owns the device arrays and the stream.
"""
from __future__ import annotations

import ctypes
import math
import weakref
from enum import Enum
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import CeCacheConfig, CeCallStats, check, lib, ptr, stream_ptr


class EvictionStrategy(Enum):
    """recsys/models/dlrm.py:66,80 choose between these two."""
    LFU = 1
    DATASET = 2


class _PinnedBlock:
    """Owner of one ce_host_alloc block; freed when the last tensor over it is gone."""

    def __init__(self, host_ptr: int):
        self._fin = weakref.finalize(self, _PinnedBlock._free, host_ptr)

    @staticmethod
    def _free(host_ptr: int):
        try:
            lib.ce_host_free(ctypes.c_void_p(host_ptr))
        except Exception:
            pass


class HostTable:
    """[N, D] fp32 table in pinned, device-mapped host DRAM (CachedParamMgr.weight, A.1)."""

    def __init__(self, tensor: torch.Tensor, host_ptr: int, dev_ptr: int, registered: bool):
        self.tensor = tensor
        self.host_ptr = host_ptr
        self.dev_ptr = dev_ptr
        self._fin = weakref.finalize(self, HostTable._release, host_ptr, registered)

    @staticmethod
    def _release(host_ptr: int, registered: bool):
        try:
            if registered:
                lib.ce_host_unregister(ctypes.c_void_p(host_ptr))
        except Exception:
            pass

    @classmethod
    def allocate(cls, num_embeddings: int, dim: int, threads: int = 0) -> "HostTable":
        _lib.require_gpu()
        nbytes = num_embeddings * dim * 4
        hp, dp = ctypes.c_void_p(), ctypes.c_void_p()
        check(lib.ce_host_alloc(nbytes, threads or _default_threads(), ctypes.byref(hp), ctypes.byref(dp)))
        carr = (ctypes.c_float * (num_embeddings * dim)).from_address(hp.value)
        # the pinned block lives exactly as long as the tensors over it: tensor -> numpy array -> ctypes array ->
        # block owner (a `.weight` view that outlives the module keeps the memory instead of dangling)
        carr._ce_block = _PinnedBlock(hp.value)
        t = torch.from_numpy(np.ctypeslib.as_array(carr)).view(num_embeddings, dim)
        return cls(t, hp.value, dp.value, registered=False)

    @classmethod
    def wrap(cls, weight: torch.Tensor) -> "HostTable":
        """Pin + map a tensor the caller owns (the `_weight` / from_pretrained path)."""
        _lib.require_gpu()
        assert weight.device.type == "cpu" and weight.dtype == torch.float32 and weight.is_contiguous()
        dp = ctypes.c_void_p()
        check(lib.ce_host_register(ctypes.c_void_p(weight.data_ptr()), weight.numel() * 4, ctypes.byref(dp)))
        return cls(weight, weight.data_ptr(), dp.value, registered=True)

    def fill_uniform_(self, lo: float, hi: float, seed: int, threads: int = 0):
        check(lib.ce_host_fill_uniform(ctypes.c_void_p(self.host_ptr), self.tensor.numel(), lo, hi, seed,
                                       threads or _default_threads()))
        return self


def _default_threads() -> int:
    import os
    return max(1, min(64, (os.cpu_count() or 1)))


class CachedParamMgr(torch.nn.Module):
    """Manages a [cuda_row_num, D] HBM cache of rows of a host-resident [N, D] table.

    Args mirror upstream: weight (CPU fp32 [N, D], or a HostTable), cuda_row_num,
    buffer_size (rows of staging the async_copy transport may use at a time -- upstream's
    LimitBuffIndexCopyer; 0 = stage a whole swap at once; the default zero-copy transport has no staging),
    pin_weight (the table is always pinned + mapped here), evict_strategy,
    async_copy (maps to the staged hipMemcpyAsync transport when True)."""

    def __init__(self, weight, cuda_row_num: int = 0, buffer_size: int = 0, pin_weight: bool = True,
                 evict_strategy: EvictionStrategy = EvictionStrategy.DATASET, async_copy: bool = False,
                 device: Optional[torch.device] = None, strict: bool = True, use_idx_map: Optional[bool] = None):
        super().__init__()
        _lib.require_gpu()
        if cuda_row_num == 0:
            raise NotImplementedError("cuda_row_num == 0 (no cache) is not implemented")
        self._table = weight if isinstance(weight, HostTable) else HostTable.wrap(weight)
        self._weight = self._table.tensor
        self.num_embeddings, self.embedding_dim = self._weight.shape
        self.cuda_row_num = int(cuda_row_num)
        self.buffer_size = buffer_size
        self.pin_weight = pin_weight
        self._evict_strategy = evict_strategy
        self._async_copy = async_copy
        self.strict = strict           # True: prepare_ids raises on overflow like upstream (one tiny sync)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        N, C, D = self.num_embeddings, self.cuda_row_num, self.embedding_dim
        assert N < 2 ** 31 - 1, "row ids are int32 on the device"
        dev = self.device
        # the only trainable parameter the optimiser sees (A.1)
        self.cuda_cached_weight = torch.nn.Parameter(torch.zeros(C, D, device=dev, dtype=torch.float32))
        # DATASET+freq re-rank needs idx_map; identity otherwise (kept None = no array, no gather)
        self._idx_map: Optional[torch.Tensor] = None
        if use_idx_map:
            self._idx_map = torch.arange(N, device=dev, dtype=torch.int32)
        self.register_buffer("cached_idx_map", torch.empty(C, device=dev, dtype=torch.int32), persistent=False)
        self.register_buffer("inverted_cached_idx", torch.empty(N, device=dev, dtype=torch.int32), persistent=False)
        if evict_strategy == EvictionStrategy.LFU:
            self.register_buffer("freq_cnter", torch.empty(C, device=dev, dtype=torch.int64), persistent=False)
        else:
            self.freq_cnter = None
        self._max_ids = 2 ** 31 - 2
        self._deferred_rows = False
        self._handle = None
        self._workspace = None
        self._create_handle()
        self._hist_seen = 0
        self._failures_seen = 0
        self.num_hits_history: List[int] = []
        self.num_miss_history: List[int] = []
        self.num_write_back_history: List[int] = []

    # ------------------------------------------------------------------ handle plumbing
    def _create_handle(self):
        N, C = self.num_embeddings, self.cuda_row_num
        ws_bytes = lib.ce_cache_workspace_bytes(N, C, self._max_ids, self.embedding_dim)
        self._workspace = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=self.device)
        base = self._workspace.data_ptr()
        aligned = (base + 255) & ~255
        cfg = CeCacheConfig()
        cfg.num_embeddings = N
        cfg.cuda_row_num = C
        cfg.embedding_dim = self.embedding_dim
        cfg.evict_strategy = _lib.CE_EVICT_LFU if self._evict_strategy == EvictionStrategy.LFU else _lib.CE_EVICT_DATASET
        tr = getattr(self, "_transport", None)        # None = never set (CE_TRANSPORT_ZEROCOPY is 0: test identity)
        cfg.transport = tr if tr is not None else (
            _lib.CE_TRANSPORT_STAGED if self._async_copy else _lib.CE_TRANSPORT_ZEROCOPY)
        reapply_worker = cfg.transport == _lib.CE_TRANSPORT_WORKER
        if reapply_worker:           # goes through set_transport() below: that path owns the "unsupported" fallback
            cfg.transport = _lib.CE_TRANSPORT_ZEROCOPY
        cfg.protect_depth = 0
        cfg.max_ids_per_call = self._max_ids
        cfg.host_weight = self._table.host_ptr
        cfg.host_weight_dev = self._table.dev_ptr
        cfg.cache_weight = self.cuda_cached_weight.data_ptr()
        cfg.idx_map = ptr(self._idx_map)
        cfg.inverted_cached_idx = self.inverted_cached_idx.data_ptr()
        cfg.cached_idx_map = self.cached_idx_map.data_ptr()
        cfg.freq_cnter = ptr(self.freq_cnter)
        cfg.workspace = aligned
        cfg.workspace_bytes = ws_bytes
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.ce_cache_create(ctypes.byref(cfg), stream_ptr(), ctypes.byref(h)))
        self._handle = h
        self._fin = weakref.finalize(self, lib.ce_cache_destroy, h)
        if self.buffer_size and self.buffer_size > 0:
            check(lib.ce_cache_set_buffer_rows(h, int(self.buffer_size)))
        if reapply_worker:
            self.set_transport("worker")

    @property
    def weight(self) -> torch.Tensor:
        """The host table (upstream `CachedParamMgr.weight`).  With the worker transport evicted rows reach it
        asynchronously: reading it first waits for the write-backs queued so far (no-op otherwise)."""
        if self._handle is not None:
            check(lib.ce_cache_writeback_wait(self._handle))
        return self._weight

    @property
    def idx_map(self) -> torch.Tensor:
        """id -> cpu_row_idx (identity unless DATASET + ids_freq_mapping)."""
        if self._idx_map is None:
            return torch.arange(self.num_embeddings, device=self.device, dtype=torch.int32)
        return self._idx_map

    @property
    def cuda_available_row_num(self) -> int:
        out = ctypes.c_int64()
        check(lib.ce_cache_free_rows(self._handle, ctypes.byref(out)))
        return out.value

    # ------------------------------------------------------------------ A.2
    @torch.no_grad()
    def reorder(self, ids_freq_mapping=None, warmup_ratio: float = 0.7):
        N, C = self.num_embeddings, self.cuda_row_num
        freq = None
        order = None
        if ids_freq_mapping is not None:
            freq = torch.as_tensor(ids_freq_mapping).to(device=self.device, dtype=torch.int64).view(-1)
            assert freq.numel() == N
            # canonical tie rule (SURVEY B#2): stable descending
            order = torch.argsort(freq, descending=True, stable=True)
            if self._evict_strategy == EvictionStrategy.DATASET:
                inv = torch.empty(N, device=self.device, dtype=torch.int32)
                inv[order] = torch.arange(N, device=self.device, dtype=torch.int32)
                if self._idx_map is None:
                    self._idx_map = inv
                    self._recreate_with_idx_map()
                else:
                    self._idx_map.copy_(inv)
        n = min(int(math.ceil(C * warmup_ratio)), N)
        if n > 0:
            rows = None
            fvals = None
            if self._evict_strategy == EvictionStrategy.LFU and freq is not None:
                rows = order[:n].to(torch.int32).contiguous()
                fvals = freq[order[:n]].contiguous()
            with torch.cuda.device(self.device):
                check(lib.ce_cache_preload(self._handle, ptr(rows), ptr(fvals), n, stream_ptr()))
                if fvals is not None:
                    check(lib.ce_cache_set_freq_bound(self._handle, int(fvals.max().item())))
                torch.cuda.current_stream().synchronize()

    def _recreate_with_idx_map(self):
        # only legal while the cache is still empty (reorder runs once, right after construction)
        assert self.cuda_available_row_num == self.cuda_row_num, "reorder() after rows were cached"
        self._fin.detach()
        lib.ce_cache_destroy(self._handle)
        self._create_handle()
        self._hist_seen = 0
        self._failures_seen = 0

    # ------------------------------------------------------------------ A.3
    @torch.no_grad()
    def prepare_ids(self, ids: torch.Tensor, out: Optional[torch.Tensor] = None, padded: bool = False,
                    defer_rows: bool = False) -> torch.Tensor:
        """`out` (int64, same numel, contiguous) lets a caller keep the slots in a static buffer, e.g. one
        a captured hipGraph reads (pipeline.GraphedWindow).  The buffer is SCRATCH for the whole cache op, not only
        written at its end (the call parks every id's row there first): no other stream may read it until the call has
        finished, and it must not alias `ids`.
        padded=True (ce_cache_prepare_ids_padded): entries of -1 are padding -- no lookup, slot -1 -- as the
        fixed-capacity row-wise exchange produces them.  Without it a -1 is a bad id like any other: IndexError under
        strict=True, a failed call (raise_on_failed_calls) otherwise -- as upstream's idx_map.index_select raises.
        defer_rows (only after set_deferred_rows(True)): the current stream is NOT made to wait for the missed rows; the
        caller does that with wait_rows(rows_ticket()) on the stream that first reads the cache."""
        assert ids.is_cuda, "ids must live on the GPU (recsys/dlrm_main.py:250 moves the batch first)"
        shape = ids.shape
        # (every tensor method below is microseconds of a prefetch_num = 1 step that the launch thread bounds: the common
        # case -- flat int64 ids -- takes none of them)
        flat = ids if (ids.dim() == 1 and ids.dtype == torch.int64 and ids.is_contiguous()) \
            else ids.reshape(-1).long().contiguous()
        if out is None:
            slots = torch.empty_like(flat)
        else:
            assert out.is_cuda and out.dtype == torch.int64 and out.is_contiguous() and out.numel() == flat.numel()
            slots = out if out.dim() == 1 else out.view(-1)
        fn = lib.ce_cache_prepare_ids_padded if padded else lib.ce_cache_prepare_ids
        if torch.cuda.current_device() == self.device.index:      # (the device guard costs ~5 us: only when needed)
            check(fn(self._handle, ptr(flat), flat.numel(), ptr(slots), stream_ptr()))
        else:
            with torch.cuda.device(self.device):
                check(fn(self._handle, ptr(flat), flat.numel(), ptr(slots), stream_ptr()))
        if self._deferred_rows and not defer_rows:
            self.wait_rows()
        self._strict_check()
        return slots if slots.shape == shape else slots.view(shape)

    def _strict_check(self) -> None:
        """strict=True: wait for the call just issued and raise what upstream raises (one tiny sync)"""
        if self.strict and not torch.cuda.is_current_stream_capturing():
            st = CeCallStats()
            rc = lib.ce_cache_last_stats(self._handle, ctypes.byref(st))
            self._pull_history()
            if rc != _lib.CE_OK:
                self._failures_seen += 1
            if rc == _lib.CE_ERR_CAPACITY:
                raise AssertionError(_lib.last_error())
            if rc == _lib.CE_ERR_RANGE:
                raise IndexError(_lib.last_error())
            check(rc)

    @torch.no_grad()
    def prepare_ids_begin(self, ids: torch.Tensor, out: torch.Tensor, keys_out: Optional[torch.Tensor] = None, **layout):
        """First half of prepare_ids_keys(ids, out, keys_out, **layout) (or, keys_out=None, of prepare_ids over the
        [P, n] window): everything up to the staging of the victims; prepare_ids_finish() enqueues the rest on the
        same stream.  See ce_cache_prepare_ids_begin in include/ce_api.h for why a pipeline would split the call."""
        return self.prepare_ids_keys(ids, out, keys_out, _begin_only=True, **layout)

    @torch.no_grad()
    def prepare_ids_begin_padded(self, ids: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """First half of prepare_ids(ids, out=out, padded=True) (flat int64 ids, -1 = padding); prepare_ids_finish()
        enqueues the rest on the same stream."""
        assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous()
        assert out.is_cuda and out.dtype == torch.int64 and out.is_contiguous() and out.numel() == ids.numel()
        with torch.cuda.device(self.device):
            check(lib.ce_cache_prepare_ids_begin_padded(self._handle, ptr(ids), ids.numel(), ptr(out), stream_ptr()))
        return out

    @torch.no_grad()
    def prepare_ids_finish(self, defer_rows: bool = False) -> None:
        if torch.cuda.current_device() == self.device.index:
            check(lib.ce_cache_prepare_ids_finish(self._handle, stream_ptr()))
        else:
            with torch.cuda.device(self.device):
                check(lib.ce_cache_prepare_ids_finish(self._handle, stream_ptr()))
        if self._deferred_rows and not defer_rows:
            self.wait_rows()
        self._strict_check()

    @torch.no_grad()
    def prepare_ids_keys(self, ids: torch.Tensor, out: torch.Tensor, keys_out: Optional[torch.Tensor], *,
                         offsets: Optional[torch.Tensor] = None, include_last_offset: bool = False,
                         hook_features: int = 0, identity_bags: bool = False, _begin_only: bool = False,
                         defer_rows: bool = False) -> torch.Tensor:
        """prepare_ids for a prefetch window of P equal batches (ids [P, n] int64) that also leaves the window's keys in
        keys_out ([P, presort_len(n)] int64) -- functional.presort_window's keys, written by the cache op's last kernel
        together with the slots instead of by a launch of its own (ce_cache_prepare_ids_keys).  offsets given: source-row
        keys (see presort_window).  `out` ([P, n] int64) receives the slots and is scratch during the call, as in
        prepare_ids.  Returns `out`."""
        assert ids.is_cuda and ids.dim() == 2 and ids.dtype == torch.int64 and ids.is_contiguous()
        P, n = ids.shape
        assert out.is_cuda and out.dtype == torch.int64 and out.is_contiguous() and out.numel() == P * n
        klen = int(lib.ce_bag_presort_len(n))
        assert keys_out is not None or _begin_only
        assert keys_out is None or (keys_out.is_cuda and keys_out.dtype == torch.int64 and keys_out.is_contiguous()
                                    and keys_out.numel() == P * klen)
        src, off_ptr, off64, stride, num_bags = 0, 0, 0, 0, n
        if offsets is not None:
            assert offsets.is_cuda and offsets.dtype in (torch.int32, torch.int64) and offsets.is_contiguous()
            assert offsets.dim() == 1 or (offsets.dim() == 2 and offsets.shape[0] == P)
            per = offsets.shape[-1]
            num_bags = per - 1 if include_last_offset else per
            if hook_features and num_bags % hook_features:
                raise ValueError("hook_features must divide the number of bags")
            if identity_bags:
                assert num_bags == n, "identity_bags needs one id per bag"
            src, off_ptr = 1, (0 if identity_bags else ptr(offsets))
            off64, stride = int(offsets.dtype == torch.int64), (per if offsets.dim() == 2 else 0)
        args = (self._handle, ptr(ids), P, n, ptr(out), src, off_ptr, off64, stride, num_bags, int(include_last_offset),
                int(hook_features), ptr(keys_out))
        fn = lib.ce_cache_prepare_ids_begin if _begin_only else lib.ce_cache_prepare_ids_keys
        if torch.cuda.current_device() == self.device.index:
            check(fn(*args, stream_ptr()))
        else:
            with torch.cuda.device(self.device):
                check(fn(*args, stream_ptr()))
        if not _begin_only:
            if self._deferred_rows and not defer_rows:
                self.wait_rows()
            self._strict_check()
        return out

    def graph_replayed(self, n_calls: int, ids_per_call: int) -> None:
        """Report `n_calls` prepare_ids calls that a hipGraph launched just now on the current stream replayed
        (prepare_ids issued during a stream capture is recorded into the graph; zero-copy transport only)."""
        with torch.cuda.device(self.device):
            check(lib.ce_cache_graph_replayed(self._handle, int(n_calls), int(ids_per_call), stream_ptr()))

    def _id_to_cached_cuda_id(self, ids: torch.Tensor) -> torch.Tensor:
        flat = ids.reshape(-1).long().contiguous()
        slots = torch.empty_like(flat)
        check(lib.ce_cache_lookup_slots(self._handle, ptr(flat), flat.numel(), ptr(slots), stream_ptr()))
        return slots.view(ids.shape)

    # ------------------------------------------------------------------ A.7 flush
    @torch.no_grad()
    def flush(self):
        with torch.cuda.device(self.device):
            check(lib.ce_cache_flush(self._handle, stream_ptr()))
            st = CeCallStats()
            check(lib.ce_cache_last_stats(self._handle, ctypes.byref(st)))
        self._pull_history()
        assert self.cuda_available_row_num == self.cuda_row_num

    # ------------------------------------------------------------------ stats
    def sync_stats(self) -> CeCallStats:
        st = CeCallStats()
        rc = lib.ce_cache_last_stats(self._handle, ctypes.byref(st))
        self._pull_history()
        if rc not in (_lib.CE_OK, _lib.CE_ERR_CAPACITY, _lib.CE_ERR_RANGE):
            check(rc)
        return st

    def _pull_history(self):
        cap = 4096
        buf = (CeCallStats * cap)()
        while True:
            n = lib.ce_cache_history(self._handle, self._hist_seen + 1, buf, cap)
            for i in range(n):
                r = buf[i]
                self._hist_seen = max(self._hist_seen, r.seq)
                if r.kind != _lib.CE_CALL_PREPARE:
                    continue
                self.num_hits_history.append(int(r.n_unique - r.n_miss))
                self.num_miss_history.append(int(r.n_miss))
                self.num_write_back_history.append(int(r.n_evict) if r.status == _lib.CE_OK else 0)
            if n < cap:
                break

    def totals(self):
        a, b, c, d, e = (ctypes.c_int64() for _ in range(5))
        check(lib.ce_cache_totals(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d),
                                  ctypes.byref(e)))
        return dict(cpu_to_cuda_numel=a.value, cuda_to_cpu_numel=b.value, cache_miss=c.value,
                    total_cache=d.value, calls=e.value)

    def print_comm_stats(self):
        self.sync_stats()
        t = self.totals()
        esz = 4
        msg = (f"CUDA->CPU {t['cuda_to_cpu_numel'] * esz / 1e6:.2f} MB, CPU->CUDA "
               f"{t['cpu_to_cuda_numel'] * esz / 1e6:.2f} MB, cache miss {t['cache_miss']} / {t['total_cache']} "
               f"lookups ({100.0 * t['cache_miss'] / max(1, t['total_cache']):.2f} %)")
        wb = self.writeback_stats()
        if wb["jobs"]:
            row_b = self.embedding_dim * esz
            def rate(nbytes, busy_s):
                # (the chained admission is a kernel on the library's admission stream: no worker thread times it)
                return f"{nbytes / busy_s / 1e9:.1f} GB/s" if busy_s > 0 else "a rate no worker thread timed"
            msg += (f"; swap workers: out {wb['jobs']} jobs {wb['rows'] * row_b / 1e6:.2f} MB at "
                    f"{rate(wb['rows'] * row_b, wb['out_busy_s'])}, in {wb['in_jobs']} jobs "
                    f"{wb['in_rows'] * row_b / 1e6:.2f} MB at {rate(wb['in_rows'] * row_b, wb['in_busy_s'])}")
        print(msg)
        return msg

    def set_protect_depth(self, depth: int):
        check(lib.ce_cache_set_protect_depth(self._handle, int(depth)))

    def set_async_copy(self, flag: bool):
        self._async_copy = bool(flag)
        self.set_transport("staged" if flag else "zerocopy")

    _TRANSPORTS = {"zerocopy": _lib.CE_TRANSPORT_ZEROCOPY, "staged": _lib.CE_TRANSPORT_STAGED,
                   "worker": _lib.CE_TRANSPORT_WORKER}

    def set_transport(self, name: str):
        """'zerocopy' (swap kernels address the mapped host table), 'staged' (upstream async_copy: pinned staging +
        hipMemcpyAsync + host gather/scatter on the calling thread) or 'worker' (admissions zero-copy, evictions
        through one SDMA copy + a worker thread inside the library; the host table lags until writeback_wait /
        flush / a read of `.weight`)."""
        with torch.cuda.device(self.device):
            rc = lib.ce_cache_set_transport(self._handle, self._TRANSPORTS[name])
            if rc == _lib.CE_ERR_UNSUPPORTED and name == "worker":
                # no hipStreamWaitValue64 / no stream priorities on this device: stay on the zero-copy kernels
                import warnings
                warnings.warn(f"worker transport unavailable ({_lib.last_error()}); using zerocopy")
                name = "zerocopy"
                rc = lib.ce_cache_set_transport(self._handle, self._TRANSPORTS[name])
            check(rc)
        self._transport = self._TRANSPORTS[name]

    def reserve_tail(self, rows: int) -> torch.Tensor:
        """fp32 [rows, D] lying right behind the cache in ONE allocation (the cache moves there, contents kept;
        `cuda_cached_weight` stays the same Parameter object).  The row-wise exchange receives into it, so that
        "cache + received rows" is one table for the bag kernels (ce_exchange_local_index).  Blocks until the device
        is idle; call it before anything that captured the cache's address (hipGraphs) -- every address taken earlier
        goes stale.  `cuda_cached_weight.data` is from then on a VIEW of the larger (C + rows) storage: torch.save of a
        state_dict that holds it serialises the whole storage, tail included (clone the parameter before saving, or
        save the host table after flush(), which is what a checkpoint of this module is anyway)."""
        rows = int(rows)
        C, D = self.cuda_row_num, self.embedding_dim
        full = getattr(self, "_cache_full", None)
        if full is None or full.shape[0] < C + rows:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                full = torch.empty(C + rows, D, device=self.device, dtype=torch.float32)
                full[:C].copy_(self.cuda_cached_weight.data)
                full[C:].zero_()
                check(lib.ce_cache_set_cache_weight(self._handle, full.data_ptr()))
            self.cuda_cached_weight.data = full[:C]
            self._cache_full = full
        return self._cache_full[C:C + rows]

    @property
    def cache_with_tail(self) -> torch.Tensor:
        """the allocation `reserve_tail` made: cache rows first, the tail behind them"""
        return self._cache_full

    @property
    def transport_name(self) -> str:
        """the transport in force -- the library falls back from 'worker' to 'zerocopy' (with a message on stderr)
        when its first-use self-test finds that the copy streams cannot run beside a parked cache-op stream"""
        code = lib.ce_cache_get_transport(self._handle)
        return {v: k for k, v in self._TRANSPORTS.items()}.get(code, "zerocopy")

    def set_profiling(self, on: bool = True):
        """hipEvent timers around the phases of prepare_ids (read them with phase_times())."""
        check(lib.ce_cache_set_profiling(self._handle, int(bool(on))))

    def phase_times(self, reset: bool = False) -> dict:
        """{phase name: accumulated ms} over the finished calls since profiling was switched on, plus 'calls'.
        Blocks until the calls issued so far have finished."""
        n = lib.ce_cache_phase_count()
        buf = (ctypes.c_double * n)()
        calls = ctypes.c_int64()
        check(lib.ce_cache_phase_times(self._handle, buf, n, ctypes.byref(calls), int(reset)))
        out = {lib.ce_cache_phase_name(i).decode(): buf[i] for i in range(n)}
        out["calls"] = calls.value
        return out

    def writeback_wait(self):
        check(lib.ce_cache_writeback_wait(self._handle))

    def set_deferred_rows(self, on: bool = True):
        """Worker transport (chained admission): prepare_ids no longer ends by making its stream wait for the missed rows
        (they travel on a stream of the library's own); the caller orders the first reader of the cache behind them
        with wait_rows() -- what a pipeline does that issues the next cache op on the same stream before anything trains
        on this one's slots (ce_cache_set_deferred_rows)."""
        check(lib.ce_cache_set_deferred_rows(self._handle, int(bool(on))))
        self._deferred_rows = bool(on)

    def rows_ticket(self) -> int:
        """ticket of the most recent cache op for wait_rows (0: none yet / another transport)"""
        return int(lib.ce_cache_rows_ticket(self._handle))

    def wait_rows(self, ticket: int = 0):
        """the CURRENT stream waits until the rows of cache op `ticket` (0: the most recent) are in their slots"""
        if torch.cuda.current_device() == self.device.index:
            check(lib.ce_cache_wait_rows(self._handle, int(ticket), stream_ptr()))
        else:
            with torch.cuda.device(self.device):
                check(lib.ce_cache_wait_rows(self._handle, int(ticket), stream_ptr()))

    def writeback_stats(self) -> dict:
        """Worker-transport accounting: rows / jobs / seconds of the write-back (out) and admission (in) workers."""
        sec = (ctypes.c_double * 6)()
        cnt = (ctypes.c_int64 * 4)()
        check(lib.ce_cache_swap_stats(self._handle, sec, cnt))
        return dict(out_wait_s=sec[0], out_busy_s=sec[1], in_wait_s=sec[2], in_busy_s=sec[3], in_gather_s=sec[4],
                    rows=cnt[0], jobs=cnt[1], in_rows=cnt[2], in_jobs=cnt[3], in_jobs_before_writeback_landed=int(sec[5]))

    def raise_on_failed_calls(self):
        """Non-blocking check used by the pipelines that run prepare_ids with strict=False: raises the reference's
        AssertionError (capacity) / IndexError (bad id) if any call that has FINISHED so far failed -- such a call
        returned slots of -1 and the bag kernels skipped those lookups."""
        n, st, sq = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int64()
        check(lib.ce_cache_failures(self._handle, ctypes.byref(n), ctypes.byref(st), ctypes.byref(sq)))
        if n.value > self._failures_seen:
            self._failures_seen = n.value
            if st.value == _lib.CE_ERR_CAPACITY:
                raise AssertionError(
                    f"cache op #{sq.value} needed more unique rows than the {self.cuda_row_num} rows available on "
                    "CUDA (with an overlapped window: unique(window k U window k+1)). Please increase cuda_row_num "
                    "or decrease the training batch size.")
            if st.value == _lib.CE_ERR_RANGE:
                raise IndexError(f"cache op #{sq.value}: an id is outside [0, {self.num_embeddings}) -- every slot of "
                                 "that call is -1.  (Since API 4 an id of -1 is a bad id on prepare_ids / "
                                 "ce_cache_prepare_ids too; padded id lists go through prepare_ids(..., padded=True) / "
                                 "ce_cache_prepare_ids_padded.)")
            raise _lib.CeError(st.value, f"cache op #{sq.value} failed")

    def acknowledge_failures(self) -> int:
        """Mark the calls that have failed so far as seen (they will not make raise_on_failed_calls raise) and return
        how many there are -- for a caller that PROBES with calls it expects to overflow (bench.py sizing the
        prefetch window of a shard cache) and has read their records itself."""
        n, st, sq = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int64()
        self.sync_stats()                                           # every call issued so far has finished
        check(lib.ce_cache_failures(self._handle, ctypes.byref(n), ctypes.byref(st), ctypes.byref(sq)))
        self._failures_seen = n.value
        return n.value

    def cuda_weight_data(self, slot: int) -> torch.Tensor:
        return self.cuda_cached_weight.data[slot]

    def cpu_weight_data(self, row_idx: int) -> torch.Tensor:
        return self.weight[row_idx]
