"""CPU ORACLE (test infrastructure -- never imported by the product path).

A numpy restatement of the software-cache manager that hpcaitech/CachedEmbedding uses
through ``colossalai.nn.parallel.layers.CachedParamMgr``.

PARITY UNPINNED: the reference tree holds no tests, golden vectors or fixtures for this
path, and the arithmetic lives in an un-vendored third-party package (ColossalAI,
``colossalai/nn/parallel/layers/cache_embedding/cache_mgr.py`` at the commit quoted in
/root/reference/README.md:37, e8d8eda5e7a0619bd779e35065397679e1536dcd) that is absent
from /root/reference and cannot be installed here.  The algorithm below is therefore
restated from SURVEY.md Appendix A (the recalled upstream semantics) and anchored on the
reference's own call sites:

  * ``cache_weight_mgr.prepare_ids(torch.cat(sparse_values))``  recsys/dlrm_main.py:259
  * ``CachedEmbeddingBag(N, D, sparse=True, include_last_offset=True, evict_strategy=...)``
    benchmark/benchmark_cache.py:39-40 and the forward at :62
  * ``ParallelCachedEmbeddingBag(sum(num_embeddings_per_feature), ...)`` recsys/models/dlrm.py:70-81
  * ``print_comm_stats`` / hit+miss histories  recsys/dlrm_main.py:286-294, benchmark/benchmark_cache.py:74-75

The only known-answer test is upstream ColossalAI's LFU check
(``num_hits_history[-6:] == [3,0,1,0,1,1]``), replayed in tests/test_oracle.py.

Canonical tie rules (torch.topk / argsort leave ties
implementation-defined, so "id-exact evict sets" needs a rule):
  * frequency re-rank: stable descending sort (ties -> ascending id);
  * LFU warm-up top-k: same order;
  * LFU victims: (freq ascending, slot ascending);
  * DATASET victims: cpu_row_idx descending (keys are unique, no ties).

Everything is int64/fp32 numpy; row payloads are bit-copies.
"""
from __future__ import annotations

import math
import sys
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

LFU = "lfu"
DATASET = "dataset"
MAXSIZE = sys.maxsize  # freq_cnter initial value upstream (A.1)


@dataclass
class CallTrace:
    """What one prepare_ids call did -- the quantities the parity tests compare bit-exactly."""
    unique_rows: np.ndarray
    miss_rows: np.ndarray
    evicted_rows: np.ndarray      # cpu_row_idx of the victims (any order; compare as a set)
    evicted_slots: np.ndarray
    admit_slots: np.ndarray       # i-th miss row -> i-th slot
    hits: int = 0
    misses: int = 0


class OracleCachedParamMgr:
    """Restatement of upstream ``CachedParamMgr``.

    weight            host table fp32[N, D]; row index = cpu_row_idx            (A.1)
    cuda_cached_weight  fp32[C, D], zero-initialised                             (A.1)
    idx_map           int64[N]  dataset id -> cpu_row_idx (identity unless DATASET+freq)
    cached_idx_map    int64[C]  slot -> cpu_row_idx, -1 = empty
    inverted_cached_idx int64[N] cpu_row_idx -> slot, -1 = not resident
    freq_cnter        int64[C]  LFU only, initial sys.maxsize
    """

    def __init__(self, weight: np.ndarray, cuda_row_num: int, evict_strategy: str = DATASET):
        assert weight.ndim == 2
        if cuda_row_num == 0:
            raise NotImplementedError("cuda_row_num == 0")  # A.1
        self.weight = weight
        self.num_embeddings, self.embedding_dim = weight.shape
        self.cuda_row_num = int(cuda_row_num)
        self.evict_strategy = evict_strategy
        N, C = self.num_embeddings, self.cuda_row_num
        self.cuda_cached_weight = np.zeros((C, self.embedding_dim), dtype=weight.dtype)
        self.idx_map = np.arange(N, dtype=np.int64)
        self.cached_idx_map = np.full(C, -1, dtype=np.int64)
        self.inverted_cached_idx = np.full(N, -1, dtype=np.int64)
        self.freq_cnter = np.full(C, MAXSIZE, dtype=np.int64) if evict_strategy == LFU else None
        self.cuda_available_row_num = C
        self.num_hits_history: List[int] = []
        self.num_miss_history: List[int] = []
        self.num_write_back_history: List[int] = []
        self.cpu_to_cuda_numel = 0
        self.cuda_to_cpu_numel = 0
        self.cache_miss = 0      # sum of multiplicities of missed rows (A.3-4)
        self.total_cache = 0     # ids seen
        self.traces: List[CallTrace] = []
        # build extension (not upstream): rows of the previous `protect_depth` calls stay
        # protected, which is what an overlapped prefetch pipeline needs.
        self.protect_depth = 0
        self._protect_history: List[np.ndarray] = []

    # ------------------------------------------------------------------ A.2
    def reorder(self, ids_freq_mapping: Optional[np.ndarray] = None, warmup_ratio: float = 0.7):
        N, C = self.num_embeddings, self.cuda_row_num
        order = None
        if ids_freq_mapping is not None:
            freq = np.asarray(ids_freq_mapping, dtype=np.int64)
            assert freq.shape == (N,)
            # canonical: stable descending (ties -> ascending id)             B#2
            order = np.argsort(-freq, kind="stable")
            if self.evict_strategy == DATASET:
                inv = np.empty(N, dtype=np.int64)
                inv[order] = np.arange(N, dtype=np.int64)   # argsort(argsort(desc))
                self.idx_map = inv
        n = min(int(math.ceil(C * warmup_ratio)), N)
        if n > 0:
            if self.evict_strategy == LFU and ids_freq_mapping is not None:
                rows = order[:n].astype(np.int64)           # topk(freq, n), canonical ties
                freq_vals = freq[rows]
            else:
                rows = np.arange(n, dtype=np.int64)
                freq_vals = None
            slots = np.arange(n, dtype=np.int64)
            self.cuda_cached_weight[slots] = self.weight[rows]
            self.cached_idx_map[slots] = rows
            self.inverted_cached_idx[rows] = slots
            self.cuda_available_row_num -= n
            if self.evict_strategy == LFU:
                self.freq_cnter[slots] = 0 if freq_vals is None else freq_vals

    # ------------------------------------------------------------------ A.5
    def _find_evict_gpu_idxs(self, k: int, protected_rows: np.ndarray) -> np.ndarray:
        protected = np.isin(self.cached_idx_map, protected_rows)
        slots = np.arange(self.cuda_row_num, dtype=np.int64)
        if self.evict_strategy == DATASET:
            key = self.cached_idx_map.copy()
            key[protected] = -2
            # k largest cpu_row_idx; unique keys among real entries -> no ties
            order = np.lexsort((slots, -key))
            victims = order[:k]
            assert np.all(key[victims] >= 0), "selected an empty/protected slot"
        else:
            key = self.freq_cnter.copy()
            key[protected] = MAXSIZE
            # empty slots are never eligible (their counter is maxsize at every reachable state)
            key[self.cached_idx_map < 0] = MAXSIZE
            order = np.lexsort((slots, key))                # (freq asc, slot asc)   B#1
            victims = order[:k]
            assert np.all(key[victims] < MAXSIZE), "selected an empty/protected slot"
        return victims.astype(np.int64)

    # ------------------------------------------------------------------ A.4
    def _prepare_rows_on_cuda(self, miss: np.ndarray, protected_rows: np.ndarray, trace: CallTrace):
        D = self.embedding_dim
        k = len(miss) - self.cuda_available_row_num
        if k > 0:
            victims = self._find_evict_gpu_idxs(k, protected_rows)
            evict_rows = self.cached_idx_map[victims]
            self.weight[evict_rows] = self.cuda_cached_weight[victims]    # always written back
            self.cached_idx_map[victims] = -1
            self.inverted_cached_idx[evict_rows] = -1
            self.cuda_available_row_num += k
            self.cuda_to_cpu_numel += k * D
            self.num_write_back_history[-1] += k
            trace.evicted_rows = evict_rows.copy()
            trace.evicted_slots = victims.copy()
        m = len(miss)
        if m > 0:
            free = np.nonzero(self.cached_idx_map == -1)[0][:m].astype(np.int64)
            assert len(free) == m
            self.cuda_cached_weight[free] = self.weight[miss]
            self.cached_idx_map[free] = miss
            self.inverted_cached_idx[miss] = free
            if self.evict_strategy == LFU:
                self.freq_cnter[free] = 0
            self.cuda_available_row_num -= m
            self.cpu_to_cuda_numel += m * D
            trace.admit_slots = free.copy()

    # ------------------------------------------------------------------ A.3 / A.6
    def prepare_ids(self, ids: np.ndarray) -> np.ndarray:
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        rows_all = self.idx_map[ids]
        rows, cnt = np.unique(rows_all, return_counts=True)
        if len(rows) > self.cuda_row_num:
            raise AssertionError(
                f"You move {len(rows)} embedding rows from CPU to CUDA. {self.cuda_row_num} rows are "
                "available on CUDA. Please increase cuda_row_num or decrease the training batch size.")
        resident = self.inverted_cached_idx[rows] >= 0            # == isin(rows, cached_idx_map)
        miss = rows[~resident]
        self.cache_miss += int(cnt[~resident].sum())
        self.total_cache += int(ids.size)
        self.num_hits_history.append(int(len(rows) - len(miss)))
        self.num_miss_history.append(int(len(miss)))
        self.num_write_back_history.append(0)
        empty = np.zeros(0, dtype=np.int64)
        trace = CallTrace(rows.copy(), miss.copy(), empty, empty, empty,
                          hits=int(len(rows) - len(miss)), misses=int(len(miss)))
        protected_rows = rows
        if self.protect_depth > 0 and self._protect_history:
            protected_rows = np.unique(np.concatenate([rows] + self._protect_history[-self.protect_depth:]))
        self._prepare_rows_on_cuda(miss, protected_rows, trace)
        if self.protect_depth > 0:
            self._protect_history.append(rows.copy())
            self._protect_history = self._protect_history[-self.protect_depth:]
        slots = self.inverted_cached_idx[rows_all]
        assert np.all(slots >= 0)
        if self.evict_strategy == LFU:
            np.add.at(self.freq_cnter, self.inverted_cached_idx[rows], cnt)
        self.traces.append(trace)
        return slots.reshape(ids.shape)

    # ------------------------------------------------------------------ A.7 flush()
    def flush(self):
        occupied = np.nonzero(self.cached_idx_map >= 0)[0]
        rows = self.cached_idx_map[occupied]
        self.weight[rows] = self.cuda_cached_weight[occupied]
        self.cuda_to_cpu_numel += len(rows) * self.embedding_dim
        self.cached_idx_map[occupied] = -1
        self.inverted_cached_idx[rows] = -1
        self.cuda_available_row_num += len(rows)
        if self.freq_cnter is not None:
            self.freq_cnter[:] = MAXSIZE
        assert self.cuda_available_row_num == self.cuda_row_num
        assert np.all(self.inverted_cached_idx == -1)


# ---------------------------------------------------------------------------------------
# input generators restated from the reference's data side (used by tests and bench only
# through this oracle module when a CPU-side expected value is needed)


def power_law_ids(rng: np.random.Generator, num_rows: int, n: int, s: float = 0.25) -> np.ndarray:
    """Long-tail id generator of baselines/data/custom.py:76-93 (restated, fp64):
    u ~ U[(1/e)^s, 1], id = floor(u^(-1/s)) - 1, for a table of e rows."""
    lo = (1.0 / num_rows) ** s
    u = rng.random(n, dtype=np.float64) * (1.0 - lo) + lo
    ids = np.floor(1.0 / (u ** (1.0 / s))).astype(np.int64) - 1
    return np.clip(ids, 0, num_rows - 1)


def id_freq_map(ids: np.ndarray, num_embeddings: int) -> np.ndarray:
    """GlobalFeatureCounter.compute (recsys/datasets/feature_counter.py:21-29): bincount."""
    return np.bincount(np.asarray(ids).reshape(-1), minlength=num_embeddings).astype(np.int64)
