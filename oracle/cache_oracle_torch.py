"""CPU ORACLE #2 (test infrastructure -- never imported by the product path).

An OP-SEQUENCE-LITERAL torch-CPU restatement of the cache manager that hpcaitech/CachedEmbedding drives through
``colossalai.nn.parallel.layers.CachedParamMgr`` (reference call sites
recsys/dlrm_main.py:259, benchmark/benchmark_cache.py:62, recsys/models/dlrm.py:70-81).

Where ``oracle/cache_oracle.py`` restates the algorithm with sets and lexsorts in numpy, this file follows upstream's
tensor-op sequence step by step -- ``torch.unique(..., return_counts=True)`` -> ``torch.isin`` -> ``index_fill_`` of
the protected slots -> ``topk`` -> ``index_select`` / ``index_copy_`` write-back -> ``nonzero(cached_idx_map == -1)
[:m]`` -> ``index_copy_`` admit -> ``index_select(inverted, index_select(idx_map, ids))`` -> ``scatter_add_`` -- so a
divergence between "what the set-based restatement means" and "what the op sequence does" (masking order, which
tensor is restored when, what a fresh admit's counter is during selection) shows up as a test failure in
tests/test_oracle.py, which replays the golden streams and random streams through both.

PARITY UNPINNED by the reference itself (see oracle/cache_oracle.py): ColossalAI's source is not in /root/reference
and cannot be installed here, so both files are restatements; this one removes the degree of freedom that is the
restatement's own data structures.

The only deliberate deviation from the literal ops: ``torch.topk`` leaves tie order implementation-defined (probed
in SURVEY.md Appendix B#1), so ``topk(x, k, largest)`` is realised as the first k entries of a STABLE sort -- the
canonical tie rule (LFU: freq ascending then slot ascending; DATASET: keys are unique).  ``argsort`` in ``reorder``
is the stable one for the same reason (B#2).
"""
from __future__ import annotations

import math
import sys
from typing import List, Optional

import torch

MAXSIZE = sys.maxsize


def _topk_stable(x: torch.Tensor, k: int, largest: bool) -> torch.Tensor:
    """indices of torch.topk(x, k, largest=largest) under the canonical tie rule (ties -> lower index first)"""
    order = torch.sort(x, descending=largest, stable=True).indices
    return order[:k]


class TorchCachedParamMgr:
    """Field names follow upstream: weight, cuda_cached_weight, idx_map, cached_idx_map, inverted_cached_idx,
    freq_cnter, _cuda_available_row_num, evict_backlist, num_hits_history, num_miss_history, num_write_back_history."""

    def __init__(self, weight: torch.Tensor, cuda_row_num: int, evict_strategy: str = "dataset"):
        if cuda_row_num == 0:
            raise NotImplementedError("cuda_row_num == 0")                            # A.1
        self.weight = weight                                                          # host table [N, D]
        self.num_embeddings, self.embedding_dim = weight.shape
        self.cuda_row_num = int(cuda_row_num)
        self.lfu = evict_strategy == "lfu"
        N, C = self.num_embeddings, self.cuda_row_num
        self.cuda_cached_weight = torch.zeros(C, self.embedding_dim, dtype=weight.dtype)
        self.idx_map = torch.arange(N, dtype=torch.long)
        self.cached_idx_map = torch.empty(C, dtype=torch.long).fill_(-1)
        self.inverted_cached_idx = torch.zeros(N, dtype=torch.long).fill_(-1)
        self.freq_cnter = torch.empty(C, dtype=torch.long).fill_(MAXSIZE) if self.lfu else None
        self._cuda_available_row_num = C
        self.evict_backlist = torch.tensor([], dtype=torch.long)
        self.num_hits_history: List[int] = []
        self.num_miss_history: List[int] = []
        self.num_write_back_history: List[int] = []
        self._cpu_to_cuda_numel = 0
        self._cuda_to_cpu_numel = 0
        self._cache_miss = 0
        self._total_cache = 0
        self.last_evicted_rows = torch.tensor([], dtype=torch.long)
        # build extension mirrored from oracle/cache_oracle.py: rows of the previous `protect_depth` calls stay in
        # the backlist (the overlapped prefetch pipeline)
        self.protect_depth = 0
        self._backlists: List[torch.Tensor] = []

    # ------------------------------------------------------------------ A.2
    def reorder(self, ids_freq_mapping=None, warmup_ratio: float = 0.7):
        N, C = self.num_embeddings, self.cuda_row_num
        tmp_idx = None
        freq = None
        if ids_freq_mapping is not None:
            freq = torch.as_tensor(ids_freq_mapping, dtype=torch.long)
            tmp_idx = torch.argsort(freq, descending=True, stable=True)               # B#2: stable
            if not self.lfu:
                sorted_idx = torch.argsort(tmp_idx, stable=True)                      # argsort(argsort(desc))
                self.idx_map.data.copy_(sorted_idx)
        preload_row_num = min(int(math.ceil(C * warmup_ratio)), N)
        if preload_row_num > 0:
            if self.lfu and freq is not None:
                preload_row_ids = tmp_idx[:preload_row_num]                           # topk(freq, n), canonical ties
                freq_value = freq.index_select(0, preload_row_ids)
            else:
                preload_row_ids = torch.arange(preload_row_num)
                freq_value = None
            preload_slot_ids = torch.arange(preload_row_num)
            preload_rows = self.weight.index_select(0, preload_row_ids)
            self.cuda_cached_weight.index_copy_(0, preload_slot_ids, preload_rows)
            self.cached_idx_map.index_copy_(0, preload_slot_ids, preload_row_ids)
            self.inverted_cached_idx.index_copy_(0, preload_row_ids, preload_slot_ids)
            self._cuda_available_row_num -= preload_row_num
            if self.lfu:
                if freq_value is None:
                    self.freq_cnter.index_fill_(0, preload_slot_ids, 0)
                else:
                    self.freq_cnter.index_copy_(0, preload_slot_ids, freq_value)

    # ------------------------------------------------------------------ A.6
    def _id_to_cached_cuda_id(self, ids: torch.Tensor) -> torch.Tensor:
        ids = self.idx_map.index_select(0, ids.view(-1))
        return self.inverted_cached_idx.index_select(0, ids)

    # ------------------------------------------------------------------ A.3
    def prepare_ids(self, ids: torch.Tensor) -> torch.Tensor:
        ids = torch.as_tensor(ids, dtype=torch.long).view(-1)
        if self.lfu:
            cpu_row_idxs, repeat_times = torch.unique(ids, return_counts=True)
        else:
            cpu_row_idxs, repeat_times = torch.unique(self.idx_map.index_select(0, ids), return_counts=True)
        assert len(cpu_row_idxs) <= self.cuda_row_num, (
            f"You move {len(cpu_row_idxs)} embedding rows from CPU to CUDA. {self.cuda_row_num} rows are available "
            "on CUDA. Please increase cuda_row_num or decrease the training batch size.")
        self.evict_backlist = cpu_row_idxs
        if self.protect_depth > 0 and self._backlists:
            self.evict_backlist = torch.unique(torch.cat([cpu_row_idxs] + self._backlists[-self.protect_depth:]))
        tmp = torch.isin(cpu_row_idxs, self.cached_idx_map, invert=True)
        comm_cpu_row_idxs = cpu_row_idxs[tmp]
        self._cache_miss += int(torch.sum(repeat_times[tmp]))
        self._total_cache += ids.numel()
        self.num_hits_history.append(len(cpu_row_idxs) - len(comm_cpu_row_idxs))
        self.num_miss_history.append(len(comm_cpu_row_idxs))
        self.num_write_back_history.append(0)
        self.last_evicted_rows = torch.tensor([], dtype=torch.long)
        if comm_cpu_row_idxs.numel() > 0:
            self._prepare_rows_on_cuda(comm_cpu_row_idxs)
        if self.protect_depth > 0:
            self._backlists = (self._backlists + [cpu_row_idxs.clone()])[-self.protect_depth:]
        self.evict_backlist = torch.tensor([], dtype=torch.long)
        gpu_row_idxs = self._id_to_cached_cuda_id(ids)
        if self.lfu:
            unique_gpu_row_idxs = self.inverted_cached_idx[cpu_row_idxs]
            self.freq_cnter.scatter_add_(0, unique_gpu_row_idxs, repeat_times)
        return gpu_row_idxs

    # ------------------------------------------------------------------ A.5
    def _find_evict_gpu_idxs(self, evict_num: int) -> torch.Tensor:
        mask_cpu_row_idx = torch.isin(self.cached_idx_map, self.evict_backlist)
        backup_idxs = self.cached_idx_map[mask_cpu_row_idx].clone()
        invalid_idxs = torch.nonzero(mask_cpu_row_idx).squeeze(1)
        if not self.lfu:
            self.cached_idx_map.index_fill_(0, invalid_idxs, -2)
            evict_gpu_row_idxs = _topk_stable(self.cached_idx_map, evict_num, largest=True)
            self.cached_idx_map.index_copy_(0, invalid_idxs, backup_idxs)
            assert bool((self.cached_idx_map[evict_gpu_row_idxs] >= 0).all())
        else:
            backup_freqs = self.freq_cnter[invalid_idxs].clone()
            self.freq_cnter.index_fill_(0, invalid_idxs, MAXSIZE)
            evict_gpu_row_idxs = _topk_stable(self.freq_cnter, evict_num, largest=False)
            self.freq_cnter.index_copy_(0, invalid_idxs, backup_freqs)
            # (empty slots carry sys.maxsize from construction / flush, so they are never selected while enough
            #  occupied, unprotected slots exist -- which the assertion in prepare_ids guarantees)
            assert bool((self.cached_idx_map[evict_gpu_row_idxs] >= 0).all())
        return evict_gpu_row_idxs

    # ------------------------------------------------------------------ A.4
    def _prepare_rows_on_cuda(self, cpu_row_idxs: torch.Tensor) -> None:
        evict_num = cpu_row_idxs.numel() - self._cuda_available_row_num
        if evict_num > 0:
            evict_gpu_row_idxs = self._find_evict_gpu_idxs(evict_num)
            evict_info = self.cached_idx_map[evict_gpu_row_idxs]
            rows = self.cuda_cached_weight.index_select(0, evict_gpu_row_idxs)
            self.weight.index_copy_(0, evict_info, rows)                              # always written back (B#4)
            self.cached_idx_map.index_fill_(0, evict_gpu_row_idxs, -1)
            self.inverted_cached_idx.index_fill_(0, evict_info, -1)
            self._cuda_available_row_num += evict_num
            self._cuda_to_cpu_numel += evict_num * self.embedding_dim
            self.num_write_back_history[-1] += evict_num
            self.last_evicted_rows = evict_info.clone()
        slots = torch.nonzero(self.cached_idx_map == -1).squeeze(1)[:cpu_row_idxs.numel()]
        rows = self.weight.index_select(0, cpu_row_idxs)
        self.cuda_cached_weight.index_copy_(0, slots, rows)
        self.cached_idx_map.index_copy_(0, slots, cpu_row_idxs)
        self.inverted_cached_idx.index_copy_(0, cpu_row_idxs, slots)
        if self.lfu:
            self.freq_cnter.index_fill_(0, slots, 0)
        self._cuda_available_row_num -= cpu_row_idxs.numel()
        self._cpu_to_cuda_numel += cpu_row_idxs.numel() * self.embedding_dim

    # ------------------------------------------------------------------ A.7
    def flush(self) -> None:
        slots = torch.nonzero(self.cached_idx_map > -1).squeeze(1)
        row_ids = self.cached_idx_map[slots]
        rows = self.cuda_cached_weight.index_select(0, slots)
        self.weight.index_copy_(0, row_ids, rows)
        self.cached_idx_map.index_fill_(0, slots, -1)
        self.inverted_cached_idx.index_fill_(0, row_ids, -1)
        self._cuda_available_row_num += slots.numel()
        if self.lfu:
            self.freq_cnter.fill_(MAXSIZE)
        assert self._cuda_available_row_num == self.cuda_row_num
        assert bool(torch.all(self.inverted_cached_idx == -1))
        assert bool(torch.all(self.cached_idx_map == -1))
