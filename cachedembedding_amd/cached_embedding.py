"""CachedEmbeddingBag: nn.EmbeddingBag-compatible module whose table lives in pinned host
DRAM with a frequency-aware cache of hot rows in HBM.

Drop-in for ``colossalai.nn.parallel.layers.CachedEmbeddingBag`` as the reference uses it
(benchmark/benchmark_cache.py:39-40,62; benchmark/benchmark_fbgemm_uvm.py:98-105,148);
constructor order and forward signature per SURVEY.md 8(b) / Appendix A.7.
"""
from __future__ import annotations

from typing import Callable, Iterator, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .cache_mgr import CachedParamMgr, EvictionStrategy, HostTable
from .functional import FusedSGD, embedding_bag


class CachedEmbeddingBag(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None,
                 max_norm: Optional[float] = None, norm_type: float = 2.0, scale_grad_by_freq: bool = False,
                 sparse: bool = False, _weight: Optional[torch.Tensor] = None, mode: str = "mean",
                 include_last_offset: bool = False, dtype=None, device=None, cache_ratio: float = 0.01,
                 ids_freq_mapping=None, warmup_ratio: float = 0.7, buffer_size: int = 0, pin_weight: bool = False,
                 evict_strategy: EvictionStrategy = EvictionStrategy.DATASET, *, cuda_row_num: Optional[int] = None,
                 init_seed: int = 1024, strict: bool = True):
        super().__init__()
        _lib.require_gpu()
        assert cache_ratio <= 1.0, f"cache ratio {cache_ratio} must less than 1.0"
        if dtype not in (None, torch.float32):
            raise NotImplementedError("only fp32 tables are implemented")
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        if padding_idx is not None:
            if padding_idx > 0:
                assert padding_idx < num_embeddings, "Padding_idx must be within num_embeddings"
            elif padding_idx < 0:
                assert padding_idx >= -num_embeddings, "Padding_idx must be within num_embeddings"
                padding_idx = num_embeddings + padding_idx
        self.padding_idx = padding_idx
        self.max_norm = max_norm
        self.norm_type = norm_type
        self.scale_grad_by_freq = scale_grad_by_freq
        self.sparse = sparse
        self.mode = mode
        self.include_last_offset = include_last_offset
        self.evict_strategy = evict_strategy
        self.cache_ratio = cache_ratio
        self.cuda_row_num = int(num_embeddings * cache_ratio) if cuda_row_num is None else int(cuda_row_num)
        self.pool_str = mode
        self.cache_op = True
        self.fused_sgd = FusedSGD(None)

        if _weight is None:
            table = HostTable.allocate(num_embeddings, embedding_dim)
            table.fill_uniform_(-1.0 / num_embeddings, 1.0 / num_embeddings, init_seed)
            if padding_idx is not None:
                table.tensor[padding_idx].zero_()
        else:
            w = _weight.detach()
            assert tuple(w.shape) == (num_embeddings, embedding_dim)
            if w.device.type != "cpu" or w.dtype != torch.float32 or not w.is_contiguous():
                w = w.to("cpu", torch.float32).contiguous()
            table = HostTable.wrap(w)
        self.cache_weight_mgr = CachedParamMgr(table, self.cuda_row_num, buffer_size, pin_weight,
                                               evict_strategy=evict_strategy, device=device, strict=strict)
        self.cache_weight_mgr.reorder(ids_freq_mapping, warmup_ratio)

    # -- the host table (upstream `.weight`) and the parameter protocol (A.7) ------------
    @property
    def weight(self) -> torch.Tensor:
        return self.cache_weight_mgr.weight

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True
                         ) -> Iterator[Tuple[str, nn.Parameter]]:
        yield (prefix + ("." if prefix else "") + "weight", self.cache_weight_mgr.cuda_cached_weight)

    def parameters(self, recurse: bool = True) -> Iterator[nn.Parameter]:
        yield self.cache_weight_mgr.cuda_cached_weight

    def set_cache_op(self, cache_op: bool = True):
        self.cache_op = cache_op

    def set_cache_mgr_async_copy(self, flag: bool):
        self.cache_weight_mgr.set_async_copy(flag)

    def set_fused_sgd(self, lr: Optional[float], deterministic: bool = False):
        """Apply SGD(lr) to the cache rows inside backward (K13+K14 fused).  lr=None restores
        the plain autograd behaviour (grad handed to torch.optim)."""
        self.fused_sgd.lr = lr
        self.fused_sgd.deterministic = deterministic

    def forward(self, input: torch.Tensor, offsets: Optional[torch.Tensor] = None,
                per_sample_weights: Optional[torch.Tensor] = None, shape_hook: Optional[Callable] = None,
                *, hook_features: int = 0, presorted: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        # out (addition): the pooled output is written into this tensor (functional.embedding_bag(out=...))
        masked = False
        if self.cache_op:
            with torch.no_grad():
                ids = input
                input = self.cache_weight_mgr.prepare_ids(ids)
                if self.padding_idx is not None:
                    if presorted is not None:
                        raise NotImplementedError("padding_idx cannot be combined with presorted keys: they were "
                                                  "built from slots that still hold the padding lookups")
                    masked = True
                    # nn.EmbeddingBag semantics in ID space (upstream hands padding_idx to F.embedding_bag in slot
                    # space, which is not meaningful): lookups of the padding id take no part in the
                    # reduction and get no gradient -- the kernels skip slot -1.  With cache_op=False the caller
                    # passes slots and masks them itself.
                    input = torch.where(ids == self.padding_idx, torch.full_like(input, -1), input)
        out = embedding_bag(input, self.cache_weight_mgr.cuda_cached_weight, offsets, self.max_norm,
                            self.norm_type, self.scale_grad_by_freq, self.mode, self.sparse, per_sample_weights,
                            self.include_last_offset, None, hook_features=hook_features,
                            fused_sgd=self.fused_sgd, presorted=presorted, masked_indices=masked, out=out)
        if shape_hook is not None:
            out = shape_hook(out)
        return out

    # -- observability surface (recsys/dlrm_main.py:286-294) -----------------------------
    @property
    def num_hits_history(self) -> List[int]:
        self.cache_weight_mgr.sync_stats()
        return self.cache_weight_mgr.num_hits_history

    @property
    def num_miss_history(self) -> List[int]:
        self.cache_weight_mgr.sync_stats()
        return self.cache_weight_mgr.num_miss_history

    @property
    def num_write_back_history(self) -> List[int]:
        self.cache_weight_mgr.sync_stats()
        return self.cache_weight_mgr.num_write_back_history

    def print_comm_stats_(self):
        return self.cache_weight_mgr.print_comm_stats()

    def element_size(self) -> int:
        return self.weight.element_size()

    def flush(self):
        self.cache_weight_mgr.flush()

    @classmethod
    def from_pretrained(cls, embeddings: torch.Tensor, freeze: bool = True, **kwargs) -> "CachedEmbeddingBag":
        rows, cols = embeddings.shape
        m = cls(rows, cols, _weight=embeddings, **kwargs)
        m.cache_weight_mgr.cuda_cached_weight.requires_grad_(not freeze)
        return m
