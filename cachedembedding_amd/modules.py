"""Wiring either side of the operator, mirroring the reference's own glue for this path:

* `FiniteDataIter` / `CudaStreamDataIter` -- one-batch-ahead host->HBM copy on a side HIP stream with
  wait_stream + record_stream on consume (recsys/utils/dataloader/cuda_stream_dataloader.py:11-82,
  base class recsys/utils/dataloader/base_dataiter.py:10-83): the "Stream1" lane of pics/prefetch.png.
* `FusedSparseModules` -- the module the DLRM model calls (recsys/models/dlrm.py:32-113): picks the
  embedding operator, sets cache_op, optionally gathers the KJT across ranks, dispatches list-vs-KJT input
  and applies the shape hook (recsys/models/dlrm.py:26-30).

A sparse batch here is the `[values, offsets, stride]` list `_train` builds (recsys/dlrm_main.py:253) or
any object with `.values() .offsets() .stride()` (torchrec's KeyedJaggedTensor quacks like that).
"""
from __future__ import annotations

from typing import Any, Iterable, List, Optional, Sequence, Union

import torch
import torch.nn as nn

from .cache_mgr import EvictionStrategy
from .parallel import KJTAllToAll, ParallelCachedEmbeddingBag


def sparse_embedding_shape_hook(embeddings: torch.Tensor, feature_size: int, batch_size: int) -> torch.Tensor:
    """[F*B, D] -> [B, F, D] view (recsys/models/dlrm.py:26-27)."""
    return embeddings.view(feature_size, batch_size, -1).transpose(0, 1)


def sparse_embedding_shape_hook_for_tablewise(embeddings: torch.Tensor, feature_size: int, batch_size: int):
    """recsys/models/dlrm.py:29-30"""
    return embeddings.view(embeddings.shape[0], feature_size, -1)


def _to_device(obj: Any, device, non_blocking: bool = True):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=non_blocking)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(o, device, non_blocking) for o in obj)
    if isinstance(obj, dict):
        return {k: _to_device(v, device, non_blocking) for k, v in obj.items()}
    if hasattr(obj, "to"):
        return obj.to(device, non_blocking=non_blocking)
    return obj


def _record_stream(obj: Any, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record_stream(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record_stream(o, stream)
    elif hasattr(obj, "record_stream"):
        obj.record_stream(stream)


class CudaStreamDataIter:
    """Infinite iterator: wraps a loader, restarts it when exhausted, keeps one batch in flight on a side stream
    (cuda_stream_dataloader.py:11-47)."""

    def __init__(self, loader: Iterable, device=None):
        self.loader = loader
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self.iter = iter(loader)
        self.batch_data = None
        self._preload()

    def _next_host(self):
        try:
            return next(self.iter)
        except StopIteration:
            self.iter = iter(self.loader)
            return next(self.iter)

    def _preload(self):
        host = self._next_host()
        with torch.cuda.stream(self.stream):
            self.batch_data = _to_device(host, self.device, non_blocking=True)

    def __iter__(self):
        return self

    def __next__(self):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        data = self.batch_data
        _record_stream(data, cur)
        self._preload()
        return data


class FiniteDataIter(CudaStreamDataIter):
    """One pass over the loader; StopIteration ends the epoch like recsys/dlrm_main.py:292-295 expects
    (cuda_stream_dataloader.py:50-82)."""

    def _preload(self):
        try:
            host = next(self.iter)
        except StopIteration:
            self.batch_data = None
            return
        with torch.cuda.stream(self.stream):
            self.batch_data = _to_device(host, self.device, non_blocking=True)

    def __next__(self):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        data = self.batch_data
        if data is None:
            raise StopIteration
        _record_stream(data, cur)
        self._preload()
        return data


class FusedSparseModules(nn.Module):
    """recsys/models/dlrm.py:32-113 over the HIP operator.  fold_hook=True writes [B, F, D] directly from the
    gather kernel instead of creating the transposed view (same values, contiguous)."""

    def __init__(self, num_embeddings_per_feature: Sequence[int], embedding_dim: int, fused_op: str = "all_to_all",
                 reduction_mode: str = "sum", sparse: bool = False, output_device_type=None, use_cache: bool = False,
                 cache_ratio: float = 0.01, id_freq_map=None, warmup_ratio: float = 0.7, buffer_size: int = 50_000,
                 is_dist_dataloader: bool = True, use_lfu_eviction: bool = False, use_tablewise_parallel: bool = False,
                 dataset: Optional[str] = None, fold_hook: bool = False, group=None):
        super().__init__()
        self.sparse_feature_num = len(num_embeddings_per_feature)
        self.fold_hook = fold_hook
        if not use_cache:
            raise NotImplementedError("Other EmbeddingBags are under development")       # dlrm.py:83-84
        strategy = EvictionStrategy.LFU if use_lfu_eviction else EvictionStrategy.DATASET
        if use_tablewise_parallel:
            from .tablewise import ParallelCachedEmbeddingBagTablewise, prepare_tablewise_config
            world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1
            cfgs = prepare_tablewise_config(num_embeddings_per_feature, 0.01, id_freq_map, dataset, world)
            self.embed = ParallelCachedEmbeddingBagTablewise(cfgs, embedding_dim, sparse=sparse, mode=reduction_mode,
                                                             include_last_offset=True, warmup_ratio=warmup_ratio,
                                                             buffer_size=buffer_size, evict_strategy=strategy, group=group)
            self.shape_hook = sparse_embedding_shape_hook_for_tablewise
            self.fold_hook = False
        else:
            self.embed = ParallelCachedEmbeddingBag(sum(num_embeddings_per_feature), embedding_dim, sparse=sparse,
                                                    mode=reduction_mode, include_last_offset=True,
                                                    cache_ratio=cache_ratio, ids_freq_mapping=id_freq_map,
                                                    warmup_ratio=warmup_ratio, buffer_size=buffer_size,
                                                    evict_strategy=strategy, group=group)
            self.shape_hook = sparse_embedding_shape_hook
        dist_on = torch.distributed.is_initialized()
        self.kjt_collector = KJTAllToAll(group) if (is_dist_dataloader and dist_on) else None

    def forward(self, sparse_features: Union[List, Any], cache_op: bool = True, presorted=None) -> torch.Tensor:
        # presorted (an addition; fold_hook=True, cache_op=False): the window keys of this batch from
        # pipeline.PrefetchWindow(presort=True, bag_layout=...) -- forward and fused backward then run from them
        self.embed.set_cache_op(cache_op)
        if isinstance(sparse_features, list):
            values, offsets, batch_size = sparse_features[0], sparse_features[1], sparse_features[2]
        elif all(hasattr(sparse_features, a) for a in ("values", "offsets", "stride")):
            values, offsets, batch_size = sparse_features.values(), sparse_features.offsets(), sparse_features.stride()
        else:
            raise TypeError(type(sparse_features))
        if self.kjt_collector is not None and self.kjt_collector.world_size > 1:
            lengths = (offsets[1:] - offsets[:-1]).to(torch.int32)
            values, lengths = self.kjt_collector.all_to_all(values, lengths, self.sparse_feature_num)
            offsets = torch.cat([lengths.new_zeros(1), torch.cumsum(lengths, 0)]).to(offsets.dtype)
            batch_size = batch_size * self.kjt_collector.world_size
        F = self.sparse_feature_num
        if self.fold_hook:
            return self.embed(values, offsets, hook_features=F, presorted=presorted)
        assert presorted is None, "window keys need fold_hook=True"
        return self.embed(values, offsets, shape_hook=lambda x: self.shape_hook(x, F, batch_size))
