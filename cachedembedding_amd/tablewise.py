"""Table-wise sharding of the cached EmbeddingBag (`--use_tablewise`, recsys/dlrm_main.py:136-137):
every rank owns whole tables (recsys/models/dlrm.py:53-68, recsys/utils/misc.py:157-209; upstream
ParallelCachedEmbeddingBagTablewise).

Contract kept from the reference: the rank's loader already restricts the KJT to the rank's tables and
offsets ids into the rank's concatenated table (recsys/datasets/criteo.py:91-96,230); forward turns the local
[F_loc*B, D] into [B, F_loc*D], then one all-to-all scatters the batch and gathers the features:
[B/W, F*D] with the tables in rank-major order.

The reference hard-codes the table->rank maps for two datasets and a few world sizes
(recsys/utils/misc.py:184-209, "TODO: automatic arrange").  Here the arrangement is computed: longest-processing-
time greedy on the table sizes, which balances rows (= host memory and cache slots) per rank for any dataset /
world size.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

from .cache_mgr import EvictionStrategy
from .cached_embedding import CachedEmbeddingBag
from .parallel import dual_all_to_all


@dataclass
class TablewiseEmbeddingBagConfig:
    """recsys/utils/misc.py:174-181 builds these."""
    num_embeddings: int
    cuda_row_num: int
    assigned_rank: int = 0
    buffer_size: int = 50_000
    ids_freq_mapping: Optional[Sequence[int]] = None
    initial_weight: Optional[torch.Tensor] = None
    name: str = ""


def get_tablewise_rank_arrange(num_embeddings_per_feature: Sequence[int], world_size: int) -> List[int]:
    """table -> rank, balancing total rows per rank (largest table first onto the lightest rank)."""
    load = [0] * world_size
    arrange = [0] * len(num_embeddings_per_feature)
    for t in sorted(range(len(num_embeddings_per_feature)), key=lambda i: (-num_embeddings_per_feature[i], i)):
        r = min(range(world_size), key=lambda k: (load[k], k))
        arrange[t] = r
        load[r] += num_embeddings_per_feature[t]
    return arrange


def prepare_tablewise_config(num_embeddings_per_feature: Sequence[int], cache_ratio: float, id_freq_map_total=None,
                             dataset: Optional[str] = None, world_size: int = 2) -> List[TablewiseEmbeddingBagConfig]:
    """recsys/utils/misc.py:157-182: per-table config with cuda_row_num = int(ratio*rows) + 2000 capped at rows."""
    arrange = get_tablewise_rank_arrange(num_embeddings_per_feature, world_size)
    cfgs, off = [], 0
    for i, n in enumerate(num_embeddings_per_feature):
        freq = None if id_freq_map_total is None else id_freq_map_total[off:off + n]
        cfgs.append(TablewiseEmbeddingBagConfig(num_embeddings=n, cuda_row_num=min(n, int(cache_ratio * n) + 2000),
                                                assigned_rank=arrange[i], ids_freq_mapping=freq, name=f"t{i}"))
        off += n
    return cfgs


class ParallelCachedEmbeddingBagTablewise(nn.Module):
    def __init__(self, embedding_bag_config_list: List[TablewiseEmbeddingBagConfig], embedding_dim: int,
                 padding_idx=None, max_norm=None, norm_type=2.0, scale_grad_by_freq=False, sparse=False,
                 mode: str = "mean", include_last_offset: bool = False, dtype=None, device=None,
                 warmup_ratio: float = 0.7, buffer_size: int = 50_000, pin_weight: bool = False,
                 evict_strategy: EvictionStrategy = EvictionStrategy.LFU, group=None):
        super().__init__()
        self.group = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
        self.rank = dist.get_rank(self.group) if self.group is not None else 0
        self.world_size = dist.get_world_size(self.group) if self.group is not None else 1
        self.rank_of_tables = [c.assigned_rank for c in embedding_bag_config_list]
        self.global_tables_num = len(embedding_bag_config_list)
        self.embedding_dim = embedding_dim
        self.assigned_table_list = [i for i, r in enumerate(self.rank_of_tables) if r == self.rank]
        local = [embedding_bag_config_list[i] for i in self.assigned_table_list]
        assert local, f"rank {self.rank} owns no table"
        self.include_last_offset = include_last_offset
        self.pool_str = mode
        self.global_tables_offsets = [0]
        for c in embedding_bag_config_list:
            self.global_tables_offsets.append(self.global_tables_offsets[-1] + c.num_embeddings)
        rows = sum(c.num_embeddings for c in local)
        slots = max(1, min(rows, sum(c.cuda_row_num for c in local)))
        freq = None
        if all(c.ids_freq_mapping is not None for c in local):
            freq = torch.cat([torch.as_tensor(c.ids_freq_mapping).view(-1).long() for c in local])
        weight = None
        if all(c.initial_weight is not None for c in local):
            weight = torch.cat([c.initial_weight for c in local], dim=0)
        self.cache_weight_mgr_module = CachedEmbeddingBag(rows, embedding_dim, padding_idx, max_norm, norm_type,
                                                          scale_grad_by_freq, sparse, weight, mode, include_last_offset,
                                                          dtype, device, cuda_row_num=slots, ids_freq_mapping=freq,
                                                          warmup_ratio=warmup_ratio, buffer_size=buffer_size,
                                                          pin_weight=pin_weight, evict_strategy=evict_strategy)
        self.cache_weight_mgr = self.cache_weight_mgr_module.cache_weight_mgr
        # features per rank, for the feature gather of the output exchange
        self._feat_per_rank = [sum(1 for r in self.rank_of_tables if r == p) for p in range(self.world_size)]

    def set_cache_op(self, cache_op: bool = True):
        self.cache_weight_mgr_module.set_cache_op(cache_op)

    def set_cache_mgr_async_copy(self, flag: bool):
        self.cache_weight_mgr_module.set_cache_mgr_async_copy(flag)

    def print_comm_stats_(self):
        return self.cache_weight_mgr.print_comm_stats()

    def element_size(self) -> int:
        return self.cache_weight_mgr_module.element_size()

    def forward(self, indices: torch.Tensor, offsets: Optional[torch.Tensor] = None, per_sample_weights=None,
                shape_hook: Optional[Callable] = None, already_split_along_rank: bool = True):
        if not already_split_along_rank:
            raise NotImplementedError("pass the KJT already restricted to this rank's tables, as the reference's "
                                      "loaders do (recsys/datasets/criteo.py:91-96)")
        f_loc = len(self.assigned_table_list)
        out = self.cache_weight_mgr_module(indices, offsets, per_sample_weights)        # [F_loc*B, D]
        batch = out.shape[0] // f_loc
        local = torch.cat(out.split(batch, 0), dim=1)                                     # [B, F_loc*D]
        if self.world_size > 1:
            sizes = [f * self.embedding_dim for f in self._feat_per_rank]
            local = dual_all_to_all(local, self.group, scatter_dim=0, gather_dim=1, gather_sizes=sizes)
        return shape_hook(local) if shape_hook is not None else local
