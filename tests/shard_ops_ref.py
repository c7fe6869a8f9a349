"""Reference ShardOps for the CPU (gloo) tests of the row-wise exchange logic: torch ops on CPU
tensors + the oracle cache manager.  Test infrastructure only -- the product uses HipShardOps."""
import numpy as np
import torch
import torch.nn.functional as F

from cachedembedding_amd.parallel import ShardOps
from oracle.cache_oracle import DATASET, OracleCachedParamMgr


class TorchShardOps(ShardOps):
    def __init__(self, weight_shard: np.ndarray, cuda_row_num: int, idx_map, world: int, rank: int):
        self.mgr = OracleCachedParamMgr(weight_shard, cuda_row_num, DATASET)
        self.mgr.reorder(None, 0.7)
        self.idx_map = idx_map
        self.world, self.rank = world, rank
        self.dim = weight_shard.shape[1]

    def bucketize(self, ids):
        ids = ids.reshape(-1).long()
        rows_all = self.idx_map[ids].long() if self.idx_map is not None else ids
        rows, inv = torch.unique(rows_all, return_inverse=True)       # only unique rows travel
        owner = rows % self.world
        order = torch.argsort(owner, stable=True)              # stable counting sort by owner
        perm = torch.empty_like(order)
        perm[order] = torch.arange(rows.numel())
        counts = torch.bincount(owner, minlength=self.world).long()
        return (rows[order] // self.world).contiguous(), perm[inv], counts

    def owner_prepare(self, local_rows):
        return torch.from_numpy(self.mgr.prepare_ids(local_rows.numpy()))

    def owner_gather(self, slots):
        return torch.from_numpy(self.mgr.cuda_cached_weight[slots.numpy()].copy())

    def pool(self, rows, perm, offsets, psw, mode, include_last, hook_features):
        out = F.embedding_bag(perm, rows, offsets.long(), mode=mode, per_sample_weights=psw,
                              include_last_offset=include_last)
        if hook_features:
            out = out.view(hook_features, -1, self.dim).transpose(0, 1).contiguous()
        return out

    def grad_rows(self, grad_out, perm, offsets, psw, mode, include_last, hook_features, n, keys=None):
        if hook_features:
            grad_out = grad_out.transpose(0, 1).reshape(-1, self.dim)
        rows = torch.zeros(n, self.dim, requires_grad=True)
        F.embedding_bag(perm, rows, offsets.long(), mode=mode, per_sample_weights=psw,
                        include_last_offset=include_last).backward(grad_out)
        return rows.grad

    def owner_update(self, slots, grad_rows, lr):
        w = torch.from_numpy(self.mgr.cuda_cached_weight)
        w.index_add_(0, slots, grad_rows, alpha=-lr)
