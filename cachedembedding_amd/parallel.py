"""Multi-GPU forms of the cached EmbeddingBag (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

* ROW-WISE sharding (BASELINE.json north_star) -- the build's own design.  After the
  frequency re-rank, row r is owned by rank r % W as local row r // W (interleaving keeps every
  shard's hot set equal).  Per window: lookups are bucketed by owner on the device
  (ce_bucketize_rows), counts then ids travel in one all-to-all-v each; per step the owner gathers
  the requested cache rows, one all-to-all-v returns them, the requester pools them per bag; the
  backward sends per-lookup gradient rows the reverse way and the owner applies the fused SGD update.
  This replaces the reference's KJT all-gather (recsys/datasets/utils.py:20-54) + column-wise
  dual_all_to_all: every GPU touches only B_loc*F lookups instead of the global batch, and the
  all-to-all uses all 7 xGMI links of a GPU concurrently.

* COLUMN-WISE sharding -- the contract of the reference's ParallelCachedEmbeddingBag
  (recsys/models/dlrm.py:70-81): every rank holds all rows x D/W columns, looks up the
  GLOBAL batch, and dual_all_to_all turns [B_glob, F, D/W] into [B_glob/W, F, D].

The communication/layout logic (RowwiseExchange) is device-agnostic and takes the local
compute as a `ShardOps` object; the product wires in HipShardOps (C ABI, no fallback).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .cache_mgr import CachedParamMgr, EvictionStrategy, HostTable
from .cached_embedding import CachedEmbeddingBag
from .functional import _MODES, FORWARD_FROM_KEYS, SrcKeys


def get_partition(embedding_dim: int, rank: int, world_size: int) -> Tuple[int, int, bool]:
    """Column split rule of the reference (recsys/utils/misc.py:138-154 == torch.tensor_split)."""
    if world_size == 1:
        return 0, embedding_dim, True
    assert embedding_dim >= world_size, \
        f"Embedding dimension {embedding_dim} must be larger than the world size {world_size} of the process group"
    chunk, rem = divmod(embedding_dim, world_size)
    if rem == 0:
        return rank * chunk, (rank + 1) * chunk, True
    sizes = [chunk + 1 if i < rem else chunk for i in range(world_size)]
    off = sum(sizes[:rank])
    return off, off + sizes[rank], False


# ----------------------------------------------------------------------------------------------
# dual_all_to_all: forward scatters dim `scatter_dim`, gathers `gather_dim`;
# backward is the same exchange with the dims swapped.


def _a2a(out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits, group) -> None:
    """all_to_all_single; with the gloo backend (CPU tests, or several ranks sharing one GPU in the GPU
    tests) device tensors are staged through host memory because gloo only moves CPU buffers."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def _split_sizes(n: int, world: int) -> List[int]:
    chunk, rem = divmod(n, world)
    return [chunk + 1 if i < rem else chunk for i in range(world)]        # torch.tensor_split rule


def _all_to_all_dims(x: torch.Tensor, group, scatter_dim: int, gather_dim: int, scatter_sizes: List[int],
                     gather_sizes: List[int]) -> torch.Tensor:
    """Split x along scatter_dim into `scatter_sizes`, send piece p to peer p; receive from peer p a piece
    whose gather_dim extent is gather_sizes[p]; concatenate along gather_dim.  One all-to-all-v on flat
    buffers (the only all-to-all form gloo offers, used by the CPU tests)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ins = [t.contiguous() for t in torch.split(x, scatter_sizes, dim=scatter_dim)]
    shapes = []
    for p in range(world):
        shp = list(ins[rank].shape)
        shp[gather_dim] = gather_sizes[p]
        shapes.append(shp)
    out_splits = [int(torch.Size(s).numel()) for s in shapes]
    flat_in = torch.cat([t.reshape(-1) for t in ins])
    flat_out = torch.empty(sum(out_splits), dtype=x.dtype, device=x.device)
    _a2a(flat_out, flat_in, out_splits, [t.numel() for t in ins], group)
    outs = [c.view(s) for c, s in zip(torch.split(flat_out, out_splits), shapes)]
    return torch.cat(outs, dim=gather_dim)


class _DualAllToAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group, scatter_dim, gather_dim, gather_sizes):
        world = dist.get_world_size(group)
        scatter_sizes = _split_sizes(x.shape[scatter_dim], world)
        ctx.group, ctx.sd, ctx.gd = group, scatter_dim, gather_dim
        ctx.scatter_sizes, ctx.gather_sizes = scatter_sizes, gather_sizes
        return _all_to_all_dims(x, group, scatter_dim, gather_dim, scatter_sizes, gather_sizes)

    @staticmethod
    def backward(ctx, g):
        # the transpose exchange: no size negotiation needed, both size lists are known from forward
        out = _all_to_all_dims(g.contiguous(), ctx.group, ctx.gd, ctx.sd, ctx.gather_sizes, ctx.scatter_sizes)
        return out, None, None, None, None


def dual_all_to_all(x: torch.Tensor, group=None, scatter_dim: int = 0, gather_dim: int = -1,
                    gather_sizes: Optional[List[int]] = None) -> torch.Tensor:
    """forward scatters `scatter_dim`, gathers `gather_dim`; backward swaps the dims.
    gather_sizes[p] = extent of peer p's tensor along gather_dim (None: exchanged with an all_gather)."""
    group = group if group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    if world == 1:
        return x
    sd = scatter_dim % x.dim()
    gd = gather_dim % x.dim()
    if gather_sizes is None:
        t = torch.tensor([x.shape[gd]], dtype=torch.int64, device=x.device)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
        gather_sizes = [int(o.item()) for o in out]
    return _DualAllToAll.apply(x, group, sd, gd, list(gather_sizes))


# ----------------------------------------------------------------------------------------------
# KJT collective of the reference (recsys/datasets/utils.py:8-54): an all-gather of (lengths, values)
# whose output is ordered [key][rank][sample].  Done with all_gather (no cloned send lists) and ONE
# host sync for the per-rank value counts.


class KJTAllToAll:
    def __init__(self, group=None):
        self.group = group if group is not None else dist.group.WORLD
        self.rank = dist.get_rank(self.group)
        self.world_size = dist.get_world_size(self.group)

    @torch.no_grad()
    def all_to_all(self, values: torch.Tensor, lengths: torch.Tensor, num_keys: int):
        """values int64[sum(lengths)] and lengths int32[num_keys * B_loc] of the local KJT (key-major)
        -> (all_values, all_lengths) of the global batch, key-major with samples ordered [rank][sample]."""
        W = self.world_size
        if W == 1:
            return values, lengths
        b_loc = lengths.numel() // num_keys
        all_len = [torch.empty_like(lengths) for _ in range(W)]
        dist.all_gather(all_len, lengths, group=self.group)
        per_key = torch.stack([l.view(num_keys, b_loc).sum(dim=1) for l in all_len])      # [W, K]
        per_key_host = per_key.cpu()
        totals = per_key_host.sum(dim=1).tolist()
        n_max = max(totals)
        pad = torch.zeros(n_max, dtype=values.dtype, device=values.device)
        pad[:values.numel()] = values
        all_val = [torch.empty_like(pad) for _ in range(W)]
        dist.all_gather(all_val, pad, group=self.group)
        pieces = []
        starts = torch.cumsum(per_key_host, dim=1) - per_key_host
        for k in range(num_keys):
            for r in range(W):
                s, n = int(starts[r, k]), int(per_key_host[r, k])
                pieces.append(all_val[r][s:s + n])
        all_values = torch.cat(pieces)
        all_lengths = torch.cat([l.view(num_keys, b_loc) for l in all_len], dim=1).reshape(-1)
        return all_values, all_lengths


# ----------------------------------------------------------------------------------------------
# row-wise sharding


class ShardOps:
    """Local compute of one shard.  HipShardOps is the product implementation."""

    world: int
    rank: int
    dim: int

    def bucketize(self, ids: torch.Tensor):
        """-> (local_rows[n_u] of the batch's UNIQUE rows in owner-bucket order, pos[n]: lookup j -> position of
        its row in that list, counts[W]).  Only unique rows travel: a Criteo-shaped batch of 425,984 lookups
        holds ~40 k distinct rows, so the all-to-all payload shrinks ~10x and the requester expands locally."""
        raise NotImplementedError

    # Window form.  The three calls let an implementation run the whole bucketing of a window without a host
    # wait: launch -> device-side counts (fixed-size message, all-to-all'ed by the caller) -> results once the
    # caller has read the counts back.  The defaults wrap the synchronous single-batch `bucketize`.
    def bucketize_launch(self, ids_list: Sequence[torch.Tensor]):
        return [self.bucketize(ids) for ids in ids_list]

    def bucketize_counts(self, token) -> torch.Tensor:
        """int64[P, W]: unique rows of batch b owned by rank w"""
        return torch.stack([t[2] for t in token])

    def bucketize_results(self, token, counts_host: Sequence[Sequence[int]]):
        """-> [(local_rows[n_u], pos[n], counts)] per batch; counts_host = bucketize_counts on the host"""
        return list(token)

    def token_tensors(self, token) -> List[torch.Tensor]:
        return [x for t in token for x in t if torch.is_tensor(x)]

    def bucketize_many(self, ids_list: Sequence[torch.Tensor]):
        token = self.bucketize_launch(ids_list)
        return self.bucketize_results(token, self.bucketize_counts(token).tolist())

    def owner_prepare(self, local_rows: torch.Tensor):       # -> slots[n_recv]
        raise NotImplementedError

    def owner_gather(self, slots: torch.Tensor):             # -> rows fp32[n_recv, D]
        raise NotImplementedError

    def pool(self, rows, pos, offsets, psw, mode, include_last, hook_features):   # -> pooled
        raise NotImplementedError

    def grad_rows(self, grad_out, pos, offsets, psw, mode, include_last, hook_features, n_u, keys=None):
        """-> fp32[n_u, D]: gradient of every unique row (duplicates of the batch already summed).  `keys`: what
        the implementation's bucketize returned as 4th element (grouping of the lookups by pos), if anything."""
        raise NotImplementedError

    def owner_update(self, slots, grad_rows, lr):            # cache rows -= lr * grad (duplicates summed)
        raise NotImplementedError


class HipShardOps(ShardOps):
    """ShardOps over libce_hip.so: ce_bucketize_rows, the local CachedParamMgr, ce_bag_* kernels."""

    def __init__(self, mgr: CachedParamMgr, idx_map: Optional[torch.Tensor], world: int, rank: int,
                 num_global_rows: Optional[int] = None):
        _lib.require_gpu()
        self.mgr = mgr
        self.idx_map = idx_map            # GLOBAL id -> global rank row (replicated), None = identity
        self.world, self.rank = world, rank
        self.dim = mgr.embedding_dim
        self._ws = None
        self.num_global_rows = num_global_rows
        self._stamp = None                # ce_dedupe_bucket_rows scratch: int32[N] x 2
        self._slot_of_row = None
        self._layout = None               # (offsets, include_last, hook_features) -> source-row keys at plan time

    def set_bag_layout(self, offsets: Optional[torch.Tensor], include_last_offset: bool = True,
                       hook_features: int = 0) -> None:
        """Static bag layout of the batches this rank trains on (every batch uses these offsets).  With it the plan
        stage groups the lookups into source-row keys (ce_bag_presort_window_src) and the fold of a batch's gradients
        (grad_rows, mode='sum' without per-sample weights) runs the streaming backward; None restores the
        (row, lookup) keys that any layout can use."""
        self._layout = None if offsets is None else (offsets.contiguous(), bool(include_last_offset),
                                                     int(hook_features))

    def bucketize_launch(self, ids_list):
        """ce_dedupe_bucket_rows per batch (2 launches each, no sync); the bucket sizes stay on the device."""
        dev = ids_list[0].device
        N, W = self.num_global_rows, self.world
        if W > 64:
            raise NotImplementedError("row-wise sharding over more than 64 ranks")
        if self._stamp is None:             # scratch of the dedupe passes; contents carry nothing across calls
            self._stamp = torch.empty(N, dtype=torch.int32, device=dev)      # (also holds the rows' places: one array)
        P = len(ids_list)
        counts = torch.empty(P, W, dtype=torch.int64, device=dev)
        n_max = max(int(ids.numel()) for ids in ids_list)
        if self._ws is None or self._ws.numel() < (W + 1) * n_max:
            self._ws = torch.empty(max((W + 1) * n_max, 1 << 16), dtype=torch.int32, device=dev)
        staged = []
        sp = stream_ptr()
        for b, ids in enumerate(ids_list):
            ids = ids.reshape(-1).long().contiguous()
            n = ids.numel()
            rows = torch.empty(n, dtype=torch.int64, device=dev)
            pos = torch.empty(n, dtype=torch.int64, device=dev)
            check(lib.ce_dedupe_bucket_rows(ptr(ids), n, ptr(self.idx_map), N, W, ptr(self._stamp),
                                            ptr(self._slot_of_row), ptr(self._ws), ptr(rows), ptr(pos),
                                            counts[b].data_ptr(), sp))
            # the fold of the batch's gradients (grad_rows) groups lookups by `pos`: do the grouping here, once,
            # on the planning stream (any pos < n is a valid row of the [n_u, D] gradient buffer)
            keys = torch.empty(lib.ce_bag_presort_len(n), dtype=torch.int64, device=dev)
            lay = self._layout
            if lay is not None:
                off, incl, hookf = lay
                nb = off.numel() - 1 if incl else off.numel()
                check(lib.ce_bag_presort_window_src(ptr(pos), n, 1, max(n, 1), ptr(off),
                                                    int(off.dtype == torch.int64), 0, nb, int(incl), hookf,
                                                    ptr(keys), sp))
                keys = SrcKeys(keys, nb, incl, hookf)
            else:
                check(lib.ce_bag_presort(ptr(pos), n, max(n, 1), ptr(keys), sp))
            staged.append((rows, pos, keys))
        return ("hip", staged, counts)

    def bucketize_counts(self, token):
        return token[2]

    def bucketize_results(self, token, counts_host):
        _, staged, _counts = token
        return [(rows[:sum(c)], pos, list(c), keys) for (rows, pos, keys), c in zip(staged, counts_host)]

    def token_tensors(self, token):
        return [t.keys if isinstance(t, SrcKeys) else t for tup in token[1] for t in tup] + [token[2]]

    def bucketize(self, ids):
        return self.bucketize_many([ids])[0]

    def owner_prepare(self, local_rows):
        return self.mgr.prepare_ids(local_rows)

    def owner_gather(self, slots):
        n = slots.numel()
        w = self.mgr.cuda_cached_weight
        out = torch.empty(n, self.dim, dtype=torch.float32, device=w.device)
        if n:
            off = _arange_offsets(n, w.device)
            check(lib.ce_bag_forward(ptr(w), w.shape[0], self.dim, ptr(slots), n, ptr(off), 0, n, 1, None,
                                     _lib.CE_MODE_SUM, 0, ptr(out), stream_ptr()))
        return out

    def pool(self, rows, pos, offsets, psw, mode, include_last, hook_features):
        perm = pos
        num_bags = offsets.numel() - 1 if include_last else offsets.numel()
        if hook_features:
            out = torch.empty(num_bags // hook_features, hook_features, self.dim, dtype=torch.float32,
                              device=rows.device)
        else:
            out = torch.empty(num_bags, self.dim, dtype=torch.float32, device=rows.device)
        check(lib.ce_bag_forward(ptr(rows), rows.shape[0], self.dim, ptr(perm), perm.numel(), ptr(offsets),
                                 int(offsets.dtype == torch.int64), num_bags, int(include_last), ptr(psw),
                                 _MODES[mode], hook_features, ptr(out), stream_ptr()))
        return out

    def pool_from_keys(self, table, keys: "SrcKeys", nnz: int, out: Optional[torch.Tensor] = None):
        """pooled output of a one-id-per-bag batch straight from its source-row keys over `table` (cache + exchange
        buffer): a row is loaded once per run of equal rows (ce_bag_forward_src_keys); into `out` when given"""
        hf, nb = int(keys.hook_features), int(keys.num_bags)
        shape = (nb // hf, hf, self.dim) if hf else (nb, self.dim)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=table.device)
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError(f"out= must be a contiguous fp32 tensor of shape {shape}")
        check(lib.ce_bag_forward_src_keys(ptr(table), table.shape[0], self.dim, int(nnz), ptr(keys.keys), ptr(out),
                                          stream_ptr()))
        return out

    def grad_rows(self, grad_out, pos, offsets, psw, mode, include_last, hook_features, n_u, keys=None):
        num_bags = offsets.numel() - 1 if include_last else offsets.numel()
        g = torch.zeros(n_u, self.dim, dtype=torch.float32, device=grad_out.device)
        # duplicates of a row inside the batch are summed here, before they travel
        if isinstance(keys, SrcKeys):
            if psw is not None or mode != "sum" or (keys.num_bags, keys.include_last_offset, keys.hook_features) != \
                    (num_bags, bool(include_last), int(hook_features)):
                raise ValueError("this batch was planned with source-row keys for another bag layout / mode "
                                 "(HipShardOps.set_bag_layout)")
            check(lib.ce_bag_backward_dense_presorted_src(ptr(g), n_u, self.dim, pos.numel(),
                                                          ptr(grad_out.contiguous()), ptr(keys.keys), stream_ptr()))
        elif keys is not None:
            check(lib.ce_bag_backward_dense_presorted(ptr(g), n_u, self.dim, ptr(pos), pos.numel(), ptr(offsets),
                                                      int(offsets.dtype == torch.int64), num_bags, int(include_last),
                                                      ptr(psw), _MODES[mode], hook_features,
                                                      ptr(grad_out.contiguous()), ptr(keys), stream_ptr()))
        else:
            check(lib.ce_bag_backward_dense(ptr(g), n_u, self.dim, ptr(pos), pos.numel(), ptr(offsets),
                                            int(offsets.dtype == torch.int64), num_bags, int(include_last), ptr(psw),
                                            _MODES[mode], hook_features, ptr(grad_out.contiguous()), stream_ptr()))
        return g

    def update_table(self, table, grad_out, keys: "SrcKeys", nnz: int, lr: float):
        """table[row] -= lr * (sum of the gradient rows of its lookups) in place, lookups grouped by `keys`
        (source-row keys over `table`'s rows): the fused backward + SGD of the unsharded module on a table that
        holds the cache and, behind it, the exchange buffer (GraphedShardedWindow)."""
        check(lib.ce_bag_backward_sgd_presorted_src(ptr(table), table.shape[0], self.dim, int(nnz),
                                                    ptr(grad_out.contiguous()), float(lr), ptr(keys.keys),
                                                    stream_ptr()))

    def owner_update(self, slots, grad_rows, lr):
        n = slots.numel()
        if n == 0:
            return
        w = self.mgr.cuda_cached_weight
        # one folded gradient row per requested row: plain row axpy (atomic: peers may request the same row)
        check(lib.ce_rows_axpy(ptr(w), w.shape[0], self.dim, ptr(slots), n, ptr(grad_rows.contiguous()),
                               -float(lr), stream_ptr()))


_ARANGE = {}


def _arange_offsets(n: int, device) -> torch.Tensor:
    key = str(device)
    t = _ARANGE.get(key)
    if t is None or t.numel() < n + 1:
        t = torch.arange(0, max(n + 1, 1 << 20), dtype=torch.int32, device=device)
        _ARANGE[key] = t
    return t[:n + 1]


@dataclass
class BatchPlan:
    n: int                      # unique rows of the batch (= rows exchanged)
    perm: torch.Tensor          # int64[lookups]: lookup j -> position of its row in my bucket-ordered list
    send_splits: List[int]      # lookups I send to each peer (== rows each peer returns to me)
    recv_splits: List[int]      # lookups each peer sends me (rows I own)
    recv_rows: torch.Tensor     # local row ids I serve, peer-major
    slots: Optional[torch.Tensor] = None
    keys: Optional[torch.Tensor] = None     # grouping of the batch's lookups by perm (ShardOps-specific), or None


class RowwiseExchange:
    """Exchange logic of the row-wise sharded lookup (device-agnostic; collectives via torch.distributed)."""

    def __init__(self, ops: ShardOps, group=None):
        self.ops = ops
        self.group = group if group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        assert ops.world == self.world and ops.rank == self.rank

    @torch.no_grad()
    def plan_window(self, ids_list: Sequence[torch.Tensor]) -> List[BatchPlan]:
        """Bucket every batch of the window by owner, exchange counts and ids, then run ONE owner-side cache
        op over all rows this rank serves in the window.  = plan_begin + plan_end back to back; a pipeline
        enqueues training steps between the two so that the single host wait of a window (the exchanged
        bucket sizes) never stalls the launch thread."""
        return self.plan_end(self.plan_begin(ids_list))

    @torch.no_grad()
    def plan_begin(self, ids_list: Sequence[torch.Tensor]):
        """Everything that needs no host knowledge: dedupe + owner bucketing of the P batches, the all-to-all
        of the bucket sizes (a fixed-size [W, P] message) and the async readback of both size matrices."""
        W = self.world
        token = self.ops.bucketize_launch(ids_list)
        send = self.ops.bucketize_counts(token).t().contiguous()            # [W, P]: rows I request from w
        recv = torch.empty_like(send)                                       # [W, P]: rows w requests from me
        if W > 1:
            _a2a(recv, send, None, None, self.group)
        else:
            recv.copy_(send)
        both = torch.stack([send, recv])
        if both.is_cuda:
            host = torch.empty(both.shape, dtype=both.dtype, pin_memory=True)
            host.copy_(both, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = both, None
        return {"phase": 2, "P": len(ids_list), "token": token, "counts_host": host, "counts_event": ev,
                "keep": both}

    @torch.no_grad()
    def plan_end(self, st) -> List[BatchPlan]:
        W, P = self.world, st["P"]
        if st["counts_event"] is not None:
            st["counts_event"].synchronize()                                # the window's host wait
        send_h, recv_h = st["counts_host"][0].tolist(), st["counts_host"][1].tolist()      # [W][P]
        ss_all = [[send_h[p][b] for p in range(W)] for b in range(P)]
        rs_all = [[recv_h[p][b] for p in range(W)] for b in range(P)]
        buck = self.ops.bucketize_results(st["token"], ss_all)
        plans: List[BatchPlan] = []
        if W > 1:
            # ONE all-to-all-v for the ids of the whole window: the send buffer is peer-major, batch-minor
            pieces = [torch.split(buck[b][0], ss_all[b]) for b in range(P)]
            send_buf = torch.cat([pieces[b][p] for p in range(W) for b in range(P)])
            in_splits = [sum(ss_all[b][p] for b in range(P)) for p in range(W)]
            out_splits = [sum(rs_all[b][p] for b in range(P)) for p in range(W)]
            got_all = torch.empty(sum(out_splits), dtype=send_buf.dtype, device=send_buf.device)
            _a2a(got_all, send_buf, out_splits, in_splits, self.group)
            segs = torch.split(got_all, [rs_all[b][p] for p in range(W) for b in range(P)])
        for b in range(P):
            rows, perm = buck[b][0], buck[b][1]
            if W > 1:
                got = torch.cat([segs[p * P + b] for p in range(W)])
            else:
                got = rows
            plans.append(BatchPlan(rows.numel(), perm, ss_all[b], rs_all[b], got,
                                   keys=buck[b][3] if len(buck[b]) > 3 else None))
        all_rows = plans[0].recv_rows if P == 1 else torch.cat([p.recv_rows for p in plans])
        slots = self.ops.owner_prepare(all_rows)
        for p, s in zip(plans, torch.split(slots, [p.recv_rows.numel() for p in plans])):
            p.slots = s
        return plans

    def fetch_rows(self, plan: BatchPlan) -> torch.Tensor:
        """owner gathers -> all-to-all-v -> fp32[n, D] in my bucket order."""
        mine = self.ops.owner_gather(plan.slots)
        if self.world == 1:
            return mine
        got = torch.empty(plan.n, self.ops.dim, dtype=mine.dtype, device=mine.device)
        _a2a(got, mine, plan.send_splits, plan.recv_splits, self.group)
        return got

    def return_grads(self, plan: BatchPlan, grad_rows: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return grad_rows
        got = torch.empty(plan.recv_rows.numel(), self.ops.dim, dtype=grad_rows.dtype, device=grad_rows.device)
        _a2a(got, grad_rows.contiguous(), plan.recv_splits, plan.send_splits, self.group)
        return got


class _RowwiseFn(torch.autograd.Function):
    """pooled = pool(fetch_rows(plan)); backward ships per-lookup grads to the owners, which apply SGD."""

    @staticmethod
    def forward(ctx, anchor, ex: RowwiseExchange, plan: BatchPlan, offsets, psw, mode, include_last, hook, lr_box):
        rows = ex.fetch_rows(plan)
        out = ex.ops.pool(rows, plan.perm, offsets, psw, mode, include_last, hook)
        ctx.ex, ctx.plan, ctx.lr_box = ex, plan, lr_box
        ctx.args = (offsets, psw, mode, include_last, hook)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ex, plan = ctx.ex, ctx.plan
        offsets, psw, mode, include_last, hook = ctx.args
        with torch.no_grad():
            g = ex.ops.grad_rows(grad_out, plan.perm, offsets, psw, mode, include_last, hook, plan.n, plan.keys)
            g_own = ex.return_grads(plan, g)
            lr = ctx.lr_box[0]
            if lr is None:
                raise RuntimeError("row-wise sharded embedding needs set_fused_sgd(lr): the update is applied "
                                   "by the owner inside backward")
            ex.ops.owner_update(plan.slots, g_own, lr)
        return (None,) * 9


class RowwiseShardedEmbeddingBag(nn.Module):
    """Row-wise sharded cached EmbeddingBag: local batch in ([values, offsets] of B_loc samples),
    pooled [B_loc*F, D] (or [B_loc, F, D] with hook_features) out."""

    def __init__(self, num_embeddings: int, embedding_dim: int, mode: str = "sum", include_last_offset: bool = True,
                 cache_ratio: float = 0.01, ids_freq_mapping=None, warmup_ratio: float = 0.7,
                 evict_strategy: EvictionStrategy = EvictionStrategy.DATASET, group=None,
                 _weight_shard: Optional[torch.Tensor] = None, init_seed: int = 1024,
                 cuda_row_num: Optional[int] = None):
        super().__init__()
        _lib.require_gpu()
        self.group = group if group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.mode, self.include_last_offset = mode, include_last_offset
        dev = torch.device("cuda", torch.cuda.current_device())
        W, r = self.world, self.rank
        n_local = (num_embeddings - r + W - 1) // W
        c_total = int(num_embeddings * cache_ratio) if cuda_row_num is None else int(cuda_row_num)
        c_local = max(1, min(n_local, c_total // W))
        # global id -> frequency rank (replicated); the owner split is taken on the RANK so every shard
        # gets an equal share of the hot rows
        self.idx_map = None
        if ids_freq_mapping is not None and evict_strategy == EvictionStrategy.DATASET:
            freq = torch.as_tensor(ids_freq_mapping).to(device=dev, dtype=torch.int64).view(-1)
            order = torch.argsort(freq, descending=True, stable=True)
            inv = torch.empty(num_embeddings, device=dev, dtype=torch.int32)
            inv[order] = torch.arange(num_embeddings, device=dev, dtype=torch.int32)
            self.idx_map = inv
            del freq, order
        if _weight_shard is None:
            table = HostTable.allocate(n_local, embedding_dim)
            table.fill_uniform_(-1.0 / num_embeddings, 1.0 / num_embeddings, init_seed + 7919 * r)
        else:
            assert tuple(_weight_shard.shape) == (n_local, embedding_dim)
            table = HostTable.wrap(_weight_shard.detach().to("cpu", torch.float32).contiguous())
        self.cache_weight_mgr = CachedParamMgr(table, c_local, evict_strategy=evict_strategy, device=dev)
        local_freq = None
        if ids_freq_mapping is not None and evict_strategy == EvictionStrategy.LFU:
            local_freq = torch.as_tensor(ids_freq_mapping).view(-1)[r::W]
        self.cache_weight_mgr.reorder(local_freq, warmup_ratio)
        self.ops = HipShardOps(self.cache_weight_mgr, self.idx_map, W, r, num_global_rows=num_embeddings)
        self.exchange = RowwiseExchange(self.ops, self.group)
        self._lr = [None]
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)

    @property
    def weight(self):
        return self.cache_weight_mgr.weight

    def set_fused_sgd(self, lr: Optional[float]):
        self._lr[0] = lr

    def plan_window(self, ids_list: Sequence[torch.Tensor]) -> List[BatchPlan]:
        return self.exchange.plan_window(ids_list)

    def forward(self, input, offsets: Optional[torch.Tensor] = None, per_sample_weights=None,
                shape_hook: Optional[Callable] = None, *, hook_features: int = 0):
        """`input` is either a BatchPlan from plan_window (prefetch mode) or the local id tensor."""
        plan = input if isinstance(input, BatchPlan) else self.exchange.plan_window([input])[0]
        out = _RowwiseFn.apply(self._anchor, self.exchange, plan, offsets, per_sample_weights, self.mode,
                               self.include_last_offset, int(hook_features), self._lr)
        return shape_hook(out) if shape_hook is not None else out

    @torch.no_grad()
    def forward_backward(self, plan: BatchPlan, offsets: torch.Tensor, grad_out: torch.Tensor,
                         per_sample_weights=None, *, hook_features: int = 0) -> torch.Tensor:
        """One training step of the operator without the autograd engine (what `forward` + `.backward(grad_out)`
        run, as straight-line launches): rows in, pooled output, folded gradient rows back, owner-side SGD."""
        ex, lr = self.exchange, self._lr[0]
        if lr is None:
            raise RuntimeError("row-wise sharded embedding needs set_fused_sgd(lr)")
        rows = ex.fetch_rows(plan)
        out = ex.ops.pool(rows, plan.perm, offsets, per_sample_weights, self.mode, self.include_last_offset,
                          int(hook_features))
        g = ex.ops.grad_rows(grad_out, plan.perm, offsets, per_sample_weights, self.mode,
                             self.include_last_offset, int(hook_features), plan.n, plan.keys)
        ex.ops.owner_update(plan.slots, ex.return_grads(plan, g), lr)
        return out

    def flush(self):
        self.cache_weight_mgr.flush()


class ShardedWindowPipeline:
    """Prefetch window for the row-wise sharded module: the window plan (dedupe, owner bucketing, count/id
    all-to-alls, owner-side cache op) of window k+1 is built on a side stream while window k trains.
    The host syncs inside plan_window then only wait for that side stream, so the CPU can keep enqueueing
    training steps; rows of window k stay protected while window k+1's victims are chosen (protect_depth 1).
    Call order per window (identical on every rank, which keeps the collectives matched):
        submit(ids of window k+1)  ->  collect()  ->  train on the returned plans of window k ..."""

    def __init__(self, embed: "RowwiseShardedEmbeddingBag", overlap: bool = True,
                 transport: Optional[str] = None):
        # transport (overlap only): how the owner-side cache op moves rows while training runs beside it; None keeps
        # the manager's setting (zero-copy).  "worker" measured SLOWER here at W = 1 (1.29 vs 1.59 G lookups/s): the
        # plan stream parks in the worker transport's wait, and this pipeline queues the next window's id exchange
        # behind it
        self.embed = embed
        self.overlap = overlap
        self._pending = []
        dev = embed.cache_weight_mgr.device
        # two side streams: dedupe + owner bucketing of window k+1 must not queue behind the (long) cache op
        # of window k, so the cache-state-free phases get their own stream
        self._side = torch.cuda.Stream(device=dev) if overlap else None       # phases 1-2
        self._side2 = torch.cuda.Stream(device=dev) if overlap else None      # phase 3: id exchange + cache op
        if overlap:
            embed.cache_weight_mgr.set_protect_depth(1)
            embed.cache_weight_mgr.strict = False
            if transport:
                embed.cache_weight_mgr.set_transport(transport)

    def submit(self, ids_list: Sequence[torch.Tensor], wait_for_current: bool = True) -> None:
        """Phase 1 of the next window's plan (dedupe kernels + async count readback) on the side stream."""
        if not self.overlap:
            # reference semantics: the cache op of a window must not run before the previous window finished
            # training (its rows are not protected), so planning is deferred to collect()
            self._pending.append({"ids": list(ids_list)})
            return
        dev = self.embed.cache_weight_mgr.device
        if wait_for_current:          # ids produced on the current stream just now
            self._side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._side):
            st = self.embed.exchange.plan_begin(ids_list)
        self._pending.append(st)

    def pump(self) -> None:
        """Finish the oldest unfinished plan (call it between training steps: by then the bucket sizes have
        landed in pinned memory and the host does not wait).  Every rank must call submit/pump/collect in
        the same order -- the phases contain collectives."""
        if not self.overlap:
            return
        for st in self._pending:
            if st.get("phase") == 2:
                self._side2.wait_stream(self._side)
                for t in self.embed.ops.token_tensors(st["token"]):
                    if t.is_cuda:
                        t.record_stream(self._side2)
                with torch.cuda.stream(self._side2):
                    plans = self.embed.exchange.plan_end(st)
                    ev = torch.cuda.Event()
                    ev.record(self._side2)
                st.clear()
                st.update(phase=3, plans=plans, event=ev)
                return

    def collect(self) -> List[BatchPlan]:
        st = self._pending.pop(0)
        if not self.overlap:
            return self.embed.plan_window(st["ids"])
        while st.get("phase") != 3:           # phases not pumped yet: finish them now
            self._pending.insert(0, st)
            self.pump()
            st = self._pending.pop(0)
        plans, ev = st["plans"], st["event"]
        # strict=False: an owner-side cache op that overflowed handed back slots of -1; raise like the reference as
        # soon as its record has arrived (non-blocking)
        self.embed.cache_weight_mgr.raise_on_failed_calls()
        cur = torch.cuda.current_stream(self.embed.cache_weight_mgr.device)
        cur.wait_event(ev)
        for p in plans:
            for t in (p.perm, p.recv_rows, p.slots, p.keys.keys if isinstance(p.keys, SrcKeys) else p.keys):
                if t is not None and t.is_cuda:
                    t.record_stream(cur)
        return plans


class GraphedShardedWindow:
    """Row-wise sharded prefetch window with FIXED-CAPACITY exchanges, so that the P training steps of a window --
    owner gather, padded equal-split row all-to-all, pooling, the caller's dense part, gradient fold, padded gradient
    all-to-all, owner-side SGD -- replay as ONE hipGraph, like pipeline.GraphedWindow does for the unsharded module.

    Every (batch, owner) bucket of unique rows gets `capacity` places (ce_dedupe_bucket_rows_padded: unused places
    hold -1, which the cache op, the gather and the row update all skip), so no tensor of a step has a data-dependent
    size, the collectives are plain equal-split all_to_all_single calls, and nothing in a window needs the host:
    the one read-back left is a single overflow flag per window, checked when the window is about to train (it was
    written a whole window earlier).  A bucket that does not fit (the flag) sends that window through the
    variable-size path (RowwiseExchange.plan_window + forward_backward).

    Local bypass: the receive buffer lies right behind the cache in ONE allocation (CachedParamMgr.reserve_tail) and
    every lookup's index is either the cache slot of its row -- when this rank owns the row -- or a row of that buffer
    (ce_exchange_local_index), so pooling and the fused gradient fold + SGD run ONCE over "cache + received rows":
    rows a rank owns itself are read and updated in place and never pass the exchange buffers (1 / W of the rows; all
    of them at W = 1, where the step is the unsharded module's).  For the rest the fold leaves -lr * (sum of
    gradients) in the zeroed buffer, which travels back and is added to the owners' rows.

    The planning of window k+1 runs on two side streams while window k trains: dedupe + id exchange on one, the
    owner-side cache op (which parks its stream when the worker transport moves the rows) + the keys on the other, so
    the next window's id exchange never queues behind a parked stream.

    dense_fn(pooled [B, F, D] or [num_bags, D], i) -> gradient of the same shape for batch i of the window: the part
    of the model between the embedding's forward and backward (it is captured with the step; every tensor it reads
    or writes must be static)."""

    def __init__(self, embed: "RowwiseShardedEmbeddingBag", prefetch_num: int, ids_per_batch: int, offsets: torch.Tensor,
                 dense_fn, capacity: int, hook_features: int = 0, overlap: bool = True,
                 use_graph: Optional[bool] = None,
                 transport: Optional[str] = None, warmup_ids: Optional[Sequence[torch.Tensor]] = None,
                 split: Optional[bool] = None, split_caps: Optional[Sequence[int]] = None,
                 arrangement: Optional[str] = None, arrangement_trial: Optional[dict] = None,
                 static_out_candidates: int = 0, before_capture=None):
        # static_out_candidates > 0 (one-id-per-bag layouts; needs warmup_ids): the pooled output of every step is ONE
        # static buffer, the fastest of that many candidate allocations for this window's own forward
        # (functional.pick_fast_buffer(work=...): the same launch takes 37 or 45 us depending on how its output is
        # mapped); `out_lottery` reports what was measured.  before_capture(window): called once, after the warm-up
        # window has trained and before the steps are captured -- e.g. to choose the tensor dense_fn returns the same
        # way (enqueue_update_lr0 is the work to time on a gradient candidate).
        # arrangement (overlap=True): where the NEXT window's plan runs -- "overlap": on the two side streams beside this
        # window's steps; "interleaved": on the training stream, the owner-side cache op in two halves around this
        # window's steps (ce_cache_prepare_ids_begin_padded / _finish), so that no kernel of the plan runs beside a bag
        # kernel and only the PCIe admission overlaps -- what pipeline.GraphedWindow's one-stream arrangement is to the
        # unsharded module; "auto": pipeline.ArrangementTrial measures both while training (W > 1: the verdict is
        # COLLECTIVE -- the ranks' block times, MAX over the ranks, reduced in the same window on every rank).
        # None: "auto" where the steps are hipGraph replays (W = 1; W > 1 with use_graph / CE_SHARDED_GRAPH=1),
        # "overlap" where they are launched one by one: with EAGER steps the two halves of a cache op on the training
        # stream cost every later dispatch of that stream ~40 us on the one box this could be measured on
        # (profiles/r06_dlrm_interleaved_dispatch.md), and a trial that starts with such a block measures both
        # arrangements slow.
        # split (W > 1; default on, CE_SHARDED_SPLIT=0 switches it off): the EARLY / LATE split of both row exchanges of
        # a step (class docstring).  split_caps = (cap_early, cap_late, cap_deferred, cap_urgent) rows per peer and
        # step; None: measured on the warm-up window (mean + 4.5 sigma of every class, agreed over the ranks).
        from .functional import is_identity_layout, presort_len
        self.embed, self.ex, self.ops = embed, embed.exchange, embed.ops
        self.mgr = embed.cache_weight_mgr
        self.P, self.n, self.cap = int(prefetch_num), int(ids_per_batch), int(capacity)
        self.W = embed.world
        self.offsets = offsets.contiguous()
        self.incl = bool(embed.include_last_offset)
        self.hook = int(hook_features)
        self.dense_fn = dense_fn
        self.overlap = overlap
        dev = self.mgr.device
        W, P, cap, n = self.W, self.P, self.cap, self.n
        self.num_bags = self.offsets.numel() - 1 if self.incl else self.offsets.numel()
        self._identity = self.incl and self.num_bags == n and is_identity_layout(self.offsets, True)
        klen = presort_len(n)
        i64 = dict(dtype=torch.int64, device=dev)
        self._req = [torch.full((P, W, cap), -1, **i64) for _ in range(2)]          # rows I ask owner w for
        self._serve = [torch.full((W, P, cap), -1, **i64) for _ in range(2)]        # rows peer w asks me for
        self._slots = [torch.full((P, W * cap), -1, **i64) for _ in range(2)]       # their cache slots, per batch
        self._pos = [torch.full((P, n), -1, **i64) for _ in range(2)]
        self._keys = [torch.full((P, klen), -1, **i64) for _ in range(2)]
        self._counts = [torch.zeros(P, W, **i64) for _ in range(2)]
        self._ovf = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(2)]
        self._ovf_host = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
        self._ovf_event = [None, None]               # behind the flag's copy to the host
        self._planned = [False, False]               # a plan has been submitted into the buffer and not trained yet
        self._events = [None, None]
        self._ids = [None, None]
        self._side = torch.cuda.Stream(device=dev) if overlap else None        # dedupe + id exchange
        self._side2 = torch.cuda.Stream(device=dev) if overlap else None       # owner cache op + keys
        if overlap:
            self.mgr.set_protect_depth(1)
            self.mgr.strict = False
            if transport:
                self.mgr.set_transport(transport)
        from .pipeline import ARRANGEMENTS, ArrangementTrial, DEFAULT_ARRANGEMENT
        if arrangement is None and overlap:
            graphed = (self.W == 1 and use_graph is not False) or use_graph is True or \
                (use_graph is None and os.environ.get("CE_SHARDED_GRAPH", "0") == "1")
            arrangement = DEFAULT_ARRANGEMENT if graphed else "overlap"
        if arrangement is not None:
            if arrangement not in ("auto",) + ARRANGEMENTS:
                raise ValueError(f"arrangement={arrangement!r}: 'auto', 'overlap' or 'interleaved'")
            if not overlap:
                raise ValueError("arrangement= needs overlap=True (the next window planned while this one trains)")
        # W > 1: the trial's verdict is COLLECTIVE -- every rank's block times, MAX over the ranks (the step is as slow as
        # its slowest rank), reduced at the same window on every rank, so that all of them switch in the same window
        self.trial = ArrangementTrial(P, reduce_fn=self._reduce_max if W > 1 else None, **(arrangement_trial or {})) \
            if arrangement == "auto" else None
        self._mode = self.trial.mode if self.trial is not None else (arrangement or "sequential")
        self._begun: Optional[int] = None            # interleaved: buffer whose owner-side cache op is begun, not finished
        self._begun_slots = None
        # scratch of the dedupe passes: one stamp / place array per batch of the window, so that the P batches are
        # deduped by ONE launch per pass (P * N * 4 bytes: 5.7 GB for the Criteo-1TB table at P = 8 -- 0.7 GB per batch
        # of the window, 2 % of a GPU's HBM; CE_DEDUPE_PER_BATCH=1 dedupes batch by batch with one array instead, at
        # 3 launches per batch)
        N = embed.num_embeddings
        self._window_dedupe = P > 1 and not int(os.environ.get("CE_DEDUPE_PER_BATCH", "0"))
        if self._window_dedupe:
            # ONE int32 per (batch, row): the claim pass leaves a row's place in its bucket in the stamp entry it has
            # just checked (as -1 - place: never a lookup index), so there is no second array
            self._stamp = torch.empty(P * N, dtype=torch.int32, device=dev)
            self._slot_of_row = None
            self._ids_win = torch.empty(P, n, **i64)
        else:
            if self.ops._stamp is None:
                self.ops._stamp = torch.empty(N, dtype=torch.int32, device=dev)
            self._stamp, self._slot_of_row = self.ops._stamp, None
        self._ws = torch.empty(max(P * (W + 1) * n, 1 << 16), dtype=torch.int32, device=dev)
        if embed.mode != "sum":
            raise NotImplementedError("GraphedShardedWindow: mode='sum' only (the fused fold + update)")
        self.rank = self.ops.rank
        self._C = self.mgr.cuda_row_num
        if split is None:
            split = os.environ.get("CE_SHARDED_SPLIT", "1") != "0"
        self._split = bool(split) and W > 1
        self._force_late = self._split and os.environ.get("CE_SPLIT_FORCE", "") == "late"
        self._graphs = None
        self._out_static = None
        self.out_lottery = None
        self.fallback_windows = 0
        self.split_stats = None
        if self._split:
            self._setup_split(split_caps, warmup_ids, klen, i64)
        else:
            self._tail = self.mgr.reserve_tail(W * cap)                             # rows received, behind the cache
            self._table = self.mgr.cache_with_tail[:self._C + W * cap]
        self._idx = [torch.full((P, n), -1, **i64) for _ in range(2)]               # lookup -> row of _table
        self._slots_remote = [torch.full((P, W * cap), -1, **i64) for _ in range(2)] if W > 1 else None
        self._recv = torch.empty(W * cap, embed.embedding_dim, dtype=torch.float32, device=dev) \
            if (W > 1 and not self._split) else None
        if warmup_ids is not None:
            self._plan(list(warmup_ids), 0)
            self._plan_owner(0)
            twins = (self._slots, self._pos, self._keys, self._idx) + ((self._slots_remote,) if W > 1 else ())
            if self._split:
                twins += (self._idx_b, self._keys_b, self._lists_f, self._lists_b)
            for t in twins:
                t[1].copy_(t[0])
            torch.cuda.synchronize(dev)
            if int(self._ovf[0].item()) != 0:
                raise ValueError(f"capacity {cap} is smaller than a bucket of the warm-up window "
                                 f"(largest: {int(self._counts[0].max().item())} rows)")
            if self._split:                          # eager once (lazy initialisation must not happen in a capture)
                self._window_split(0)
            else:
                for i in range(P):
                    self._step(0, i)
            torch.cuda.synchronize(dev)
            # use_graph None: at W = 1 (no collective in the steps, the host is the bottleneck: 1.73 -> 2.05 G) yes; at
            # W > 1 only when asked for (CE_SHARDED_GRAPH=1): a step then holds two all-to-alls of tens of microseconds
            # each, which hide the ~0.1 ms the host needs to launch it, and a capture with RCCL collectives inside has
            # never run on hardware here -- the launched-one-by-one form is the one the world-2/3 tests exercise
            if use_graph is None:
                use_graph = W == 1 or os.environ.get("CE_SHARDED_GRAPH", "0") == "1"
            if static_out_candidates > 0 and self._identity and FORWARD_FROM_KEYS:
                from .functional import pick_fast_buffer
                k0 = self._keys0()
                shape = (self.num_bags // self.hook, self.hook, embed.embedding_dim) if self.hook \
                    else (self.num_bags, embed.embedding_dim)
                self._out_static, self.out_lottery = pick_fast_buffer(
                    shape, dev, max(1, self.hook), candidates=int(static_out_candidates), use="write",
                    work=lambda b: self.ops.pool_from_keys(self._table, k0, self.n, out=b))
            if before_capture is not None:
                before_capture(self)
            if use_graph and not (W > 1 and dist.get_backend(self.ex.group) == "gloo"):     # gloo stages through the host
                self._capture()

    def _keys0(self) -> "SrcKeys":
        """source-row keys of the warm-up window's first batch (valid once that window has been planned)"""
        return SrcKeys(self._keys[0][0], self.num_bags, self.incl, self.hook, None, self._identity)

    def enqueue_update_lr0(self, grad_buf: torch.Tensor) -> None:
        """The table update of the warm-up window's first batch with `grad_buf` as the upstream gradient and lr = 0
        (nothing changes): the work to time on a candidate gradient buffer (functional.pick_fast_buffer(work=...))."""
        keys = SrcKeys(self._keys_b[0][0], self.num_bags, self.incl, self.hook, None, self._identity) if self._split \
            else self._keys0()
        self.ops.update_table(self._table, grad_buf, keys, self.n, 0.0)

    # ---- planning (per window, no host wait)
    @torch.no_grad()
    def _plan(self, ids_list: Sequence[torch.Tensor], buf: int) -> None:
        W, P, cap, n = self.W, self.P, self.cap, self.n
        assert len(ids_list) == P
        sp = stream_ptr()
        self._ovf[buf].zero_()
        if self._window_dedupe:
            # the window's ids as one [P, n] block: as they are when the batches are rows of one tensor, copied otherwise
            first = ids_list[0]
            win = None
            if first.dtype == torch.int64 and all(
                    t.dtype == torch.int64 and t.is_contiguous() and t.numel() == n and
                    t.data_ptr() == first.data_ptr() + 8 * n * b for b, t in enumerate(ids_list)):
                win = first
            else:
                for b, t in enumerate(ids_list):
                    assert t.numel() == n
                    self._ids_win[b].copy_(t.reshape(-1))
                win = self._ids_win
            check(lib.ce_dedupe_bucket_rows_padded_window(win.data_ptr(), n, P, ptr(self.ops.idx_map),
                                                          self.embed.num_embeddings, W, cap, ptr(self._stamp),
                                                          ptr(self._slot_of_row), ptr(self._ws),
                                                          ptr(self._req[buf]), ptr(self._pos[buf]),
                                                          ptr(self._counts[buf]), ptr(self._ovf[buf]), sp))
        else:
            for b, ids in enumerate(ids_list):
                ids = ids.reshape(-1).long().contiguous()
                assert ids.numel() == n
                check(lib.ce_dedupe_bucket_rows_padded(ptr(ids), n, ptr(self.ops.idx_map), self.embed.num_embeddings, W,
                                                       cap, ptr(self._stamp), ptr(self._slot_of_row), ptr(self._ws),
                                                       self._req[buf][b].data_ptr(), self._pos[buf][b].data_ptr(),
                                                       self._counts[buf][b].data_ptr(), ptr(self._ovf[buf]), sp))
        if not self._split:
            self._agree_on_overflow(buf)
        if W > 1:
            # peer-major for the exchange; local rows fit 32 bits (N / W < 2^31): half the bytes of the id exchange
            req_wpc = self._req[buf].permute(1, 0, 2).to(torch.int32, memory_format=torch.contiguous_format)
            got = torch.empty_like(req_wpc)
            _a2a(got, req_wpc, None, None, self.ex.group)
            self._serve[buf].copy_(got)
        else:
            self._serve[buf].copy_(self._req[buf].permute(1, 0, 2))
        if self._split and getattr(self, "_caps", None) is not None:
            self._plan_classify(buf)

    @torch.no_grad()
    def _plan_classify(self, buf: int) -> None:
        """The split's share of the plan that needs only what every peer asked for (not the slots): classification,
        the flags' all-to-all, the places of every row in the four messages of a step, the ranks' agreement on the
        overflow flag.  HERE, right behind the id exchange and BEFORE the owner-side cache op: a process
        group issues its collectives on one internal stream in host order, and submit(k+1) comes before run(k) -- with
        these two collectives behind the cache op, window k's row exchanges would queue behind window k+1's PCIe
        admission."""
        W, P, cap, r = self.W, self.P, self.cap, self.rank
        sp = stream_ptr()
        self._classify(buf)
        got = torch.empty_like(self._flags_o[buf])
        _a2a(got, self._flags_o[buf], None, None, self.ex.group)                     # [owner, P, cap]
        self._flags_r[buf].copy_(got.permute(1, 0, 2))
        serve_pwc = self._serve[buf].permute(1, 0, 2).contiguous()
        flags_o_pwc = self._flags_o[buf].permute(1, 0, 2).contiguous()
        check(lib.ce_split_places(ptr(serve_pwc), ptr(flags_o_pwc), P, W, cap, r, ptr(self._caps), ptr(self._pf_srv[buf]),
                                  ptr(self._pb_srv[buf]), None, ptr(self._ovf[buf]), sp))
        check(lib.ce_split_places(ptr(self._req[buf]), ptr(self._flags_r[buf]), P, W, cap, r, ptr(self._caps),
                                  ptr(self._pf_req[buf]), ptr(self._pb_req[buf]), None, ptr(self._ovf[buf]), sp))
        self._agree_on_overflow(buf)

    def _reduce_max(self, values):
        """MAX over the ranks of a short list of floats (the arrangement trial's block times)"""
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if dist.get_backend(self.ex.group) == "gloo":
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.ex.group)
        else:
            d = t.to(self.mgr.device)
            dist.all_reduce(d, op=dist.ReduceOp.MAX, group=self.ex.group)
            t = d.cpu()
        return [float(v) for v in t]

    def _agree_on_overflow(self, buf: int) -> None:
        """the ranks must agree on which path a window takes (its collectives differ): MAX over the ranks' flags, then
        the one read-back of the window, asynchronously"""
        if self.W > 1:
            if dist.get_backend(self.ex.group) == "gloo":
                f = self._ovf[buf].cpu()
                dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self.ex.group)
                self._ovf[buf].copy_(f)
            else:
                dist.all_reduce(self._ovf[buf], op=dist.ReduceOp.MAX, group=self.ex.group)
        self._ovf_host[buf].copy_(self._ovf[buf], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ovf_event[buf] = ev

    @torch.no_grad()
    def _plan_owner(self, buf: int) -> None:
        slots = self.mgr.prepare_ids(self._serve[buf].view(-1), padded=True)         # -1 = padding: slot -1
        self._plan_owner_rest(buf, slots)

    @torch.no_grad()
    def _plan_owner_begin(self, buf: int) -> None:
        """interleaved: the first half of the owner-side cache op (everything that needs nothing from the host table)"""
        self._begun_slots = torch.empty(self._serve[buf].numel(), dtype=torch.int64, device=self.mgr.device)
        self.mgr.prepare_ids_begin_padded(self._serve[buf].view(-1), self._begun_slots)
        self._begun = buf

    @torch.no_grad()
    def _finish_begun(self) -> None:
        if self._begun is None:
            return
        buf, slots = self._begun, self._begun_slots
        self._begun, self._begun_slots = None, None
        self.mgr.prepare_ids_finish()
        self._plan_owner_rest(buf, slots)

    @property
    def arrangement(self) -> str:
        return self._mode

    def set_arrangement(self, mode: str) -> None:
        """'overlap' / 'interleaved' for the windows submitted from now on (between windows)"""
        assert mode in ("overlap", "interleaved") and self.overlap
        self._finish_begun()
        self._mode = mode

    def drain(self) -> None:
        self._finish_begun()

    @torch.no_grad()
    def _plan_owner_rest(self, buf: int, slots: torch.Tensor) -> None:
        W, P, cap = self.W, self.P, self.cap
        self._slots[buf].view(P, W, cap).copy_(slots.view(W, P, cap).permute(1, 0, 2))
        r = self.rank
        from .functional import presort_window
        if self._split:
            self._plan_split(buf)
            return
        check(lib.ce_exchange_local_index(ptr(self._pos[buf]), self.n, P, ptr(self._slots[buf]), W * cap, r * cap,
                                          (r + 1) * cap, self._C, ptr(self._idx[buf]), stream_ptr()))
        if W > 1:                      # what the owner side still serves: everything but this rank's own requests
            self._slots_remote[buf].copy_(self._slots[buf])
            self._slots_remote[buf].view(P, W, cap)[:, r].fill_(-1)
        presort_window(self._idx[buf], self._C + W * cap, keys_out=self._keys[buf], offsets=self.offsets,
                       include_last_offset=self.incl, hook_features=self.hook, identity_bags=self._identity)

    # ---- the early / late split (W > 1)
    def _setup_split(self, split_caps, warmup_ids, klen: int, i64: dict) -> None:
        W, P, cap, n, dev = self.W, self.P, self.cap, self.n, self.mgr.device
        D = self.embed.embedding_dim
        n_local = int(self.mgr.num_embeddings)
        self._mask = torch.zeros(n_local, dtype=torch.int64, device=dev)            # ce_split_classify's scratch
        self._flags_o = [torch.zeros(W, P, cap, dtype=torch.uint8, device=dev) for _ in range(2)]    # what I serve
        self._flags_r = [torch.zeros(P, W, cap, dtype=torch.uint8, device=dev) for _ in range(2)]    # what I request
        if self._force_late:
            split_caps = (128, cap, 128, cap)
        if split_caps is None and warmup_ids is not None:
            split_caps = self._measure_split_caps(list(warmup_ids))
        if split_caps is None:         # nothing to measure on: every class may take a whole bucket
            split_caps = (cap, cap, cap, cap)
        ce_, cl, cd, cu = (int(v) for v in split_caps)
        assert min(ce_, cl, cd, cu) >= 1 and max(ce_, cl, cd, cu) <= cap
        self.split_caps = (ce_, cl, cd, cu)
        caps = torch.tensor([[ce_, cl, cd, cu]] * P, dtype=torch.int32)
        caps[P - 1, 2], caps[P - 1, 3] = 1, cap       # the last batch returns everything at once (next window unknown)
        self._caps_host = caps.tolist()
        self._caps = caps.to(dev)
        ne, nl, nd, nu = W * ce_, W * cl, W * max(cd, 1), W * cap
        self._n = (ne, nl, nd, nu)
        self._bwd_base = 2 * ne + nl
        T = 2 * ne + nl + 2 * nd + nu
        self._tail = self.mgr.reserve_tail(T)
        self._table = self.mgr.cache_with_tail[:self._C + T]
        t = self._tail
        self._E = [t[0:ne], t[ne:2 * ne]]
        self._L = t[2 * ne:2 * ne + nl]
        b0 = self._bwd_base
        # [D0 | U | D1]: a step's fold writes its deferred copy and U -- one contiguous range to zero either way
        self._Dg = [t[b0:b0 + nd], t[b0 + nd + nu:b0 + 2 * nd + nu]]
        self._U = t[b0 + nd:b0 + nd + nu]
        self._bwd_zero = [t[b0:b0 + nd + nu], t[b0 + nd:b0 + 2 * nd + nu]]
        i32 = dict(dtype=torch.int32, device=dev)
        self._pf_req = [torch.full((P, W * cap), -1, **i32) for _ in range(2)]
        self._pb_req = [torch.full((P, W * cap), -1, **i32) for _ in range(2)]
        self._pf_srv = [torch.full((P, W * cap), -1, **i32) for _ in range(2)]
        self._pb_srv = [torch.full((P, W * cap), -1, **i32) for _ in range(2)]
        # owner-side gather / update lists of every batch: [P, ne + nl (+1 dummy)] and [P, nd + nu (+1)]
        self._lists_f = [torch.full((P, ne + nl + 1), -1, **i64) for _ in range(2)]
        self._lists_b = [torch.full((P, nd + nu + 1), -1, **i64) for _ in range(2)]
        self._idx_b = [torch.full((P, n), -1, **i64) for _ in range(2)]
        self._keys_b = [torch.full((P, klen), -1, **i64) for _ in range(2)]
        self._recv_u = torch.empty(nu, D, dtype=torch.float32, device=dev)
        self._recv_d = torch.empty(nd, D, dtype=torch.float32, device=dev)
        self._comm = torch.cuda.Stream(device=dev)

    def _classify(self, buf: int) -> None:
        W, P, cap = self.W, self.P, self.cap
        check(lib.ce_split_classify(ptr(self._serve[buf]), W, P, cap, self._mask.numel(), None, 0, ptr(self._mask),
                                    ptr(self._flags_o[buf]), stream_ptr()))
        if self._force_late:           # test mode: every row late and urgent (the synchronous exchange, in split form)
            self._flags_o[buf].fill_(3)

    @torch.no_grad()
    def _measure_split_caps(self, ids_list) -> Tuple[int, int, int, int]:
        """capacities of the four classes from one window: dedupe + id exchange + classification, then every
        (batch, peer) chunk's class counts; mean + 4.5 sigma (or the largest seen), rounded to 128 rows, such that an
        early / deferred surplus always finds room behind the late / urgent rows; MAX over the ranks"""
        W, P, cap, dev = self.W, self.P, self.cap, self.mgr.device
        self._plan(ids_list, 0)
        self._classify(0)
        caps = torch.tensor([[cap] * 4] * P, dtype=torch.int32, device=dev)
        counts = torch.zeros(P, W, 4, dtype=torch.int32, device=dev)
        ovf = torch.zeros(1, dtype=torch.int32, device=dev)
        pf = torch.empty(P, W * cap, dtype=torch.int32, device=dev)
        pb = torch.empty_like(pf)
        serve_pwc = self._serve[0].permute(1, 0, 2).contiguous()        # (named: a temporary inside the call would be
        flags_pwc = self._flags_o[0].permute(1, 0, 2).contiguous()       # freed, and its block reused, before the launch)
        check(lib.ce_split_places(ptr(serve_pwc), ptr(flags_pwc), P, W, cap, -1, ptr(caps), ptr(pf), ptr(pb), ptr(counts),
                                  ptr(ovf), stream_ptr()))
        c = counts.double().cpu()                                                # [P, W, 4]: early, late, deferred, urgent
        fwd = c[1:] if P > 1 else c                  # batch 0 is all early (nothing precedes the window)
        bwd = c[:-1] if P > 1 else c                 # the last batch is all urgent
        def need(x):
            x = x.reshape(-1)
            return max(float(x.max()), float(x.mean() + 4.5 * x.std(unbiased=False)))
        r128 = lambda v: min(cap, (int(v) + 1 + 127) // 128 * 128)
        cl, cu = r128(need(fwd[..., 1])), r128(need(bwd[..., 3]))
        ce_ = max(r128(need(fwd[..., 0])), min(cap, cap - cl + 128))
        cd = max(r128(need(bwd[..., 2])), min(cap, cap - cu + 128))
        out = torch.tensor([ce_, cl, cd, cu], dtype=torch.int64)
        self.split_stats = {"early_frac": float(fwd[..., 0].sum() / max(1.0, float(fwd[..., :2].sum()))),
                            "deferred_frac": float(bwd[..., 2].sum() / max(1.0, float(bwd[..., 2:].sum()))),
                            "mean_bucket": float(c[..., :2].sum(dim=-1).mean())}
        if dist.get_backend(self.ex.group) == "gloo":
            dist.all_reduce(out, op=dist.ReduceOp.MAX, group=self.ex.group)
        else:
            o = out.to(dev)
            dist.all_reduce(o, op=dist.ReduceOp.MAX, group=self.ex.group)
            out = o.cpu()
        return tuple(int(v) for v in out)

    @torch.no_grad()
    def _plan_split(self, buf: int) -> None:
        """behind the owner-side cache op (the slots are known): the gather / update lists of every step from the
        places _plan_classify worked out, then the two lookup indices and their keys"""
        from .functional import presort_window
        W, P, cap, n, r = self.W, self.P, self.cap, self.n, self.rank
        ne, nl, nd, nu = self._n
        sp = stream_ptr()
        # owner lists: the slot of every served row at its place in the step's messages (early | late, deferred | urgent)
        slots = self._slots[buf]                                                     # [P, W * cap]
        caps = self._caps.long()
        for pl, lists, first, n_first, col in ((self._pf_srv[buf], self._lists_f[buf], ne, ne, 0),
                                               (self._pb_srv[buf], self._lists_b[buf], nd, nd, 2)):
            lists.fill_(-1)
            wfirst = (W * caps[:, col]).unsqueeze(1)                                 # rows of the first region, per batch
            p64 = pl.long()
            dest = torch.where(p64 < wfirst, p64, n_first + p64 - wfirst)
            dest = torch.where(p64 >= 0, dest, torch.full_like(dest, lists.shape[1] - 1))     # the dummy column
            lists.scatter_(1, dest, slots)
            lists[:, -1] = -1
        check(lib.ce_exchange_local_index_split(ptr(self._pos[buf]), n, P, ptr(slots), ptr(self._pf_req[buf]),
                                                ptr(self._pb_req[buf]), W * cap, r * cap, (r + 1) * cap, self._C,
                                                self._bwd_base, ptr(self._caps), W, ne, nu, nd, ptr(self._idx[buf]),
                                                ptr(self._idx_b[buf]), sp))
        rows = self._table.shape[0]
        for idx, keys in ((self._idx[buf], self._keys[buf]), (self._idx_b[buf], self._keys_b[buf])):
            presort_window(idx, rows, keys_out=keys, offsets=self.offsets, include_last_offset=self.incl,
                           hook_features=self.hook, identity_bags=self._identity)

    def _send_rows(self, buf: int, i: int, late: bool) -> None:
        """owner gather of batch i's early / late rows and their all-to-all into the peers' forward buffers"""
        ne, nl, _, _ = self._n
        ce_, cl = self._caps_host[i][0], self._caps_host[i][1]
        if late:
            sl, dst = self._lists_f[buf][i, ne:ne + self.W * cl], self._L[:self.W * cl]
        else:
            sl, dst = self._lists_f[buf][i, :self.W * ce_], self._E[i & 1][:self.W * ce_]
        rows = self.ops.owner_gather(sl)
        _a2a(dst, rows, None, None, self.ex.group)

    def _return_grads(self, buf: int, i: int, urgent: bool) -> None:
        """all-to-all of batch i's urgent / deferred row deltas back to their owners, which add them to the cache"""
        _, _, nd, _ = self._n
        cd, cu = self._caps_host[i][2], self._caps_host[i][3]
        if urgent:
            src, rcv, sl = self._U[:self.W * cu], self._recv_u[:self.W * cu], self._lists_b[buf][i, nd:nd + self.W * cu]
        else:
            src, rcv, sl = self._Dg[i & 1][:self.W * cd], self._recv_d[:self.W * cd], self._lists_b[buf][i, :self.W * cd]
        _a2a(rcv, src, None, None, self.ex.group)
        self.ops.owner_update(sl, rcv, -1.0)                                         # cache row += received delta

    def _window_split(self, buf: int) -> None:
        """The P steps of a window with both row exchanges split (W > 1).  Communication stream, per step i:
            U(i-1) urgent deltas of the step before -> L(i) late rows of this step -> D(i-1) deferred deltas -> E(i+1)
            early rows of the NEXT step
        Compute stream: wait for E(i) and L(i), pool, the caller's dense part, fold + SGD (into delta buffers the
        communication stream zeroed behind L(i)).
        Only U(i-1) and L(i) sit between two steps' kernels; D(i-1) and E(i+1) travel while step i computes."""
        ex, ops, W, P = self.ex, self.ops, self.W, self.P
        lr = self.embed._lr[0]
        if lr is None:
            raise RuntimeError("row-wise sharded embedding needs set_fused_sgd(lr)")
        dev = self.mgr.device
        cur, comm = torch.cuda.current_stream(dev), self._comm
        comm.wait_stream(cur)                  # everything before this window, its last gradients included
        with torch.cuda.stream(comm):
            self._send_rows(buf, 0, late=False)
            ev_e = torch.cuda.Event()
            ev_e.record(comm)
        ev_fold = None
        for i in range(P):
            ev_e_next = None
            with torch.cuda.stream(comm):
                if i > 0:
                    comm.wait_event(ev_fold)
                    self._return_grads(buf, i - 1, urgent=True)
                self._send_rows(buf, i, late=True)
                ev_l = torch.cuda.Event()
                ev_l.record(comm)
                # the delta buffers step i folds into, zeroed HERE -- behind the messages that last carried them (U: the
                # urgent return just above; D[i & 1]: step i - 2's deferred return, an iteration ago), beside step i's
                # pooling -- instead of between the dense part and the fold on the training stream (6 us of every step)
                self._bwd_zero[i & 1].zero_()
                ev_z = torch.cuda.Event()
                ev_z.record(comm)
                if i > 0 and self._caps_host[i - 1][2] > 0:
                    self._return_grads(buf, i - 1, urgent=False)
                if i + 1 < P:
                    self._send_rows(buf, i + 1, late=False)
                    ev_e_next = torch.cuda.Event()
                    ev_e_next.record(comm)
            cur.wait_event(ev_e)
            cur.wait_event(ev_l)
            keys = SrcKeys(self._keys[buf][i], self.num_bags, self.incl, self.hook, None, self._identity)
            if self._identity and FORWARD_FROM_KEYS and hasattr(ops, "pool_from_keys"):
                out = ops.pool_from_keys(self._table, keys, self.n, out=self._out_static)
            else:
                out = ops.pool(self._table, self._idx[buf][i], self.offsets, None, "sum", self.incl, self.hook)
            grad = self.dense_fn(out, i)
            cur.wait_event(ev_z)
            keys_b = SrcKeys(self._keys_b[buf][i], self.num_bags, self.incl, self.hook, None, self._identity)
            ops.update_table(self._table, grad, keys_b, self.n, lr)       # own rows: SGD in place; buffers: -lr * sum g
            ev_fold = torch.cuda.Event()
            ev_fold.record(cur)
            ev_e = ev_e_next
        with torch.cuda.stream(comm):
            comm.wait_event(ev_fold)
            self._return_grads(buf, P - 1, urgent=True)                   # (the last batch has no deferred rows)
        cur.wait_stream(comm)

    def submit(self, ids_list: Sequence[torch.Tensor], buf: int) -> None:
        """Plan of a window into buffer `buf` (0/1); call it before run() of the previous window so they overlap."""
        dev = self.mgr.device
        self._ids[buf] = list(ids_list)
        self._planned[buf] = True
        if not self.overlap:
            self._plan(ids_list, buf)
            self._plan_owner(buf)
            ev = torch.cuda.Event()
            ev.record()
            self._events[buf] = ev
            return
        cur = torch.cuda.current_stream(dev)
        if self._mode == "interleaved":
            # one stream: dedupe + id exchange + the first half of the owner-side cache op now, i.e. BEFORE the steps of
            # the window in training; run() enqueues the second half and the rest of the plan behind those steps
            self._finish_begun()
            for e_ in self._events:                    # (a plan still running on the side streams comes first)
                if e_ is not None:
                    cur.wait_event(e_)
            self._plan(ids_list, buf)
            self._plan_owner_begin(buf)
            self._events[buf] = None
            return
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self._plan(ids_list, buf)
        self._side2.wait_stream(self._side)
        with torch.cuda.stream(self._side2):
            self._plan_owner(buf)
            ev = torch.cuda.Event()
            ev.record(self._side2)
        for t in ids_list:
            t.record_stream(self._side)
        self._events[buf] = ev

    # ---- one training step on batch i of buffer buf (captured)
    def _step(self, buf: int, i: int) -> None:
        ex, ops, W, cap = self.ex, self.ops, self.W, self.cap
        lr = self.embed._lr[0]
        if lr is None:
            raise RuntimeError("row-wise sharded embedding needs set_fused_sgd(lr)")
        idx = self._idx[buf][i]
        if W > 1:
            rows = ops.owner_gather(self._slots_remote[buf][i])             # [W * cap, D]; padding / own chunk: zeros
            _a2a(self._tail, rows, None, None, ex.group)
        keys = SrcKeys(self._keys[buf][i], self.num_bags, self.incl, self.hook, None, self._identity)
        if self._identity and FORWARD_FROM_KEYS and hasattr(ops, "pool_from_keys"):
            out = ops.pool_from_keys(self._table, keys, self.n, out=self._out_static)      # one id per bag: from the keys too
        else:
            out = ops.pool(self._table, idx, self.offsets, None, "sum", self.incl, self.hook)
        grad = self.dense_fn(out, i)
        if W > 1:
            self._tail.zero_()
        ops.update_table(self._table, grad, keys, self.n, lr)               # own rows: SGD in place; tail: -lr * sum g
        if W > 1:
            _a2a(self._recv, self._tail, None, None, ex.group)
            ops.owner_update(self._slots_remote[buf][i], self._recv, -1.0)   # cache row += received delta

    def _capture(self) -> None:
        dev = self.mgr.device
        mode = {}
        if self.W > 1:
            # the process group's watchdog thread polls events of finished collectives: let it drain (it would
            # invalidate a capture in the default "global" error mode) and keep other threads' calls out of the capture
            torch.cuda.synchronize(dev)
            dist.barrier(group=self.ex.group)
            torch.cuda.synchronize(dev)
            import time
            time.sleep(0.5)
            mode = dict(capture_error_mode="thread_local")
        try:
            graphs = []
            for b in range(2):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **mode):
                    if self._split:
                        self._window_split(b)
                    else:
                        for i in range(self.P):
                            self._step(b, i)
                graphs.append(g)
            self._graphs = graphs
        except Exception as e:                      # e.g. a collective backend that cannot be captured
            import warnings
            warnings.warn(f"the window's steps could not be captured in a hipGraph ({e}); launching them one by one")
            torch.cuda.synchronize(dev)
            self._graphs = None

    def run(self, buf: int) -> None:
        """Train the P batches of the window in buffer `buf` (waits for its plan)."""
        dev = self.mgr.device
        self._planned[buf] = False
        if self._begun == buf:             # (no window trained in between: the two halves back to back)
            self._finish_begun()
        ev = self._events[buf]
        if ev is not None:
            ev.synchronize()               # the plan was enqueued a whole window ago: the flag is there
            torch.cuda.current_stream(dev).wait_event(ev)
            self._events[buf] = None
        elif self._ovf_event[buf] is not None:
            # the plan ran on THIS stream (submitted under 'interleaved' -- whatever the arrangement is by now: it can
            # change between submit and run): its overflow flag was copied to the host behind it
            self._ovf_event[buf].synchronize()
        if not self.mgr.strict:
            self.mgr.raise_on_failed_calls()
        if int(self._ovf_host[buf][0]) != 0:
            # a bucket did not fit `capacity`: this window goes through the variable-size exchange (its rows are
            # resident already -- the padded plan admitted every row that did fit; plan_window admits the rest)
            self.fallback_windows += 1
            self._finish_begun()             # (interleaved: the next window's cache op is begun: finish it first)
            if self._side2 is not None:      # the next window's plan is in flight: the cache manager is not re-entrant
                self._side2.synchronize()
            plans = self.embed.plan_window(self._ids[buf])
            for i, p_ in enumerate(plans):
                rows = self.ex.fetch_rows(p_)
                out = self.ops.pool(rows, p_.perm, self.offsets, None, self.embed.mode, self.incl, self.hook)
                grad_like = self.dense_fn(out, i)
                g = self.ops.grad_rows(grad_like, p_.perm, self.offsets, None, self.embed.mode, self.incl, self.hook,
                                       p_.n, p_.keys)
                self.ops.owner_update(p_.slots, self.ex.return_grads(p_, g), self.embed._lr[0])
            nbuf = 1 - buf
            if self._planned[nbuf]:
                # The NEXT window's plan ran before this fallback's cache op, so its rows are no longer "the call
                # before" when the window after it is planned: protect_depth 1 would let that plan evict rows the next
                # window has not trained on yet.  Its owner-side cache op is issued again (every row it names is
                # stamped again, and re-admitted if the fallback evicted it) and the indices that depend on the slots
                # are rebuilt.  Rare by construction (a bucket beyond mean + 4.5 sigma), so the cost does not matter.
                if self._events[nbuf] is not None:
                    torch.cuda.current_stream(dev).wait_event(self._events[nbuf])
                slots = self.mgr.prepare_ids(self._serve[nbuf].view(-1), padded=True)
                self._plan_owner_rest(nbuf, slots)
            return
        if self._graphs is not None:
            self._graphs[buf].replay()
        elif self._split:
            self._window_split(buf)
        else:
            for i in range(self.P):
                self._step(buf, i)
        self._finish_begun()                 # interleaved: the NEXT window's second half, behind this window's steps
        if self.trial is not None:
            mode = self.trial.window_done(torch.cuda.current_stream(dev))
            if mode != self._mode:
                self.set_arrangement(mode)


class ParallelCachedEmbeddingBag(CachedEmbeddingBag):
    """Column-wise parallel cached EmbeddingBag -- the class recsys/models/dlrm.py:70-81 builds.
    Every rank holds num_embeddings x (embedding_dim / W) and receives the GLOBAL batch; forward returns
    shape_hook(local pooled) exchanged with dual_all_to_all(scatter batch, gather dim)."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2.0,
                 scale_grad_by_freq=False, sparse=False, _weight=None, mode="mean", include_last_offset=False,
                 dtype=None, device=None, cache_ratio=0.01, ids_freq_mapping=None, warmup_ratio=0.7, buffer_size=0,
                 pin_weight=False, evict_strategy: EvictionStrategy = EvictionStrategy.DATASET, group=None, **kw):
        self.group = group if group is not None else (dist.group.WORLD if dist.is_initialized() else None)
        self.rank = dist.get_rank(self.group) if self.group is not None else 0
        self.world_size = dist.get_world_size(self.group) if self.group is not None else 1
        self.full_embedding_dim = embedding_dim
        lo, hi, _ = get_partition(embedding_dim, self.rank, self.world_size)
        self.partition_start_index, self.partition_end_index = lo, hi
        if _weight is not None:
            _weight = _weight[:, lo:hi].contiguous()
        super().__init__(num_embeddings, hi - lo, padding_idx, max_norm, norm_type, scale_grad_by_freq, sparse,
                         _weight, mode, include_last_offset, dtype, device, cache_ratio, ids_freq_mapping,
                         warmup_ratio, buffer_size, pin_weight, evict_strategy, **kw)

    def forward(self, indices, offsets=None, per_sample_weights=None, shape_hook=None, scatter_dim=0, gather_dim=-1,
                *, hook_features: int = 0, presorted=None):
        out = super().forward(indices, offsets, per_sample_weights, shape_hook, hook_features=hook_features,
                              presorted=presorted)
        if self.world_size == 1:
            return out
        sizes = None
        if gather_dim in (-1, out.dim() - 1):      # column partition of every peer is known without talking
            sizes = [hi - lo for lo, hi, _ in (get_partition(self.full_embedding_dim, p, self.world_size)
                                               for p in range(self.world_size))]
        return dual_all_to_all(out, self.group, scatter_dim=scatter_dim, gather_dim=gather_dim, gather_sizes=sizes)
