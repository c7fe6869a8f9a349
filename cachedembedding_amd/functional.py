"""torch-facing wrappers of the HIP EmbeddingBag kernels (ce_bag_* in include/ce_api.h).

`embedding_bag` mirrors the call the reference makes through ColossalAI:
``F.embedding_bag(slots, cuda_cached_weight, offsets, max_norm, norm_type, scale_grad_by_freq,
mode, sparse, per_sample_weights, include_last_offset, padding_idx)`` (SURVEY.md A.7;
call sites recsys/models/dlrm.py:99-110, benchmark/benchmark_cache.py:62), with two
additions the reference does not have: `hook_features` folds sparse_embedding_shape_hook
(recsys/models/dlrm.py:26-27) into the output store, and `fused_sgd_lr` applies
SGD.step (recsys/dlrm_main.py:279) inside the backward pass.
"""
from __future__ import annotations

from typing import NamedTuple, Optional, Union

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

_MODES = {"sum": _lib.CE_MODE_SUM, "mean": _lib.CE_MODE_MEAN}


def _prep(indices: torch.Tensor, offsets: Optional[torch.Tensor], include_last_offset: bool):
    if indices.dim() == 2:
        if offsets is not None:
            raise ValueError("if input is 2D, then offsets has to be None, as input is treated is a "
                             "mini-batch of fixed length sequences")
        rows, length = indices.shape
        offsets = torch.arange(0, rows * length + 1, length, device=indices.device, dtype=torch.int64)
        indices = indices.reshape(-1)
        include_last_offset = True
    elif indices.dim() == 1:
        if offsets is None:
            raise ValueError("offsets has to be a 1D Tensor but got None")
        if offsets.dim() != 1:
            raise ValueError("offsets has to be a 1D Tensor")
    else:
        raise ValueError(f"input has to be 1D or 2D Tensor, but got Tensor of dimension {indices.dim()}")
    if indices.dtype != torch.int64:
        indices = indices.long()
    if offsets.dtype not in (torch.int32, torch.int64):
        offsets = offsets.long()
    indices = indices.contiguous()
    offsets = offsets.contiguous()
    num_bags = offsets.numel() - 1 if include_last_offset else offsets.numel()
    return indices, offsets, include_last_offset, num_bags


class SrcKeys(NamedTuple):
    """Keys of one batch from presort_window(..., offsets=...): row << 32 | grad_out row (ce_bag_presort_window_src).
    They already hold the bag layout they were built for, so embedding_bag(presorted=SrcKeys) checks that the call
    uses the same one; mode must be 'sum' without per-sample weights.
    ranges (presort_window(..., ids=...)): the [min, max] id of every 16384-lookup segment of the batch, int64
    [segments, 2] -- with them the fused-SGD backward updates rows that one lane group owns entirely (flagged in the
    keys) by plain read-modify-write instead of atomics, after checking that no id can occur in two segments."""
    keys: torch.Tensor
    num_bags: int
    include_last_offset: bool
    hook_features: int
    ranges: Optional[torch.Tensor] = None
    # the keys were built for the one-id-per-bag layout (presort_window(identity_bags=True)): the FORWARD can then run
    # from them as well (ce_bag_forward_src_keys: a cache row is loaded once per run of equal rows)
    identity: bool = False


class _BagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, indices, offsets, psw, mode, include_last, hook_features, sparse, fused, presorted,
                bwd_scale=None, masked=False, out_box=None):
        _lib.require_gpu()
        assert weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous()
        if psw is not None and ctx.needs_input_grad[3] and fused is not None and fused.lr is not None:
            # d loss / d per_sample_weights[j] = <grad_out[bag of j], weight[indices[j]]> needs the rows as the forward
            # saw them, and the fused update moves them inside backward: refused HERE, before any kernel has run (a
            # refusal inside backward would leave the caller with an exception AND an updated table)
            raise NotImplementedError("gradient w.r.t. per_sample_weights with the fused SGD update")
        num_bags = offsets.numel() - 1 if include_last else offsets.numel()
        dim = weight.shape[1]
        shape = (num_bags // hook_features, hook_features, dim) if hook_features else (num_bags, dim)
        if out_box is not None:
            # the caller's buffer (embedding_bag(out=...)): a static output for graph-captured steps, or one chosen by
            # pick_fast_buffer.  It comes in a box because it is no input of the autograd function.
            out = out_box[0]
            if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or \
                    out.device != weight.device:
                raise ValueError(f"out= must be a contiguous fp32 tensor of shape {shape} on {weight.device}")
        else:
            out = torch.empty(shape, device=weight.device, dtype=torch.float32)
        if FORWARD_FROM_KEYS and isinstance(presorted, SrcKeys) and presorted.identity and psw is None \
                and mode == _lib.CE_MODE_SUM and num_bags == indices.numel():
            # one id per bag: out[bag] = W[slot], and the window's keys hold (slot, output row) grouped by slot
            check(lib.ce_bag_forward_src_keys(ptr(weight), weight.shape[0], dim, indices.numel(), ptr(presorted.keys),
                                              ptr(out), stream_ptr()))
        else:
            check(lib.ce_bag_forward(ptr(weight), weight.shape[0], dim, ptr(indices), indices.numel(), ptr(offsets),
                                     int(offsets.dtype == torch.int64), num_bags, int(include_last), ptr(psw), mode,
                                     hook_features, ptr(out), stream_ptr()))
        # bwd_scale (scale_grad_by_freq): per-lookup factor of the backward only; it takes the place of psw there
        ctx.save_for_backward(indices, offsets, psw if bwd_scale is None else bwd_scale.contiguous())
        ctx.weight = weight
        ctx.args = (mode, include_last, hook_features, sparse, fused, num_bags)
        ctx.presorted = presorted
        ctx.masked = masked
        return out

    @staticmethod
    def backward(ctx, grad_out):
        indices, offsets, psw = ctx.saved_tensors
        weight = ctx.weight
        mode, include_last, hook_features, sparse, fused, num_bags = ctx.args
        grad_out = grad_out.contiguous()
        dim = weight.shape[1]
        off64 = int(offsets.dtype == torch.int64)
        nnz = indices.numel()
        gw = None
        if fused is not None and fused.lr is not None:
            # K13+K14 in one pass; the optimizer sees grad=None for the cache parameter
            with torch.no_grad():
                if fused.deterministic:
                    ws = fused.workspace(weight.shape[0], nnz, weight.device)
                    check(lib.ce_bag_backward_sgd_sorted(ptr(weight), weight.shape[0], dim, ptr(indices), nnz,
                                                         ptr(offsets), off64, num_bags, int(include_last), ptr(psw),
                                                         mode, hook_features, ptr(grad_out), float(fused.lr),
                                                         ptr(ws), ws.numel(), stream_ptr()))
                elif isinstance(ctx.presorted, SrcKeys) and ctx.presorted.ranges is not None:
                    check(lib.ce_bag_backward_sgd_presorted_src_excl(ptr(weight), weight.shape[0], dim, nnz,
                                                                     ptr(grad_out), float(fused.lr),
                                                                     ptr(ctx.presorted.keys),
                                                                     ptr(ctx.presorted.ranges), stream_ptr()))
                elif isinstance(ctx.presorted, SrcKeys):
                    check(lib.ce_bag_backward_sgd_presorted_src(ptr(weight), weight.shape[0], dim, nnz,
                                                                ptr(grad_out), float(fused.lr),
                                                                ptr(ctx.presorted.keys), stream_ptr()))
                elif ctx.presorted is not None:
                    check(lib.ce_bag_backward_sgd_presorted(ptr(weight), weight.shape[0], dim, ptr(indices), nnz,
                                                            ptr(offsets), off64, num_bags, int(include_last), ptr(psw),
                                                            mode, hook_features, ptr(grad_out), float(fused.lr),
                                                            ptr(ctx.presorted), stream_ptr()))
                else:
                    check(lib.ce_bag_backward_sgd(ptr(weight), weight.shape[0], dim, ptr(indices), nnz, ptr(offsets),
                                                  off64, num_bags, int(include_last), ptr(psw), mode, hook_features,
                                                  ptr(grad_out), float(fused.lr), stream_ptr()))
        elif sparse and COALESCED_SPARSE_GRAD and nnz > 0 and not torch.cuda.is_current_stream_capturing():
            # sparse=True (scripts/kaggle.sh:71 --use_sparse_embed_grad): the COO gradient is handed over COALESCED --
            # unique rows (ce_dedupe_bucket_rows), ascending, each with the sum of its lookups' gradient rows (the dense
            # backward kernel over the unique positions) -- so torch.optim.SGD's grad.coalesce() (a sort of every lookup
            # and a segmented sum over nnz x D floats, most of an unchanged trainer's step) has nothing left to do.
            # One host read (the number of unique rows), as coalesce() has too -- which is why a backward that is being
            # captured into a hipGraph takes the one-row-per-lookup form below instead.
            # The dedupe scratch (int32[rows] + int32[2 nnz]; nothing in it needs initialising) comes from the
            # caching allocator per call, so its lifetime is ordered on the stream this backward runs on: two
            # same-sized tables running their backwards on two streams do not share it.
            dev = weight.device
            R = weight.shape[0]
            ws = (torch.empty(R, dtype=torch.int32, device=dev), None,
                  torch.empty(max(2 * nnz, 1 << 16), dtype=torch.int32, device=dev))
            urows = torch.empty(nnz, dtype=torch.int64, device=dev)
            pos = torch.empty(nnz, dtype=torch.int64, device=dev)
            cnt = torch.empty(1, dtype=torch.int64, device=dev)
            check(lib.ce_dedupe_bucket_rows(ptr(indices), nnz, None, R, 1, ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(urows),
                                            ptr(pos), ptr(cnt), stream_ptr()))
            n_u = int(cnt.item())           # the ONE host sync of this path (torch's coalesce() has the same one)
            folded = torch.zeros(max(n_u, 1), dim, device=dev, dtype=torch.float32)
            if n_u:
                check(lib.ce_bag_backward_dense(ptr(folded), n_u, dim, ptr(pos), nnz, ptr(offsets), off64, num_bags,
                                                int(include_last), ptr(psw), mode, hook_features, ptr(grad_out),
                                                stream_ptr()))
            order = torch.argsort(urows[:n_u])
            gw = torch.sparse_coo_tensor(urows[:n_u][order].view(1, -1), folded[:n_u][order], weight.shape,
                                         is_coalesced=True, check_invariants=False)
        elif sparse:
            rows = torch.empty(nnz, dim, device=weight.device, dtype=torch.float32)
            check(lib.ce_bag_backward_rows(ptr(rows), None, dim, nnz, ptr(offsets), off64, num_bags, int(include_last),
                                           ptr(psw), mode, hook_features, ptr(grad_out), stream_ptr()))
            if ctx.masked:          # lookups masked to -1 (padding_idx): zero contribution at a valid index
                rows = rows * (indices >= 0).unsqueeze(1)
                indices = indices.clamp(min=0)
            gw = torch.sparse_coo_tensor(indices.view(1, -1), rows, weight.shape, check_invariants=False)
        else:
            gw = torch.zeros_like(weight)
            if isinstance(ctx.presorted, SrcKeys):
                check(lib.ce_bag_backward_dense_presorted_src(ptr(gw), weight.shape[0], dim, nnz, ptr(grad_out),
                                                              ptr(ctx.presorted.keys), stream_ptr()))
            elif ctx.presorted is not None:
                check(lib.ce_bag_backward_dense_presorted(ptr(gw), weight.shape[0], dim, ptr(indices), nnz,
                                                          ptr(offsets), off64, num_bags, int(include_last), ptr(psw),
                                                          mode, hook_features, ptr(grad_out), ptr(ctx.presorted),
                                                          stream_ptr()))
            else:
                check(lib.ce_bag_backward_dense(ptr(gw), weight.shape[0], dim, ptr(indices), nnz, ptr(offsets), off64,
                                                num_bags, int(include_last), ptr(psw), mode, hook_features,
                                                ptr(grad_out), stream_ptr()))
        gpsw = None
        if ctx.needs_input_grad[3]:
            # d loss / d per_sample_weights[j] = <grad_out[bag of j], weight[indices[j]]> (the combination with the
            # fused update was refused in forward)
            gpsw = torch.empty(nnz, device=weight.device, dtype=torch.float32)
            check(lib.ce_bag_backward_psw(ptr(weight), weight.shape[0], dim, ptr(indices), nnz, ptr(offsets), off64,
                                          num_bags, int(include_last), hook_features, ptr(grad_out), ptr(gpsw),
                                          stream_ptr()))
        return gw, None, None, gpsw, None, None, None, None, None, None, None, None, None


# forward from the window's source-row keys when they were built for the one-id-per-bag layout (False: always the
# gather-shaped kernel over slots + offsets); sparse=True hands torch a coalesced COO gradient (False: one value row per
# lookup).  Module constants (they were environment switches until round 6): a test or a probe may set them.
FORWARD_FROM_KEYS = True
COALESCED_SPARSE_GRAD = True


class _BagMaxFn(torch.autograd.Function):
    """mode='max' (ce_bag_forward_max / ce_bag_backward_max): dense gradient or the fused SGD update."""

    @staticmethod
    def forward(ctx, weight, indices, offsets, include_last, hook_features, fused):
        _lib.require_gpu()
        assert weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous()
        num_bags = offsets.numel() - 1 if include_last else offsets.numel()
        dim = weight.shape[1]
        shape = (num_bags // hook_features, hook_features, dim) if hook_features else (num_bags, dim)
        out = torch.empty(shape, device=weight.device, dtype=torch.float32)
        pos = torch.empty(num_bags, dim, device=weight.device, dtype=torch.int32)
        check(lib.ce_bag_forward_max(ptr(weight), weight.shape[0], dim, ptr(indices), indices.numel(), ptr(offsets),
                                     int(offsets.dtype == torch.int64), num_bags, int(include_last), hook_features,
                                     ptr(out), ptr(pos), stream_ptr()))
        ctx.save_for_backward(indices, pos)
        ctx.weight = weight
        ctx.args = (num_bags, hook_features, fused)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        indices, pos = ctx.saved_tensors
        weight = ctx.weight
        num_bags, hook_features, fused = ctx.args
        grad_out = grad_out.contiguous()
        dim = weight.shape[1]
        if fused is not None and fused.lr is not None:
            with torch.no_grad():
                check(lib.ce_bag_backward_max(ptr(weight), weight.shape[0], dim, ptr(indices), indices.numel(), num_bags,
                                              hook_features, ptr(grad_out), ptr(pos), -float(fused.lr), stream_ptr()))
            return None, None, None, None, None, None
        gw = torch.zeros_like(weight)
        check(lib.ce_bag_backward_max(ptr(gw), weight.shape[0], dim, ptr(indices), indices.numel(), num_bags,
                                      hook_features, ptr(grad_out), ptr(pos), 1.0, stream_ptr()))
        return gw, None, None, None, None, None


def renorm_rows_(weight: torch.Tensor, indices: torch.Tensor, max_norm: float, norm_type: float = 2.0) -> None:
    """torch.embedding_renorm_ on the rows `indices` names (out-of-range entries are skipped), in place."""
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous()
    idx = indices.reshape(-1).long().contiguous()
    ws = torch.empty(lib.ce_rows_renorm_workspace(weight.shape[0]), dtype=torch.uint8, device=weight.device)
    with torch.no_grad():
        check(lib.ce_rows_renorm(ptr(weight), weight.shape[0], weight.shape[1], ptr(idx), idx.numel(), float(max_norm),
                                 float(norm_type), ptr(ws), ws.numel(), stream_ptr()))


class FusedSGD:
    """Switch for the fused backward+SGD path of one embedding module.

    lr=None disables fusion (the module behaves exactly like nn.EmbeddingBag under
    torch.optim.SGD); deterministic=True uses the sorted segmented update instead of atomics."""

    def __init__(self, lr: Optional[float] = None, deterministic: bool = False):
        self.lr = lr
        self.deterministic = deterministic
        self._ws = None

    def workspace(self, num_rows: int, nnz: int, device) -> torch.Tensor:
        need = lib.ce_bag_backward_sgd_sorted_workspace(num_rows, nnz)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws


def embedding_bag(indices: torch.Tensor, weight: torch.Tensor, offsets: Optional[torch.Tensor] = None,
                  max_norm: Optional[float] = None, norm_type: float = 2.0, scale_grad_by_freq: bool = False,
                  mode: str = "mean", sparse: bool = False, per_sample_weights: Optional[torch.Tensor] = None,
                  include_last_offset: bool = False, padding_idx: Optional[int] = None, *,
                  hook_features: int = 0, fused_sgd: Optional[FusedSGD] = None,
                  presorted: Union[torch.Tensor, SrcKeys, None] = None, masked_indices: bool = False,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    # out: write the pooled output into this tensor (contiguous fp32, the shape the call would allocate) and return it --
    # a static output buffer for steps replayed from a hipGraph, possibly one chosen by pick_fast_buffer
    # masked_indices: the caller already replaced ignored lookups (padding) by -1 -- the kernels skip them; the
    # sparse=True backward then parks their (zero) gradient rows at index 0 so the COO tensor stays valid
    if mode not in _MODES and mode != "max":
        raise NotImplementedError(f"mode={mode!r}: 'sum', 'mean' and 'max' are implemented")
    if per_sample_weights is not None:
        if mode != "sum":
            raise NotImplementedError("embedding_bag: per_sample_weights was not None. per_sample_weights is "
                                      f"only supported for mode='sum' (got mode='{mode}').")
        if per_sample_weights.dtype != torch.float32 or not per_sample_weights.is_contiguous() or \
                per_sample_weights.dim() != 1:
            per_sample_weights = per_sample_weights.reshape(-1).float().contiguous()     # (keeps the autograd link)
    indices, offsets, include_last_offset, num_bags = _prep(indices, offsets, include_last_offset)
    if max_norm is not None:
        # F.embedding_bag renormalises the rows the input names, in place, before it looks them up
        renorm_rows_(weight.detach(), indices, max_norm, norm_type)
    if mode == "max":
        # torch: no per_sample_weights (above), no sparse gradient, no scale_grad_by_freq for 'max'
        if sparse:
            raise RuntimeError("embedding_bag: sparse gradients are not supported with mode='max'")
        if scale_grad_by_freq:
            raise RuntimeError("embedding_bag: scale_grad_by_freq is not supported with mode='max'")
        if presorted is not None:
            raise NotImplementedError("presorted keys serve the sum / mean backward only")
        if hook_features and num_bags % hook_features:
            raise ValueError("hook_features must divide the number of bags")
        if padding_idx is not None:
            if padding_idx < 0:
                padding_idx += weight.shape[0]
            indices = torch.where(indices == padding_idx, torch.full_like(indices, -1), indices)
        if out is not None:
            raise NotImplementedError("out= with mode='max'")
        return _BagMaxFn.apply(weight, indices, offsets, bool(include_last_offset), int(hook_features), fused_sgd)
    if per_sample_weights is not None and per_sample_weights.numel() != indices.numel():
        raise ValueError("per_sample_weights must have the same number of elements as input")
    bwd_scale = None
    if mode == "mean" and (padding_idx is not None or masked_indices or scale_grad_by_freq):
        # 'mean' over the NON-padding entries of a bag (torch excludes padding from the sum and from the count), and
        # 'mean' with scale_grad_by_freq: the kernels' mean divides by the bag length, so these run as a 'sum' with
        # per-lookup weights 1 / (valid entries of the bag) -- a handful of torch ops, off the benchmarked path
        if presorted is not None:
            raise NotImplementedError("padding_idx / scale_grad_by_freq with mode='mean' cannot be combined with "
                                      "presorted keys")
        if padding_idx is not None:
            pi = padding_idx + weight.shape[0] if padding_idx < 0 else padding_idx
            indices = torch.where(indices == pi, torch.full_like(indices, -1), indices)
            padding_idx = None
            masked_indices = True
        nnz = indices.numel()
        ends = offsets[1:] if include_last_offset else torch.cat(
            [offsets[1:], torch.full((1,), nnz, dtype=offsets.dtype, device=offsets.device)])
        lens = (ends - offsets[:num_bags]).long()
        bag_of = torch.repeat_interleave(torch.arange(num_bags, device=indices.device), lens, output_size=nnz)
        valid = ((indices >= 0) & (indices < weight.shape[0])).to(torch.float32)
        cnt = torch.zeros(num_bags, device=indices.device, dtype=torch.float32).index_add_(0, bag_of, valid)
        per_sample_weights = valid / cnt.clamp_(min=1.0)[bag_of]
        mode = "sum"
    if scale_grad_by_freq:
        # torch: the gradient of a row is divided by the number of times the row occurs in the mini-batch
        if mode != "sum":
            raise NotImplementedError("scale_grad_by_freq is implemented for mode='sum' and 'mean'")
        if presorted is not None:
            raise NotImplementedError("scale_grad_by_freq cannot be combined with presorted keys")
        _, inv, cnt = torch.unique(indices, return_inverse=True, return_counts=True)
        bwd_scale = (1.0 / cnt.to(torch.float32))[inv]
        if per_sample_weights is not None:
            bwd_scale = bwd_scale * per_sample_weights
    if padding_idx is not None:
        # entries equal to padding_idx take no part in the reduction and receive no gradient: the kernels skip
        # out-of-range rows, so they are redirected to -1
        if mode != "sum":
            raise NotImplementedError("padding_idx is implemented for mode='sum' only (mean would need the "
                                      "per-bag count of non-padding entries)")
        if presorted is not None:
            # the keys hold the rows they were built from: mask the slots BEFORE presort_window instead
            raise NotImplementedError("padding_idx cannot be combined with presorted keys (mask the indices to -1 "
                                      "before building the keys)")
        if padding_idx < 0:
            padding_idx += weight.shape[0]
        indices = torch.where(indices == padding_idx, torch.full_like(indices, -1), indices)
    if hook_features and num_bags % hook_features:
        raise ValueError("hook_features must divide the number of bags")
    if isinstance(presorted, SrcKeys):
        if mode != "sum" or per_sample_weights is not None:
            raise ValueError("source-row keys (presort_window(..., offsets=...)) hold no per-lookup scale: they "
                             "need mode='sum' without per_sample_weights")
        if (presorted.num_bags, bool(presorted.include_last_offset), int(presorted.hook_features)) != \
                (num_bags, bool(include_last_offset), int(hook_features)):
            raise ValueError("source-row keys were built for another bag layout "
                             f"{tuple(presorted[1:4])} than this call's {(num_bags, include_last_offset, hook_features)}")
        k = presorted.keys
        if presorted.ranges is not None:
            r = presorted.ranges
            assert r.is_cuda and r.dtype == torch.int64 and r.is_contiguous() and \
                r.numel() == 2 * (k.numel() // 16384), "ranges must come from presort_window(..., ids=...)"
        assert k.is_cuda and k.dtype == torch.int64 and k.is_contiguous() and \
            k.numel() == lib.ce_bag_presort_len(indices.numel()), "keys must come from presort_window"
    elif presorted is not None:
        assert presorted.is_cuda and presorted.dtype == torch.int64 and presorted.is_contiguous() and \
            presorted.numel() == lib.ce_bag_presort_len(indices.numel()), "presorted must come from presort_slots"
    if masked_indices and mode != "sum":
        raise NotImplementedError("masked (padding) lookups are implemented for mode='sum' only (mean would need "
                                  "the per-bag count of non-padding entries)")
    return _BagFn.apply(weight, indices, offsets, per_sample_weights, _MODES[mode], bool(include_last_offset),
                        int(hook_features), bool(sparse), fused_sgd, presorted, bwd_scale,
                        padding_idx is not None or bool(masked_indices), None if out is None else [out])


def probe_rows(buf: torch.Tensor, fold: int, reps: int = 6):
    """(us per pass of row stores, us per pass of row loads) over buf [rows, dim] visited `fold` rows apart
    (ce_probe_rows); buf's contents are overwritten"""
    import ctypes
    assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()
    flat = buf.view(-1, buf.shape[-1])
    w, r = ctypes.c_double(), ctypes.c_double()
    check(lib.ce_probe_rows(ptr(flat), flat.shape[0], flat.shape[1], int(fold), int(reps), ctypes.byref(w), ctypes.byref(r),
                            stream_ptr()))
    return w.value, r.value


def _time_work(work, buf: torch.Tensor, reps: int) -> float:
    """us per call of work(buf): `reps` calls enqueued while the GPU is kept busy by a spin kernel (so that they run
    back to back however long the host takes to launch them), timed by two events.  Deliberately NOT a captured
    hipGraph: two dozen capture / destroy cycles leave the runtime with a different assignment of streams to hardware
    queues for everything created afterwards (measured: a prefetch_num = 1 pipeline built after them ran 5 % slower)."""
    work(buf)                                    # once, untimed: lazy initialisation, first-touch of the candidate
    torch.cuda.synchronize(buf.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(6e5) * reps)           # ~0.3 ms per call at 2 GHz: room for an autograd step's host time
    e0.record()
    for _ in range(reps):
        work(buf)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def pick_fast_buffer(shape, device, fold: int, candidates: int = 5, use: str = "write", work=None, reps: int = 8):
    """A contiguous fp32 tensor of `shape` ([..., D]) that is FAST for the hook-folded access pattern: rows visited
    `fold` rows apart, every row on another page -- what the forward's stores into a [B, F, D] output (fold = F) and the
    streaming backward's loads of the upstream gradient do.  How fast depends on how the allocation happens to be
    mapped, not on its address: the same forward launch takes 38 or 45 us on two buffers of one process
    (profiles/r05_alloc_lottery.txt), which is also why the same bench line differed by 15 % in its forward between
    processes.  So: allocate `candidates` buffers, time each, keep the fastest, give the others back.
    use = "write" (an output buffer) or "read" (a gradient buffer): what is timed is the pattern alone (ce_probe_rows,
    a few passes; the search stops at the first candidate clearly of the fast kind).  work = callable(buf): what is
    timed is the caller's own work on the candidate instead -- work(buf) enqueues it on the current stream, free of
    side effects; `reps` calls run back to back behind a spin kernel -- and every candidate is tried ("read"
    candidates are zero-filled first: a gradient buffer must not hold the allocator's leftovers).
    Returns (tensor, {"us": [...], "picked": i}).  For static buffers of graph-captured steps; torch decides where
    everything else lives."""
    assert use in ("write", "read")
    bufs, us, fresh = [], [], []
    for _ in range(max(1, int(candidates))):
        r0 = torch.cuda.memory_reserved(device)
        b = torch.empty(shape, dtype=torch.float32, device=device)       # (all candidates stay alive while the search runs:
        bufs.append(b)                                                  # every one is another allocation)
        fresh.append(int(torch.cuda.memory_reserved(device) > r0))
        if work is not None:
            if use == "read":
                b.zero_()
            us.append(_time_work(work, b, reps))
            continue
        w, r = probe_rows(b, fold)
        us.append(w if use == "write" else r)
        if len(us) >= 4:
            med = sorted(us)[len(us) // 2]
            if us[-1] < 0.94 * med:          # the fast kind is ~8 % faster than the common kind in this probe: found one
                break
    best = min(range(len(bufs)), key=lambda i: us[i])
    keep = bufs[best]
    del bufs, b
    # the losers go back to the DRIVER, not into torch's cache: whatever is allocated next would otherwise be carved
    # out of them (measured: a window object built after a 12-candidate search ran 7 % slower at prefetch_num = 1)
    torch.cuda.empty_cache()
    return keep, {"us": [round(u, 2) for u in us], "picked": best, "new_segment": fresh}


def is_identity_layout(offsets: torch.Tensor, include_last_offset: bool) -> bool:
    """True when the offsets describe one id per bag, in order (offsets == arange): ONE host comparison, meant for
    set-up time of a pipeline whose batches all share these offsets."""
    if offsets.dim() != 1:
        return False
    n = offsets.numel()
    return bool(torch.equal(offsets.long(), torch.arange(n, device=offsets.device)))


def presort_len(n: int) -> int:
    """elements of the key tensor presort_slots writes for n lookups (n rounded up to whole 16384-lookup segments)"""
    return int(lib.ce_bag_presort_len(int(n)))


def presort_slots(slots: torch.Tensor, num_rows: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Segment-sorted keys of one batch's slots for the fused backward (ce_bag_presort): int64 tensor holding the
    bits of uint64 keys, presort_len(n) long.  Run it once per prefetch window on the cache-op stream, then pass
    the batch's keys to embedding_bag(presorted=...)."""
    flat = slots.reshape(-1).contiguous()
    if out is None:
        out = torch.empty(presort_len(flat.numel()), dtype=torch.int64, device=flat.device)
    assert out.dtype == torch.int64 and out.numel() == presort_len(flat.numel()) and out.is_contiguous()
    check(lib.ce_bag_presort(ptr(flat), flat.numel(), int(num_rows), ptr(out), stream_ptr()))
    return out


def presort_window(slots: torch.Tensor, num_rows: int, keys_out: Optional[torch.Tensor] = None, *,
                   offsets: Optional[torch.Tensor] = None, include_last_offset: bool = False,
                   hook_features: int = 0, ids: Optional[torch.Tensor] = None,
                   ranges_out: Optional[torch.Tensor] = None, identity_bags: bool = False):
    """Grouped keys for the P equal-sized batches of a prefetch window in one launch (ce_bag_presort_window).
    slots: [P, n] int64 (the cache op's output) -> keys [P, presort_len(n)]; row b is what
    embedding_bag(presorted=...) takes for batch b.

    offsets given (1-D: shared by the batches, or [P, len]: one row per batch): source-row keys
    (ce_bag_presort_window_src) for mode='sum' without per-sample weights -- returns a list of P SrcKeys instead,
    which the backward streams over without touching offsets or indices again.  The keys ARE the batch as far as the
    backward is concerned: pass them only to the embedding_bag call that uses the same slots and the same offsets
    (the bag layout is checked, the contents cannot be).

    ids given as well ([P, n] int64: the ids `slots` was computed from, position by position): the keys also mark the
    rows a single lane group of the backward owns, and every SrcKeys carries the id range of its segments
    (ce_bag_presort_window_src_excl) -- the fused-SGD backward then updates those rows without atomics.

    identity_bags=True: the caller has checked that `offsets` is arange(n + 1) -- one id per bag, in order, what every
    Criteo / Avazu batch is (is_identity_layout) -- so the kernel need not read the offsets to establish it."""
    assert slots.dim() == 2 and slots.is_contiguous() and slots.dtype == torch.int64
    P, n = slots.shape
    klen = presort_len(n)
    if keys_out is None:
        keys_out = torch.empty(P, klen, dtype=torch.int64, device=slots.device)
    assert keys_out.is_contiguous() and keys_out.numel() == P * klen and keys_out.dtype == torch.int64
    if offsets is None:
        check(lib.ce_bag_presort_window(ptr(slots), n, P, int(num_rows), ptr(keys_out), stream_ptr()))
        return keys_out.view(P, klen)
    assert offsets.is_cuda and offsets.dtype in (torch.int32, torch.int64) and offsets.is_contiguous()
    assert offsets.dim() == 1 or (offsets.dim() == 2 and offsets.shape[0] == P)
    per = offsets.shape[-1]
    num_bags = per - 1 if include_last_offset else per
    if hook_features and num_bags % hook_features:
        raise ValueError("hook_features must divide the number of bags")
    kv = keys_out.view(P, klen)
    if identity_bags:
        assert num_bags == n, "identity_bags needs one id per bag"
    off_ptr = None if identity_bags else ptr(offsets)
    if ids is None:
        check(lib.ce_bag_presort_window_src(ptr(slots), n, P, int(num_rows), off_ptr,
                                            int(offsets.dtype == torch.int64), per if offsets.dim() == 2 else 0,
                                            num_bags, int(include_last_offset), int(hook_features), ptr(keys_out),
                                            stream_ptr()))
        return [SrcKeys(kv[b], num_bags, bool(include_last_offset), int(hook_features), None, bool(identity_bags))
                for b in range(P)]
    assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == P * n
    segs = klen // 16384
    if ranges_out is None:
        ranges_out = torch.empty(P, segs, 2, dtype=torch.int64, device=slots.device)
    assert ranges_out.is_contiguous() and ranges_out.dtype == torch.int64 and ranges_out.numel() == P * segs * 2
    check(lib.ce_bag_presort_window_src_excl(ptr(slots), n, P, int(num_rows), off_ptr,
                                             int(offsets.dtype == torch.int64), per if offsets.dim() == 2 else 0,
                                             num_bags, int(include_last_offset), int(hook_features), ptr(ids),
                                             ptr(keys_out), ptr(ranges_out), stream_ptr()))
    rv = ranges_out.view(P, segs, 2)
    return [SrcKeys(kv[b], num_bags, bool(include_last_offset), int(hook_features), rv[b], bool(identity_bags))
            for b in range(P)]
