"""CPU ORACLE (test infrastructure -- never imported by the product path).

The numerics of the hot path are stock PyTorch on the reference side:
``F.embedding_bag(slots, cuda_cached_weight, offsets, ..., mode, sparse, per_sample_weights,
include_last_offset)`` reached from recsys/models/dlrm.py:99-110 and
benchmark/benchmark_cache.py:62, its autograd backward, and ``torch.optim.SGD.step``
(recsys/dlrm_main.py:274-279,455-461).  torch-CPU is importable here, so the oracle for
pooled outputs / grads / post-step weights is that very code on CPU tensors (fp32), plus a
plain numpy loop version used to pin the torch call on small cases.

PARITY UNPINNED by the reference's own tests (there are none); tolerance for fp32 values
is 1e-5 relative (BASELINE.json north_star), exact when every bag holds one id.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F


def bag_forward(weight: torch.Tensor, indices: torch.Tensor, offsets: torch.Tensor,
                per_sample_weights: Optional[torch.Tensor] = None, mode: str = "sum",
                include_last_offset: bool = True) -> torch.Tensor:
    """Pooled embedding per bag on CPU fp32 -- same call the reference makes (A.7)."""
    return F.embedding_bag(indices.long().cpu(), weight.float().cpu(), offsets.long().cpu(),
                           mode=mode, per_sample_weights=per_sample_weights,
                           include_last_offset=include_last_offset)


def bag_forward_numpy(weight: np.ndarray, indices: np.ndarray, offsets: np.ndarray,
                      per_sample_weights: Optional[np.ndarray] = None, mode: str = "sum",
                      include_last_offset: bool = True) -> np.ndarray:
    """Loop restatement (small cases only) used to pin bag_forward."""
    offsets = np.asarray(offsets, dtype=np.int64)
    if not include_last_offset:
        offsets = np.concatenate([offsets, [len(indices)]])
    nb = len(offsets) - 1
    out = np.zeros((nb, weight.shape[1]), dtype=np.float32)
    for b in range(nb):
        lo, hi = offsets[b], offsets[b + 1]
        acc = np.zeros(weight.shape[1], dtype=np.float32)
        for j in range(lo, hi):
            w = np.float32(1.0) if per_sample_weights is None else np.float32(per_sample_weights[j])
            acc = acc + w * weight[indices[j]]
        if mode == "mean" and hi > lo:
            acc = acc / np.float32(hi - lo)
        out[b] = acc
    return out


def bag_backward_dense(num_rows: int, indices: torch.Tensor, offsets: torch.Tensor,
                       grad_out: torch.Tensor, per_sample_weights: Optional[torch.Tensor] = None,
                       mode: str = "sum", include_last_offset: bool = True,
                       dim: Optional[int] = None) -> torch.Tensor:
    """dW (dense [num_rows, D]) from torch autograd on CPU."""
    D = grad_out.shape[1] if dim is None else dim
    w = torch.zeros(num_rows, D, dtype=torch.float32, requires_grad=True)
    out = F.embedding_bag(indices.long().cpu(), w, offsets.long().cpu(), mode=mode,
                          per_sample_weights=per_sample_weights,
                          include_last_offset=include_last_offset)
    out.backward(grad_out.float().cpu())
    return w.grad.detach()


def sgd_step(weight: torch.Tensor, indices: torch.Tensor, offsets: torch.Tensor,
             grad_out: torch.Tensor, lr: float, per_sample_weights: Optional[torch.Tensor] = None,
             mode: str = "sum", include_last_offset: bool = True, sparse: bool = True) -> torch.Tensor:
    """One forward/backward/SGD.step on CPU exactly as the reference trainer does
    (sparse COO grad when --use_sparse_embed_grad, scripts/kaggle.sh:71)."""
    w = torch.nn.Parameter(weight.detach().float().cpu().clone())
    opt = torch.optim.SGD([w], lr=lr)
    out = F.embedding_bag(indices.long().cpu(), w, offsets.long().cpu(), mode=mode, sparse=sparse,
                          per_sample_weights=per_sample_weights,
                          include_last_offset=include_last_offset)
    opt.zero_grad()
    out.backward(grad_out.float().cpu())
    opt.step()
    return w.detach()


def cpu_train_step_inplace(param: torch.nn.Parameter, opt: torch.optim.Optimizer, indices: torch.Tensor,
                           offsets: torch.Tensor, grad_out: torch.Tensor, mode: str = "sum") -> torch.Tensor:
    """One embedding training step of the reference's CPU path on a table that is updated in place
    (used by bench.py's cpu_baseline leg, where cloning a 91 GB table per step is not an option)."""
    out = F.embedding_bag(indices, param, offsets, mode=mode, sparse=True, include_last_offset=True)
    opt.zero_grad(set_to_none=True)
    out.backward(grad_out)
    opt.step()
    return out.detach()
