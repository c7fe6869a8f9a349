"""Multi-batch prefetch window: the `_train` prefetch block of the reference
(recsys/dlrm_main.py:243-262, pics/prefetch.png) as a reusable object.

Reference semantics (overlap=False): every `prefetch_num` iterations the ids of the next
P batches are concatenated, ONE prepare_ids makes all their rows resident (none can be
evicted before use because the whole window is in the evict backlist), the returned slots
are split back per batch and the forwards run with cache_op=False.

overlap=True is the build's extension: the cache op of window k+1 runs on a
side HIP stream while window k trains.  Rows of window k must then stay protected during
window k+1's victim selection, which is `protect_depth=1` in the manager; the capacity
condition becomes unique(window k U window k+1) <= cuda_row_num.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .cached_embedding import CachedEmbeddingBag


class PrefetchWindow:
    def __init__(self, embed: CachedEmbeddingBag, prefetch_num: int = 1, overlap: bool = False):
        assert prefetch_num >= 1
        self.embed = embed
        self.mgr = embed.cache_weight_mgr
        self.P = prefetch_num
        self.overlap = overlap
        self._side: Optional[torch.cuda.Stream] = None
        self._pending = None   # (event, [slots per batch])
        if overlap:
            self._side = torch.cuda.Stream(device=self.mgr.device)
            self.mgr.set_protect_depth(1)
            self.mgr.strict = False   # no host sync inside the pipelined cache op

    @torch.no_grad()
    def _cache_op(self, values: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        counts = [int(v.numel()) for v in values]
        cat = values[0] if len(values) == 1 else torch.cat(list(values))
        slots = self.mgr.prepare_ids(cat)
        # split by per-batch id counts (torch.chunk in the reference is only right for equal sizes, B#13)
        return list(torch.split(slots, counts))

    def prepare(self, values: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """Synchronous window op on the current stream (reference behaviour)."""
        assert 1 <= len(values) <= self.P
        return self._cache_op(values)

    def submit(self, values: Sequence[torch.Tensor]) -> None:
        """Start the cache op for the NEXT window on the side stream (overlap=True)."""
        assert self.overlap and self._pending is None
        cur = torch.cuda.current_stream(self.mgr.device)
        self._side.wait_stream(cur)          # ids were produced on the current stream
        with torch.cuda.stream(self._side):
            slots = self._cache_op(values)
            ev = torch.cuda.Event()
            ev.record(self._side)
        for v in values:
            v.record_stream(self._side)
        self._pending = (ev, slots)

    def collect(self) -> List[torch.Tensor]:
        """Slots of the submitted window; the current stream waits for the side stream."""
        assert self._pending is not None
        ev, slots = self._pending
        self._pending = None
        cur = torch.cuda.current_stream(self.mgr.device)
        cur.wait_event(ev)
        for s in slots:
            s.record_stream(cur)
        return slots
