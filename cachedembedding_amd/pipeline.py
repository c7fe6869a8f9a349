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

from . import _lib
from .cached_embedding import CachedEmbeddingBag
from .functional import SrcKeys, is_identity_layout, presort_len, presort_window
from .tracing import phase


def make_side_stream(device, cache_cus: int = 0, total_cus: int = 256) -> torch.cuda.Stream:
    """Stream for the overlapped cache op.  cache_cus > 0 restricts it to that many CUs (taken from the top
    of the CU range) through hipExtStreamCreateWithCUMask, leaving the rest of the chip to training."""
    if cache_cus <= 0:
        return torch.cuda.Stream(device=device)
    import ctypes
    words = (total_cus + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in range(total_cus - cache_cus, total_cus):
        mask[cu // 32] |= (1 << (cu % 32))
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        _lib.check(_lib.lib.ce_stream_create_cu_mask(mask, words, ctypes.byref(out)))
    return torch.cuda.ExternalStream(out.value, device=device)


# ids per cache op from which the worker transport pays: it costs a host round trip (event wake-up, kernel launch
# from the worker thread, release: >= 0.1 ms whatever the size) and buys the write-back traffic off the CUs and the
# admission out of the cache-op stream's critical path; the zero-copy swap kernel costs ~0.03 ms per 1000 rows moved.
# Measured: Kaggle 5 % P = 1 (426 k ids, ~25 k rows per call) 0.82 G lookups/s zero-copy / 0.93 G worker; P = 8
# (3.4 M ids, ~50 k rows) 2.0 G / 2.6 G; Avazu B = 2048 (45 k ids, ~2.6 k rows) stays zero-copy.
AUTO_WORKER_MIN_IDS = 300_000


def pick_transport(transport: Optional[str], ids_per_call: int) -> Optional[str]:
    if transport == "auto":
        return "worker" if ids_per_call >= AUTO_WORKER_MIN_IDS else "zerocopy"
    return transport


# Owner-exclusive rows in the fused backward (presort_window(..., ids=...)): measured round 3 -- the presort that
# marks them costs 4x (208 vs 50 us per window) and the backward gains 2-3 us per batch at best, so
# the window pipelines only use it when this module constant is set (a probe's business).
EXCLUSIVE_ROWS = False
# the window's keys written by the cache op's last kernel (ce_cache_prepare_ids_keys) instead of by a presort launch
# behind it (False: two calls)
FUSED_WINDOW_KEYS = True


ARRANGEMENTS = ("overlap", "interleaved")


class ArrangementTrial:
    """Which arrangement of the next window's cache op trains faster HERE, NOW -- measured by the library while it
    trains, with hipEvents on the training stream and no host wait (recsys/dlrm_main.py:243-282 is the loop it sits in).

    'overlap': the cache op of window k+1 runs on a side stream beside the steps of window k (north_star's design
    point).  'interleaved': its two halves run on the training stream around those steps, so no kernel runs beside
    another and only the PCIe admission overlaps.  Side by side the bag kernels and the cache-op chain slow each other
    by about as much as the overlap hides; which way the balance tips depends on what the host's side of the row swap
    does that hour (1.16 against 1.24 ms per window one morning, 2.03 against 1.30 on another box).
    So the window objects measure: blocks of `block_windows` windows, `rounds` per arrangement in turn, starting with
    the steady one ('interleaved': its blocks differ by < 1 %); the first `settle` windows of a block (the hand-over
    between arrangements, pipeline fill) are not timed; with three rounds the arrangement with the faster MEDIAN block is
    kept, with fewer the one whose SLOWER block is faster (blocks of one arrangement differ by up to 30-40 % on a shared
    host: choosing by the better block picked the side stream on one lucky block, choosing by the worse one picked it
    on one unlucky block of the other).  Every window of the trial trains for real.  `retrial_every` windows later the trial runs again."""

    def __init__(self, steps_per_window: int, block_windows: int = 0, rounds: int = 3, settle: int = 4,
                 retrial_every: int = 16384, reduce_fn=None, decide_lag: int = 2, event_factory=None):
        # reduce_fn (several ranks training one model, parallel.GraphedShardedWindow): a COLLECTIVE that takes the
        # 2 x rounds block times of this rank and returns their maximum over the ranks.  The verdict is then reached
        # from the same numbers on every rank, and at the same window: `decide_lag` windows behind the last block (by
        # then its events have long completed, so the wait for them costs nothing), never "whenever this rank happens
        # to find its events done" -- the ranks would call the collective in different windows.
        self.reduce_fn, self.decide_lag = reduce_fn, int(decide_lag)
        self._event = event_factory or (lambda: torch.cuda.Event(enable_timing=True))
        self._lag = 0
        self.block_windows = int(block_windows) if block_windows else max(8, -(-256 // max(1, steps_per_window)))
        self.settle = min(int(settle), self.block_windows - 2)
        self.rounds, self.retrial_every = int(rounds), int(retrial_every)
        self.decided: Optional[str] = None
        self.trials = 0
        self.history: List[dict] = []
        self._start()

    def _start(self) -> None:
        self._order = ["interleaved", "overlap"] * self.rounds
        self._blk = 0
        # events only where a block needs them -- behind its `settle`-th window and behind its last one (a mark per
        # window cost a prefetch_num = 1 pipeline whose launch thread is the bottleneck 10 us of every step, and the
        # trial then measured its own marks: Avazu B = 2048 0.22 ms per step in both arrangements where the timed
        # regions say 0.146 against 0.124)
        self._n = 0                               # marks of the current block so far (mark 0 = in front of its first window)
        self._first: Optional[torch.cuda.Event] = None
        self._blocks: List[tuple] = []            # (mode, first timed event, last event, windows between them)
        self._since = 0
        self._lag = 0

    @property
    def mode(self) -> str:
        """arrangement for the cache ops submitted from now on"""
        if self._blk < len(self._order):
            return self._order[self._blk]
        return self.decided or "interleaved"

    @property
    def running(self) -> bool:
        return self._blk < len(self._order) or (self.decided is None and bool(self._blocks))

    @property
    def blocks_enqueued(self) -> bool:
        """every block of the running trial has been trained (its verdict only waits for their events)"""
        return self._blk >= len(self._order)

    def reset_block(self) -> None:
        """a window was trained in part / out of order: the current block starts over"""
        self._n = 0
        self._first = None

    def window_done(self, stream) -> str:
        """call when a whole window has been enqueued on `stream`; returns the arrangement to use from now on"""
        if self._blk < len(self._order):
            # mark `idx` of the block sits behind its idx-th window (mark 0: behind the window before it -- the previous
            # block's last mark, or, on a cold start, the first window_done call itself); only marks `settle` and
            # `block_windows` are ever read, so only those are recorded
            idx = self._n
            self._n += 1
            if idx == self.settle or idx == self.block_windows:
                ev = self._event()
                ev.record(stream)
                if idx == self.settle:
                    self._first = ev
                else:
                    self._blocks.append((self._order[self._blk], self._first, ev, self.block_windows - self.settle))
                    # the block's last mark doubles as the next block's mark 0 (that block settles anyway)
                    self._n = 1
                    self._first = ev if self.settle == 0 else None
                    self._blk += 1
        elif self.decided is None:
            if self.reduce_fn is None:
                self.poll()
            else:
                self._lag += 1
                if self._lag >= self.decide_lag:
                    self.poll(wait=True)
        else:
            self._since += 1
            if self.retrial_every and self._since >= self.retrial_every:
                self.decided_before = self.decided
                self.decided = None
                self._start()
        return self.mode

    def poll(self, wait: bool = False) -> Optional[str]:
        """decide once every block's events have completed (wait=True: synchronise on them)"""
        if self.decided is not None or self._blk < len(self._order):
            return self.decided
        if self.reduce_fn is not None and not wait:
            return None                      # (a collective verdict is only ever reached in window_done's own window)
        if wait:
            self._blocks[-1][2].synchronize()
        if not all(b[2].query() for b in self._blocks):
            return None
        per_block = [e0.elapsed_time(e1) / n for _, e0, e1, n in self._blocks]
        if self.reduce_fn is not None:
            per_block = list(self.reduce_fn(per_block))
        ms = {m: [] for m in ARRANGEMENTS}
        for (mode, _, _, _), v in zip(self._blocks, per_block):
            ms[mode].append(v)
        # three blocks per arrangement: the MEDIAN decides (one block in five is off by 20-40 % on these shared hosts,
        # whichever arrangement it is: the minimum picks a lucky block, the maximum an unlucky one); two: the slower
        def score(v):
            v = sorted(v)
            return v[len(v) // 2] if len(v) >= 3 else v[-1]
        self.decided = min(ms, key=lambda m: score(ms[m]))
        self.trials += 1
        self.history.append({"chosen": self.decided, "ms_per_window": {m: [round(v, 4) for v in ms[m]] for m in ms}})
        self._blocks = []
        return self.decided

    def report(self) -> dict:
        last = self.history[-1] if self.history else {}
        return {"mode": self.decided, "chosen_by": (
            f"the library (pipeline.ArrangementTrial): {self.block_windows}-window blocks, {self.rounds} per arrangement in "
            f"turn while training, the first {self.settle} windows of a block untimed, hipEvents on the training stream; the "
            "arrangement with the faster median block is kept (fewer than 3 rounds: the one whose slower block is faster)"), "trial_ms_per_window": last.get("ms_per_window"),
            "trials": self.trials}


def cat_window(values: Sequence[torch.Tensor]) -> torch.Tensor:
    """The window's ids as one tensor (recsys/dlrm_main.py:259 `torch.cat(sparse_values)`) -- without the copy when the
    batches already ARE consecutive pieces of one contiguous int64 tensor (a loader that hands out a window at a time, the
    bench's generator): then the view over all of them is returned."""
    if len(values) == 1:
        return values[0]
    first = values[0]
    if first.dtype == torch.int64 and first.dim() == 1 and first.is_contiguous():
        n, st, end = first.numel(), first.untyped_storage(), first.data_ptr()
        ok = True
        for v in values:
            if not (v.dtype == torch.int64 and v.dim() == 1 and v.is_contiguous() and v.data_ptr() == end
                    and v.untyped_storage().data_ptr() == st.data_ptr()):
                ok = False
                break
            end += 8 * v.numel()
        if ok:
            total = sum(int(v.numel()) for v in values)
            return torch.empty(0, dtype=torch.int64, device=first.device).set_(st, first.storage_offset(), (total,))
    return torch.cat(list(values))


def _resolve_arrangement(arrangement: Optional[str], switchable: bool) -> Optional[str]:
    """None -> the library default for a window object that can do both (CE_ARRANGEMENT, 'auto' unless set)"""
    if arrangement is None:
        arrangement = DEFAULT_ARRANGEMENT if switchable else None
    if arrangement is not None and arrangement not in ("auto",) + ARRANGEMENTS:
        raise ValueError(f"arrangement={arrangement!r}: 'auto', 'overlap' or 'interleaved'")
    if arrangement is not None and not switchable:
        raise ValueError("arrangement= needs a window built with overlap=True (no cache-op graph)")
    return arrangement


DEFAULT_ARRANGEMENT = __import__("os").environ.get("CE_ARRANGEMENT", "auto")


class PrefetchWindow:
    def __init__(self, embed: CachedEmbeddingBag, prefetch_num: int = 1, overlap: bool = False, cache_cus: int = 0,
                 presort: bool = False, transport: Optional[str] = "auto", bag_layout=None,
                 arrangement: Optional[str] = None, arrangement_trial: Optional[dict] = None):
        # arrangement (overlap=True only): where the submitted window's cache op runs -- 'overlap' on the side stream
        # beside the current window's steps, 'interleaved' in two halves on the CURRENT stream (submit() enqueues the
        # first, collect() the second: the caller's training steps lie in between), 'auto' (the default): both are
        # measured while training and the faster one is kept (ArrangementTrial; arrangement_trial = its keyword
        # arguments).  The slots, the keys and the cache state are the same either way.
        # bag_layout = (offsets, include_last_offset, hook_features) of the batches (presort=True, mode='sum' without
        # per-sample weights): the window's keys become source-row keys (functional.SrcKeys), which the backward
        # streams over without a per-tile bag search
        self._layout = None if bag_layout is None else dict(offsets=bag_layout[0],
                                                            include_last_offset=bool(bag_layout[1]),
                                                            hook_features=int(bag_layout[2]),
                                                            identity_bags=bool(bag_layout[1]) and
                                                            is_identity_layout(bag_layout[0], bool(bag_layout[1])))
        # transport (overlap=True only): how rows move while the cache op runs beside training; "worker" keeps the
        # swap traffic off the CUs (CachedParamMgr.set_transport), "auto" decides by the size of the first window
        # (pick_transport), None leaves the manager's setting alone
        assert prefetch_num >= 1
        # presort=True: the cache op also sorts the window's slots in 16384-lookup segments (ce_bag_presort), so the
        # fused backward neither sorts nor issues as many row updates; the keys of the last prepared/collected window
        # are in `self.keys`
        self.presort = presort
        self.keys: Optional[List[torch.Tensor]] = None
        self.embed = embed
        self.mgr = embed.cache_weight_mgr
        self.P = prefetch_num
        self.overlap = overlap
        self._side: Optional[torch.cuda.Stream] = None
        self._pending = None   # (event, [slots per batch])
        if overlap:
            self._side = make_side_stream(self.mgr.device, cache_cus)
            self.mgr.set_protect_depth(1)
            self.mgr.strict = False   # no host sync inside the pipelined cache op
            if transport and transport != "auto":
                self.mgr.set_transport(transport)
            # (worker transport: the cache op's stream does not wait for the missed rows, collect() makes the
            # training stream wait for them -- GraphedWindow._wait_rows)
            self.mgr.set_deferred_rows(True)
        self._auto = overlap and transport == "auto"
        self._ticket_tmp = 0
        self._begin_stream = None
        self.trial: Optional[ArrangementTrial] = None
        self._mode = "overlap" if overlap else "sequential"
        # arrangement=None: the side stream ("overlap").  This window trains EAGER steps between its two calls: with the
        # two halves of the cache op on the training stream of an eager trainer (chained admission), every dispatch of
        # that stream cost ~40 us more from then on -- a DLRM iteration of ~60 kernels 4.5 -> 7.4 ms, in the windows
        # trained with the side stream afterwards as well, so a trial cannot even measure its way out of it
        # (profiles/r06_dlrm_interleaved_dispatch.md; not seen where the steps are hipGraph replays: GraphedWindow keeps
        # "auto").  "auto" / "interleaved" remain for callers who ask.
        if arrangement is None and overlap:
            arrangement = "overlap"
        arrangement = _resolve_arrangement(arrangement, overlap) if (overlap or arrangement is not None) else None
        if arrangement == "auto":
            self.trial = ArrangementTrial(prefetch_num, **(arrangement_trial or {}))
            self._mode = self.trial.mode
        elif arrangement is not None:
            self._mode = arrangement

    @property
    def arrangement(self) -> str:
        return self._mode

    def set_arrangement(self, mode: str) -> None:
        """'overlap' / 'interleaved' for the windows submitted from now on (between collect() and submit())"""
        assert mode in ARRANGEMENTS and self.overlap and self._pending is None
        self._mode = mode

    @torch.no_grad()
    def _cache_op(self, values: Sequence[torch.Tensor], begin_only: bool = False) -> Optional[List[torch.Tensor]]:
        """begin_only: enqueue the first half only (ce_cache_prepare_ids_begin) when the window has the shape that call
        takes -- equal int64 batches, fused keys or none -- and return None if it does not (the caller then takes the
        side stream for this window)"""
        counts = [int(v.numel()) for v in values]
        cat = cat_window(values)
        if self._auto:
            self._auto = False
            self.mgr.set_transport(pick_transport("auto", int(cat.numel())))
        lay = self._layout or {}
        equal = len(set(counts)) == 1 and counts[0] > 0 and cat.dim() == 1 and cat.dtype == torch.int64
        fused = self.presort and FUSED_WINDOW_KEYS and equal and not (lay and EXCLUSIVE_ROWS)
        if begin_only and not (equal and (fused or not self.presort)):
            return None
        with phase("prefetch cache"):                      # the reference's range name (recsys/dlrm_main.py:258)
            if fused or begin_only:       # slots and the window's keys out of one call
                P_, n_ = len(counts), counts[0]
                slots = torch.empty(P_ * n_, dtype=torch.int64, device=cat.device)
                kbuf = torch.empty(P_, presort_len(n_), dtype=torch.int64, device=cat.device) if fused else None
                self.mgr.prepare_ids_keys(cat.view(P_, n_), slots, kbuf, **(lay if fused else {}), _begin_only=begin_only,
                                          defer_rows=self.overlap)
            else:
                slots = self.mgr.prepare_ids(cat, defer_rows=self.overlap)
            self._ticket_tmp = self.mgr.rows_ticket() if self.overlap else 0
        # split by per-batch id counts (torch.chunk in the reference is only right for equal sizes, B#13)
        parts = list(torch.split(slots, counts))
        self._keys_tmp = None
        if fused:
            if lay:
                per = lay["offsets"].shape[-1]
                nb_ = per - 1 if lay["include_last_offset"] else per
                self._keys_tmp = [SrcKeys(kbuf[i], nb_, lay["include_last_offset"], lay["hook_features"], None,
                                          bool(lay["identity_bags"])) for i in range(P_)]
            else:
                self._keys_tmp = [kbuf[i] for i in range(P_)]
        elif self.presort:
            C = self.mgr.cuda_row_num
            # source-row keys: the ids go along, so rows owned by one lane group get plain read-modify-writes
            def with_ids(t, rows):
                return dict(ids=t.reshape(-1).long().contiguous().view(rows, -1)) if (lay and EXCLUSIVE_ROWS) else {}
            if len(set(counts)) == 1:                           # equal batches: one launch for the window
                keys = presort_window(slots.view(len(counts), counts[0]), C, **lay, **with_ids(cat, len(counts)))
                self._keys_tmp = [keys[i] for i in range(len(counts))]
            else:
                self._keys_tmp = [presort_window(p_.view(1, -1), C, **lay, **with_ids(v_, 1))[0]
                                  for p_, v_ in zip(parts, values)]
        return parts

    def prepare(self, values: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """Synchronous window op on the current stream (reference behaviour)."""
        assert 1 <= len(values) <= self.P
        slots = self._cache_op(values)
        if self._ticket_tmp:
            self.mgr.wait_rows(self._ticket_tmp)
        self.keys = self._keys_tmp
        return slots

    def submit(self, values: Sequence[torch.Tensor]) -> None:
        """Start the cache op for the NEXT window: on the side stream (arrangement 'overlap'), or its first half on the
        current stream ('interleaved'; collect() enqueues the second half behind whatever the caller trains meanwhile)."""
        assert self.overlap and self._pending is None
        cur = torch.cuda.current_stream(self.mgr.device)
        if self._mode == "interleaved":
            slots = self._cache_op(values, begin_only=True)
            if slots is not None:
                self._begin_stream = cur             # (the second half goes to the SAME stream: ADVICE r5)
                self._pending = (None, slots, self._keys_tmp, self._ticket_tmp)
                return
        self._side.wait_stream(cur)          # ids were produced on the current stream
        with torch.cuda.stream(self._side):
            slots = self._cache_op(values)
            keys = self._keys_tmp
            ev = torch.cuda.Event()
            ev.record(self._side)
        for v in values:
            v.record_stream(self._side)
        self._pending = (ev, slots, keys, self._ticket_tmp)

    def collect(self) -> List[torch.Tensor]:
        """Slots of the submitted window; the current stream waits for the side stream."""
        assert self._pending is not None
        ev, slots, keys, ticket = self._pending
        self._pending = None
        # strict=False: a cache op that overflowed (unique(window k U k+1) > cuda_row_num) or met a bad id handed
        # back slots of -1; raise like the reference as soon as its record has arrived (no host wait)
        self.mgr.raise_on_failed_calls()
        cur = torch.cuda.current_stream(self.mgr.device)
        if ev is None:                       # interleaved: the second half, on the stream the first one went to
            bs = self._begin_stream
            if bs is not None and bs != cur:     # the caller changed streams between submit() and collect()
                bs.wait_stream(cur)
                with torch.cuda.stream(bs):
                    self.mgr.prepare_ids_finish(defer_rows=True)
                cur.wait_stream(bs)
            else:
                self.mgr.prepare_ids_finish(defer_rows=True)
        else:
            cur.wait_event(ev)
            for s in slots:
                s.record_stream(cur)
            for k in keys or []:
                (k.keys if isinstance(k, SrcKeys) else k).record_stream(cur)
                if isinstance(k, SrcKeys) and k.ranges is not None:
                    k.ranges.record_stream(cur)
        if ticket:
            self.mgr.wait_rows(ticket)       # the missed rows of this window, before anything reads its slots
        self.keys = keys
        if self.trial is not None:           # a window boundary on the training stream: the trial's clock
            self._mode = self.trial.window_done(cur)
        return slots


class GraphedWindow:
    """The prefetch window with the P training steps of a window captured in a hipGraph.

    Every kernel of the operator is capture-safe (no host sync, no allocation inside libce_hip), so the
    per-step Python/launch cost (~0.18 ms for one forward + backward through autograd, the same order as
    the GPU time of the step) collapses into one graph launch per window.  Two slot buffers + two graphs
    alternate so the side-stream cache op of window k+1 can fill its buffer while graph k replays.

    step_fn(slots_i, i) runs one training step on batch i of the window; it is recorded once per buffer.
    All tensors it reads besides `slots_i` must be static (offsets, upstream gradient / dense inputs).

    graph_cache_op (overlap=True, zero-copy transport): the cache op + presort of a window are captured too, as a
    graph of their own that run_and_submit replays on the side stream -- a window then costs the host one copy of the
    ids into a static buffer and two graph launches instead of ~16 kernel launches (libce_hip counts the call number
    on the device for replayed calls; CachedParamMgr.graph_replayed keeps the host's books).  Measured (DESIGN.md
    section 4, Avazu B = 2048 at prefetch_num = 1): the same 0.18 ms per step as launching kernel by kernel -- the cache op
    there is 14 dependent kernels of 5-30 us, 143 us back to back even inside a graph, i.e. bound by the GPU's
    kernel-to-kernel latency and not by the host -- so it is off unless asked for.  (The cache op as a second BRANCH
    of the training graph was measured too: the HIP graph executor runs the branches one after the other with ~70 us
    between queue switches, 0.66 ms per step.)"""

    def __init__(self, embed: CachedEmbeddingBag, prefetch_num: int, ids_per_batch: int, step_fn, overlap: bool = True,
                 warmup_values: Optional[Sequence[torch.Tensor]] = None, cache_cus: int = 0, presort: bool = False,
                 transport: Optional[str] = "auto", bag_layout=None, graph_cache_op: bool = False,
                 plan_ahead: int = 1, interleaved: bool = False, arrangement: Optional[str] = None,
                 arrangement_trial: Optional[dict] = None):
        # arrangement: 'overlap' | 'interleaved' | 'auto' for a window built with overlap=True, plan_ahead 1 and no
        # cache-op graph (it can do both: set_arrangement).  'auto' -- the DEFAULT for such a window -- measures both while
        # training and keeps the faster (ArrangementTrial; `arrangement_trial` = its keyword arguments; `self.trial`).
        # interleaved (instead of overlap): NO side stream.  The cache op of window k+1 is issued on the training stream
        # in two halves around the steps of window k -- begin(k+1), graph(k), finish(k+1) (ce_cache_prepare_ids_begin /
        # _finish) -- so its kernels never run BESIDE the bag kernels (side by side they cost each other more than the
        # cache op's own kernel time: every one of them is bound by the same memory system), while the PCIe admission of
        # window k+1's rows still overlaps with window k's steps.  Same protection rule as overlap (protect_depth 1).
        assert not (interleaved and (overlap or graph_cache_op or plan_ahead != 1)), \
            "interleaved replaces overlap (one stream, plan_ahead 1)"
        self.interleaved = interleaved
        # both arrangements on one object: built with overlap=True (the side stream exists), set_arrangement() then
        # moves the NEXT window's cache op between the side stream and the two halves on the training stream.
        # plan_ahead 2 (what prefetch_num = 1 pipelines use) can switch too: the caller keeps submitting two windows
        # ahead; 'interleaved' then runs begin(k+2), steps(k), finish(k+2) on the training stream -- the same two halves
        # around one window's steps, the slots one window earlier than needed.  Which one a prefetch_num = 1 pipeline wants
        # depends on the workload (round 5, one box: Kaggle 5 % DATASET 1.55 G interleaved against 1.29 G two windows
        # ahead on the side stream; LFU 1.44 against 1.58 G; Avazu B = 2048 188 against 215 M): the trial decides.
        self.switchable = overlap and plan_ahead in (1, 2) and not graph_cache_op
        arrangement = _resolve_arrangement(arrangement, self.switchable and not interleaved) \
            if (arrangement is not None or (self.switchable and not interleaved)) else None
        self.trial: Optional[ArrangementTrial] = None
        self._begun: Optional[int] = None          # buffer whose cache op has been begun and not finished
        # plan_ahead (overlap=True): how many windows the cache op may run ahead of training.  1: the cache op of window
        # k+1 starts when window k-1 has trained (two slot buffers, protect_depth 1).  2: it starts when window k-2 has
        # -- three buffers, protect_depth 2, unique(three consecutive windows) must fit the cache, and the ids handed to
        # submit() must be COMPLETE when it is called (submit no longer waits for the compute stream, which would put
        # the cache op behind the training already enqueued there).  That takes the wait for the previous window's
        # training out of a chain that is longer than a training step: prefetch_num = 1.
        # bag_layout: see PrefetchWindow (static offsets shared by every batch; keys_i is then a SrcKeys)
        assert plan_ahead in (1, 2) and (plan_ahead == 1 or overlap)
        self.plan_ahead = plan_ahead
        nbuf = self.nbuf = plan_ahead + 1
        self._layout = None if bag_layout is None else dict(offsets=bag_layout[0],
                                                            include_last_offset=bool(bag_layout[1]),
                                                            hook_features=int(bag_layout[2]),
                                                            identity_bags=bool(bag_layout[1]) and
                                                            is_identity_layout(bag_layout[0], bool(bag_layout[1])))
        self.embed = embed
        self.mgr = embed.cache_weight_mgr
        self.P = prefetch_num
        self.n = ids_per_batch
        self.overlap = overlap
        dev = self.mgr.device
        self._bufs = [torch.zeros(self.P, self.n, dtype=torch.int64, device=dev) for _ in range(nbuf)]
        # presort=True: step_fn(slots_i, i, keys_i); the segment-grouped keys of the window (ce_bag_presort_window)
        # are produced behind the cache op into a static buffer next to the slots
        self.presort = presort
        self._klen = presort_len(self.n)
        self._keys = [torch.full((self.P, self._klen), -1, dtype=torch.int64, device=dev) for _ in range(nbuf)] \
            if self.presort else None
        # id range of every 16384-lookup segment (source-row keys only): min > max = "no ids seen" until a presort ran
        self._ranges = None
        if self.presort and self._layout is not None and EXCLUSIVE_ROWS:
            self._ranges = [torch.empty(self.P, self._klen // 16384, 2, dtype=torch.int64, device=dev)
                            for _ in range(nbuf)]
            for r in self._ranges:
                r[..., 0] = torch.iinfo(torch.int64).min       # "everything": never disjoint -> atomics
                r[..., 1] = torch.iinfo(torch.int64).max
        self._side = make_side_stream(dev, cache_cus) if overlap else None
        self._events = [None] * nbuf
        self._tickets = [0] * nbuf             # rows ticket of the cache op that filled buffer b (deferred rows)
        self._read_done = [None] * nbuf        # event behind the last training run that read buffer b
        self._step_fn = step_fn
        if overlap or interleaved:
            self.mgr.set_protect_depth(plan_ahead)
            self.mgr.strict = False
            transport = pick_transport(transport, prefetch_num * ids_per_batch)
            if transport:
                self.mgr.set_transport(transport)
            # worker transport: the missed rows of a window travel on the library's admission stream; the cache op's own
            # stream does not wait for them (the next window's cache op may follow it at once) -- the TRAINING stream
            # does, right before the first step that reads the window's slots (_wait_rows)
            self.mgr.set_deferred_rows(True)
        # eager warm-up on real slots (lazy initialisation must not happen during capture), then capture
        if warmup_values is not None:
            wcat = torch.cat(list(warmup_values))
            self._cache_op(wcat, 0)
            for b in range(1, nbuf):
                self._bufs[b].copy_(self._bufs[0])
            if self.presort:
                for b in range(1, nbuf):
                    self._keys[b].copy_(self._keys[0])
                    if self._ranges is not None:
                        self._ranges[b].copy_(self._ranges[0])
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            self._wait_rows(0)
            for i in range(self.P):
                self._call(step_fn, 0, i)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self._graphs = []
        self._step_graphs = []          # [buf][i]: batch i alone (windows a caller trains in part)
        for b in range(nbuf):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(self.P):
                    self._call(step_fn, b, i)
            self._graphs.append(g)
            per_step = []
            for i in range(self.P if self.P > 1 else 0):
                gi = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gi):
                    self._call(step_fn, b, i)
                per_step.append(gi)
            self._step_graphs.append(per_step)
        self._plan_graphs = None
        self._ids = None
        if graph_cache_op:
            if not overlap or self.mgr.transport_name != "zerocopy" or self._ranges is not None or plan_ahead != 1:
                raise ValueError("graph_cache_op needs overlap=True, plan_ahead=1 and the zero-copy transport")
            self._ids = [torch.zeros(self.P, self.n, dtype=torch.int64, device=dev) for _ in range(2)]
            if warmup_values is not None:
                for t in self._ids:
                    t.view(-1).copy_(wcat.reshape(-1))
            self._capture_plans()
        if arrangement == "auto":
            self.trial = ArrangementTrial(self.P, **(arrangement_trial or {}))
            self.set_arrangement(self.trial.mode)
        elif arrangement is not None:
            self.set_arrangement(arrangement)

    def _trial_tick(self, whole_window: bool) -> None:
        if self.trial is None:
            return
        if not whole_window:
            self.trial.reset_block()
            return
        mode = self.trial.window_done(torch.cuda.current_stream(self.mgr.device))
        if mode != self.arrangement:
            self.set_arrangement(mode)

    def settle_arrangement(self, wait: bool = True) -> Optional[str]:
        """the trial's verdict so far (None while blocks are still being trained; wait=True synchronises on the last
        block's events once all have been enqueued) -- for a caller that wants its own measurement to start after it"""
        if self.trial is None:
            return self.arrangement
        mode = self.trial.poll(wait=wait)
        if mode is not None and mode != self.arrangement and not self.trial.running:
            self.set_arrangement(mode)
        return mode

    def _capture_plans(self) -> None:
        # the cache op + presort that fill buffer b as a graph of their own, replayed on the side stream
        torch.cuda.synchronize(self.mgr.device)
        plans = []
        for b in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._cache_op(self._ids[b].view(-1), b)
            plans.append(g)
        self._plan_graphs = plans

    def _call(self, step_fn, buf: int, i: int) -> None:
        if self.presort and self._layout is not None:
            lay = self._layout
            per = lay["offsets"].shape[-1]
            step_fn(self._bufs[buf][i], i, SrcKeys(self._keys[buf][i], per - 1 if lay["include_last_offset"] else per,
                                                   lay["include_last_offset"], lay["hook_features"],
                                                   self._ranges[buf][i] if self._ranges is not None else None,
                                                   bool(lay["identity_bags"])))
        elif self.presort:
            step_fn(self._bufs[buf][i], i, self._keys[buf][i])
        else:
            step_fn(self._bufs[buf][i], i)

    def drain(self) -> None:
        """interleaved: enqueue the second half of a cache op that was begun and not finished yet (before anything else
        uses the cache manager)"""
        self._finish_begun()

    @property
    def arrangement(self) -> str:
        return "interleaved" if self.interleaved else ("overlap" if self.overlap else "sequential")

    def set_arrangement(self, mode: str) -> None:
        """'overlap' (the next window's cache op on the side stream, beside this window's steps) or 'interleaved' (its
        two halves on the training stream, around this window's steps) for the cache ops submitted FROM NOW ON.  Which
        one is faster depends on what the host's side of the row swap does to the kernels that run beside it (DESIGN.md
        section 4): the bench measures both after its warm-up and keeps the faster.  Only between windows: a cache op
        that was begun must have been finished (run() does that)."""
        assert mode in ("overlap", "interleaved")
        if mode == self.arrangement:
            return
        if not self.switchable:
            raise ValueError("this window was not built with overlap=True, plan_ahead=1 and no cache-op graph")
        self._finish_begun()
        self.interleaved = mode == "interleaved"
        self.overlap = not self.interleaved
        if self.overlap and self.plan_ahead > 1:
            # two windows ahead, submit() does not make the side stream wait for the training stream -- but the cache ops
            # issued there while the arrangement was 'interleaved' come first (the cache manager's calls are ordered)
            self._side.wait_stream(torch.cuda.current_stream(self.mgr.device))

    def _finish_begun(self) -> None:
        if self._begun is not None:
            self.mgr.prepare_ids_finish(defer_rows=True)
            self._begun = None

    def _cache_op(self, cat: torch.Tensor, buf: int, begin_only: bool = False) -> None:
        """cache op of a window into slot buffer `buf`, and its keys when the window is presorted"""
        if begin_only and self._ranges is None and cat.dtype == torch.int64 and cat.is_contiguous():
            self.mgr.prepare_ids_begin(cat.view(self.P, self.n), self._bufs[buf],
                                       self._keys[buf] if self.presort else None,
                                       **((self._layout or {}) if self.presort else {}))
            self._begun = buf
            self._tickets[buf] = self.mgr.rows_ticket()
            return
        if self.presort and FUSED_WINDOW_KEYS and self._ranges is None and cat.dtype == torch.int64 and cat.is_contiguous():
            self.mgr.prepare_ids_keys(cat.view(self.P, self.n), self._bufs[buf], self._keys[buf], defer_rows=True,
                                      **(self._layout or {}))
            self._tickets[buf] = self.mgr.rows_ticket()
            return
        self.mgr.prepare_ids(cat, out=self._bufs[buf], defer_rows=True)
        self._tickets[buf] = self.mgr.rows_ticket()
        if self.presort:
            self._presort(buf, cat)

    def _wait_rows(self, buf: int) -> None:
        """the current (training) stream waits for the rows the cache op of buffer `buf` missed"""
        if self._tickets[buf]:
            self.mgr.wait_rows(self._tickets[buf])
            self._tickets[buf] = 0

    def _presort(self, buf: int, ids: torch.Tensor) -> None:
        # one launch for the window: every batch's 16384-lookup segments grouped by row
        if self._ranges is not None:
            presort_window(self._bufs[buf], self.mgr.cuda_row_num, keys_out=self._keys[buf], **self._layout,
                           ids=ids.reshape(self.P, self.n), ranges_out=self._ranges[buf])
        else:
            presort_window(self._bufs[buf], self.mgr.cuda_row_num, keys_out=self._keys[buf], **(self._layout or {}))

    @torch.no_grad()
    def submit(self, values: Sequence[torch.Tensor], buf: int) -> None:
        """Cache op of a window into slot buffer `buf` (0/1), on the side stream when overlap=True.
        Call it BEFORE run() of the previous window so the two overlap: the side stream only waits for the
        work already enqueued on the compute stream (the graph that last read `buf`)."""
        if self.overlap and self.plan_ahead > 1:
            # the ids are complete (the caller's promise): only the last training run that read this buffer is waited for
            with torch.cuda.stream(self._side), phase("prefetch cache"):
                if self._read_done[buf] is not None:
                    self._side.wait_event(self._read_done[buf])
                cat = cat_window(values)
                assert cat.numel() == self.P * self.n
                self._cache_op(cat, buf)
                ev = torch.cuda.Event()
                ev.record(self._side)
            for v in values:
                v.record_stream(self._side)
            self._events[buf] = ev
            return
        cat = cat_window(values)
        assert cat.numel() == self.P * self.n
        if self.interleaved:
            with phase("prefetch cache"):
                self._finish_begun()                 # (a window begun and never trained: first window, or a caller's skip)
                if self._side is not None:
                    # (switchable: a cache op still running on the side stream comes first -- its event stays for run())
                    cur = torch.cuda.current_stream(self.mgr.device)
                    for ev in self._events:
                        if ev is not None:
                            cur.wait_event(ev)
                self._cache_op(cat, buf, begin_only=True)
            self._events[buf] = None
            return
        if self.overlap:
            cur = torch.cuda.current_stream(self.mgr.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side), phase("prefetch cache"):
                self._cache_op(cat, buf)
                ev = torch.cuda.Event()
                ev.record(self._side)
            cat.record_stream(self._side)
            self._events[buf] = ev
        else:
            self._cache_op(cat, buf)
            self._events[buf] = None

    def run_steps(self, buf: int, first: int, last: int) -> None:
        """Batches [first, last) of the window in buffer `buf`, one single-step graph each (a window that a caller
        only trains in part, or across two timed regions)."""
        if self._begun == buf:
            self._finish_begun()
        if self._events[buf] is not None:
            torch.cuda.current_stream(self.mgr.device).wait_event(self._events[buf])
            self._events[buf] = None
        self._wait_rows(buf)
        for i in range(first, last):
            if self._step_graphs[buf]:
                self._step_graphs[buf][i].replay()
            else:
                self._call(self._step_fn, buf, i)
        if last >= self.P:
            self._finish_begun()             # interleaved: the next window's second half, behind this window's last step
        if self.plan_ahead > 1:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.mgr.device))
            self._read_done[buf] = ev
        self._trial_tick(False)

    @torch.no_grad()
    def run_and_submit(self, buf: int, next_values: Sequence[torch.Tensor]) -> None:
        """submit(next_values, 1 - buf) + run(buf): the window in buffer `buf` trains while the cache op of the next
        window fills the other buffer; with graph_cache_op both are graph replays."""
        assert self.nbuf == 2, "run_and_submit alternates two buffers (plan_ahead = 1)"
        if self._plan_graphs is None:
            self.submit(next_values, 1 - buf)
            self.run(buf)
            return
        cur = torch.cuda.current_stream(self.mgr.device)
        if self._events[buf] is not None:
            cur.wait_event(self._events[buf])
            self._events[buf] = None
        self._wait_rows(buf)
        self.mgr.raise_on_failed_calls()
        dst = self._ids[1 - buf].view(-1)
        if len(next_values) == 1:
            dst.copy_(next_values[0].reshape(-1))
        else:
            torch.cat([v.reshape(-1) for v in next_values], out=dst)
        self._side.wait_stream(cur)
        recapture = False
        with torch.cuda.stream(self._side):
            self._plan_graphs[1 - buf].replay()
            try:
                self.mgr.graph_replayed(1, self.P * self.n)
            except _lib.CeError as e:
                if e.code != _lib.CE_ERR_UNSUPPORTED:
                    raise
                recapture = True                  # LFU: the captured key width is used up (every 2^31 ids)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._graphs[buf].replay()
        self._events[1 - buf] = ev
        if recapture:
            self._capture_plans()

    def run(self, buf: int, steps: Optional[int] = None) -> None:
        """Replay the P training steps on the slots in buffer `buf` (waits for its cache op).  steps < P runs
        only the first `steps` batches, eagerly (a trailing partial window)."""
        if self._begun == buf:                   # (no window trained in between: the two halves back to back)
            self._finish_begun()
        if self._events[buf] is not None:
            torch.cuda.current_stream(self.mgr.device).wait_event(self._events[buf])
            self._events[buf] = None
        self._wait_rows(buf)
        if not self.mgr.strict:
            self.mgr.raise_on_failed_calls()     # non-blocking; see PrefetchWindow.collect
        whole = steps is None or steps >= self.P
        if whole:
            self._graphs[buf].replay()
        else:
            self.run_steps(buf, 0, steps)
        self._finish_begun()                     # interleaved: the NEXT window's second half, behind this window's steps
        if self.plan_ahead > 1:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.mgr.device))
            self._read_done[buf] = ev
        if whole:
            self._trial_tick(True)
