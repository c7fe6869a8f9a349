"""What ANY correct restatement of upstream's CachedParamMgr must satisfy, whatever its data structures -- checked on both
CPU oracles over random streams (hypothesis).  None of these properties is read off the oracles' own code: they follow
from the manager's contract as the reference uses it (recsys/dlrm_main.py:259 `prepare_ids` -> slots that address
`cuda_cached_weight`; README "frequency-aware ... evicts the least frequently used / the lowest-ranked rows"; SURVEY.md
Appendix A.3-A.6, B#1-2 for the tie order):

  maps      slot <-> row is a bijection on the resident rows, the free-slot count is what the maps say
  service   every id of the call is resident afterwards, its slot holds ITS row; so are the rows of the previous
            `protect_depth` calls (the build's extension)
  victims   exactly max(0, misses - free slots) rows leave, none of them protected, and they are the EXTREME ones of
            the eligible rows: DATASET the largest row indices (= the least frequent by the re-rank), LFU the smallest
            (counter, slot) pairs -- stated on the pre-call state, not on how a restatement searches for them
  admission the missed rows, ascending, take the lowest free slots, ascending
  payload   a row's value is never lost or duplicated: through every eviction / admission / flush the table ends as if
            plain rows had been updated in place
  counters  hits + misses = distinct rows of the call; an LFU counter grows by the row's multiplicity, a fresh row's
            counter is its multiplicity in the admitting call"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle.cache_oracle import DATASET, LFU, OracleCachedParamMgr  # noqa: E402
from oracle.cache_oracle_torch import TorchCachedParamMgr  # noqa: E402


class _View:
    """the manager's state as numpy arrays, whichever oracle it is"""

    def __init__(self, mgr):
        self.m = mgr
        self.torch = isinstance(mgr, TorchCachedParamMgr)

    def _np(self, x):
        return x.numpy() if self.torch else x

    cached_idx_map = property(lambda s: s._np(s.m.cached_idx_map))
    inverted = property(lambda s: s._np(s.m.inverted_cached_idx))
    idx_map = property(lambda s: s._np(s.m.idx_map))
    cache = property(lambda s: s._np(s.m.cuda_cached_weight))
    table = property(lambda s: s._np(s.m.weight))
    freq = property(lambda s: None if s.m.freq_cnter is None else s._np(s.m.freq_cnter))
    avail = property(lambda s: s.m._cuda_available_row_num if s.torch else s.m.cuda_available_row_num)

    def prepare_ids(self, ids):
        out = self.m.prepare_ids(torch.from_numpy(ids) if self.torch else ids)
        return out.numpy() if self.torch else out


def _make(kind, w0, C, strategy):
    if kind == "numpy":
        return OracleCachedParamMgr(w0.copy(), C, strategy)
    return TorchCachedParamMgr(torch.from_numpy(w0.copy()), C, "lfu" if strategy == LFU else "dataset")


def _check_maps(v, C):
    cim, inv = v.cached_idx_map, v.inverted
    occ = np.nonzero(cim >= 0)[0]
    rows = cim[occ]
    assert len(np.unique(rows)) == len(rows), "a row sits in two slots"
    assert np.array_equal(inv[rows], occ), "inverted map disagrees with the slot map"
    assert int((inv >= 0).sum()) == len(occ), "a row is marked resident without a slot"
    assert v.avail == C - len(occ), "free-slot count disagrees with the maps"
    assert np.all(cim[cim < 0] == -1)


@settings(max_examples=120, deadline=None)
@given(kind=st.sampled_from(["numpy", "torch"]), strategy=st.sampled_from([DATASET, LFU]), depth=st.integers(0, 2),
       use_freq=st.booleans(), warm=st.sampled_from([0.0, 0.3, 0.7, 1.0]), N=st.integers(20, 400),
       c_frac=st.floats(0.05, 1.0), skew=st.floats(0.3, 3.0), seed=st.integers(0, 2 ** 31 - 1))
def test_manager_contract_on_random_streams(kind, strategy, depth, use_freq, warm, N, c_frac, skew, seed):
    rng = np.random.default_rng(seed)
    D = 3
    C = max(depth + 3, int(N * c_frac))
    w0 = rng.standard_normal((N, D)).astype(np.float32)
    mgr = _make(kind, w0, C, strategy)
    mgr.protect_depth = depth
    mgr.reorder(rng.integers(0, 6, size=N) if use_freq else None, warm)
    v = _View(mgr)
    _check_maps(v, C)
    truth = w0.copy()                                   # what every ROW holds, wherever it lives
    occ = np.nonzero(v.cached_idx_map >= 0)[0]
    assert np.array_equal(v.cache[occ], truth[v.cached_idx_map[occ]])      # the warm-up loaded the rows it says it did
    history = []                                        # distinct rows of the previous calls
    u_max = max(1, C // (depth + 2))                    # so that enough unprotected rows can always leave
    for call in range(12):
        n = int(rng.integers(1, 4 * u_max + 1))
        pool = rng.choice(N, size=min(N, int(rng.integers(1, u_max + 1))), replace=False)
        ids = pool[np.minimum((rng.pareto(skew, size=n)).astype(np.int64), len(pool) - 1)].astype(np.int64)
        rows_all = v.idx_map[ids]
        rows, cnt = np.unique(rows_all, return_counts=True)
        # ---- the state before the call
        cim0, inv0, avail0 = v.cached_idx_map.copy(), v.inverted.copy(), v.avail
        freq0 = None if v.freq is None else v.freq.copy()
        protected = np.unique(np.concatenate([rows] + history[-depth:])) if depth else rows
        miss = rows[inv0[rows] < 0]
        k = max(0, len(miss) - avail0)
        slots = v.prepare_ids(ids)
        # ---- maps and service
        _check_maps(v, C)
        assert slots.shape == ids.shape and np.all((slots >= 0) & (slots < C))
        assert np.array_equal(v.cached_idx_map[slots], rows_all), "a slot does not hold the row of its id"
        assert np.all(v.inverted[protected] >= 0), "a row of this call or of a protected earlier call is not resident"
        # ---- victims: how many, which
        left = np.nonzero((inv0 >= 0) & (v.inverted < 0))[0]              # rows resident before, not after
        assert len(left) == k, f"{len(left)} rows left the cache, {k} had to"
        assert len(np.intersect1d(left, protected)) == 0, "a protected row was evicted"
        eligible = np.setdiff1d(np.nonzero(inv0 >= 0)[0], protected)
        stay = np.setdiff1d(eligible, left)
        if k and len(stay):
            if strategy == DATASET:
                assert left.min() > stay.max(), "DATASET must evict the highest-ranked-last rows (largest row index)"
            else:
                key = lambda r: (freq0[inv0[r]].astype(object), inv0[r])
                worst_gone = max(zip(*key(left)))
                best_kept = min(zip(*key(stay)))
                assert worst_gone < best_kept, "LFU must evict the smallest (counter, slot) pairs"
        # ---- admission: ascending rows into ascending free slots
        cim_after_evict = cim0.copy()
        cim_after_evict[inv0[left]] = -1
        free = np.nonzero(cim_after_evict == -1)[0][:len(miss)]
        assert np.array_equal(v.inverted[miss], free), "missed rows must take the lowest free slots in ascending order"
        # ---- payload: every resident row of the call holds the row's current value
        assert np.array_equal(v.cache[v.inverted[rows]], truth[rows])
        assert np.array_equal(v.table[left], truth[left]), "an evicted row did not reach the host table"
        # ---- counters
        assert mgr.num_hits_history[-1] + mgr.num_miss_history[-1] == len(rows)
        assert mgr.num_miss_history[-1] == len(miss) and mgr.num_write_back_history[-1] == k
        if strategy == LFU:
            was = np.where(inv0[rows] >= 0, freq0[np.maximum(inv0[rows], 0)], 0)
            assert np.array_equal(v.freq[v.inverted[rows]], was + cnt)
        # ---- "train": the rows of the call change in the cache, and only there
        delta = rng.standard_normal((len(rows), D)).astype(np.float32)
        truth[rows] += delta
        if v.torch:
            mgr.cuda_cached_weight[torch.from_numpy(v.inverted[rows])] += torch.from_numpy(delta)
        else:
            mgr.cuda_cached_weight[v.inverted[rows]] += delta
        history.append(rows)
    mgr.flush()
    _check_maps(v, C)
    assert v.avail == C
    assert np.array_equal(v.table, truth), "the table after flush() is not the table plain rows would have given"


def test_a_wrong_victim_rule_is_caught():
    """the contract test is not vacuous: an LRU-ish manager (evicts the LOWEST row indices) fails it"""
    class Wrong(OracleCachedParamMgr):
        def _find_evict_gpu_idxs(self, k, protected_rows):
            ok = (self.cached_idx_map >= 0) & ~np.isin(self.cached_idx_map, protected_rows)
            cand = np.nonzero(ok)[0]
            return cand[np.argsort(self.cached_idx_map[cand])[:k]].astype(np.int64)
    rng = np.random.default_rng(0)
    N, C = 200, 20
    mgr = Wrong(rng.standard_normal((N, 2)).astype(np.float32), C, DATASET)
    mgr.reorder(None, 1.0)
    inv0 = mgr.inverted_cached_idx.copy()
    ids = np.arange(100, 110)
    mgr.prepare_ids(ids)
    left = np.nonzero((inv0 >= 0) & (mgr.inverted_cached_idx < 0))[0]
    stay = np.setdiff1d(np.nonzero(inv0 >= 0)[0], left)
    assert len(left) == 10 and not (left.min() > stay.max())
