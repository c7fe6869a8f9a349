"""CPU tests of the oracle itself: upstream's LFU known-answer, the cached==uncached
invariant, the idx_map rank property, and the committed golden vectors."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import bag_oracle
from oracle.cache_oracle import DATASET, LFU, OracleCachedParamMgr, id_freq_map, power_law_ids

GOLD = Path(__file__).resolve().parent / "golden"

# the 22 scripted lookups of upstream ColossalAI tests/test_layers/test_cache_embedding.py::test_lfu_strategy
LFU_SCRIPT = [[2], [1, 2], [0, 2], [0, 1, 2], [0, 1, 2], [0, 1, 2], [0, 1, 2], [0, 2], [0, 2], [0, 2], [0, 2],
              [0], [0], [0], [0], [0, 1, 2], [0, 1, 2], [3], [2], [4], [2], [0]]


@pytest.mark.parametrize("init_freq", [False, True])
def test_lfu_known_answer(init_freq):
    w = np.random.default_rng(0).standard_normal((5, 5)).astype(np.float32)
    mgr = OracleCachedParamMgr(w, 3, LFU)
    mgr.reorder([4, 2, 1, 3, 1] if init_freq else None, warmup_ratio=1.0)
    for ids in LFU_SCRIPT:
        mgr.prepare_ids(np.array(ids))
    assert mgr.num_hits_history[-6:] == [3, 0, 1, 0, 1, 1]
    assert set(mgr.traces[-5].evicted_rows.tolist()) == {1}      # "[3]: miss, evict 1"
    assert set(mgr.traces[-3].evicted_rows.tolist()) == {3}      # "[4]: miss, evict 3"


def test_reorder_rank_map():
    rng = np.random.default_rng(3)
    freq = rng.integers(0, 20, size=500)
    mgr = OracleCachedParamMgr(np.zeros((500, 4), np.float32), 50, DATASET)
    mgr.reorder(freq, 0.7)
    order = np.argsort(-freq, kind="stable")
    assert np.array_equal(mgr.idx_map[order], np.arange(500))
    # rank by descending frequency, ties by ascending id
    for a, b in zip(order[:-1], order[1:]):
        assert freq[a] > freq[b] or (freq[a] == freq[b] and a < b)
    # warm-up: the 35 hottest rows sit in slots 0..34
    assert np.array_equal(mgr.cached_idx_map[:35], np.arange(35))
    assert mgr.cuda_available_row_num == 15


@pytest.mark.parametrize("strategy", [DATASET, LFU])
@pytest.mark.parametrize("mode", ["sum", "mean"])
def test_equivalence_invariant(strategy, mode):
    """cached EmbeddingBag == plain EmbeddingBag for outputs, and host weights after flush
    (the contract of upstream's test_cache_embedding.py)."""
    rng = np.random.default_rng(11)
    N, D, C, lr = 300, 16, 40, 0.1
    w0 = rng.standard_normal((N, D)).astype(np.float32)
    freq = rng.integers(0, 9, size=N)
    mgr = OracleCachedParamMgr(w0.copy(), C, strategy)
    mgr.reorder(freq, 0.7)
    ref = torch.from_numpy(w0.copy())
    # DATASET re-rank permutes logical rows (SURVEY B#3): id x lives at row idx_map[x]
    id2row = mgr.idx_map
    for step in range(6):
        nb = 12
        lens = rng.integers(0, 4, size=nb)
        offs = np.concatenate([[0], np.cumsum(lens)])
        ids = rng.integers(0, N, size=int(offs[-1]))
        go = torch.from_numpy(rng.standard_normal((nb, D)).astype(np.float32))
        slots = mgr.prepare_ids(ids)
        cw = torch.from_numpy(mgr.cuda_cached_weight)
        out_c = bag_oracle.bag_forward(cw, torch.from_numpy(slots), torch.from_numpy(offs), None, mode)
        out_r = bag_oracle.bag_forward(ref, torch.from_numpy(id2row[ids]), torch.from_numpy(offs), None, mode)
        torch.testing.assert_close(out_c, out_r, rtol=1e-6, atol=1e-6)
        mgr.cuda_cached_weight[:] = bag_oracle.sgd_step(cw, torch.from_numpy(slots), torch.from_numpy(offs), go, lr,
                                                        None, mode).numpy()
        ref = bag_oracle.sgd_step(ref, torch.from_numpy(id2row[ids]), torch.from_numpy(offs), go, lr, None, mode)
    mgr.flush()
    np.testing.assert_allclose(mgr.weight, ref.numpy(), rtol=1e-5, atol=1e-6)


def test_capacity_overflow_asserts():
    mgr = OracleCachedParamMgr(np.zeros((100, 4), np.float32), 10, DATASET)
    with pytest.raises(AssertionError, match="increase cuda_row_num"):
        mgr.prepare_ids(np.arange(11))
    assert mgr.cuda_available_row_num == 10 and (mgr.cached_idx_map == -1).all()
    with pytest.raises(NotImplementedError):
        OracleCachedParamMgr(np.zeros((10, 4), np.float32), 0, DATASET)


def test_bag_oracle_pinned_by_loops():
    for name, mode, weighted in (("bag_sum", "sum", False), ("bag_sum_weighted", "sum", True),
                                 ("bag_mean", "mean", False)):
        z = np.load(GOLD / f"{name}.npz")
        psw = z["psw"] if weighted else None
        loop = bag_oracle.bag_forward_numpy(z["weight"], z["indices"], z["offsets"], psw, mode)
        np.testing.assert_allclose(loop, z["out"], rtol=1e-6, atol=1e-6)
        t = bag_oracle.bag_forward(torch.from_numpy(z["weight"]), torch.from_numpy(z["indices"]),
                                   torch.from_numpy(z["offsets"]), None if psw is None else torch.from_numpy(psw), mode)
        np.testing.assert_array_equal(t.numpy(), z["out"])


@pytest.mark.parametrize("name,strategy", [("cache_dataset_freq", DATASET), ("cache_dataset_nofreq", DATASET),
                                           ("cache_lfu_freq", LFU), ("cache_lfu_nofreq", LFU)])
def test_cache_golden_replay(name, strategy):
    z = np.load(GOLD / f"{name}.npz")
    N, C, D, n_ids, calls, warm = z["meta"]
    mgr = OracleCachedParamMgr(z["weight"].copy(), int(C), strategy)
    mgr.reorder(z["freq"] if z["freq"].size else None, warm / 1000.0)
    assert np.array_equal(mgr.idx_map, z["idx_map"])
    assert np.array_equal(mgr.cached_idx_map, z["cached_idx_map_0"])
    for c in range(int(calls)):
        slots = mgr.prepare_ids(z["ids"][c])
        mgr.cuda_cached_weight[np.unique(slots)] += np.float32(0.5)
        assert np.array_equal(slots, z["slots"][c])
        assert np.array_equal(mgr.cached_idx_map, z["cached_idx_map"][c])
        ev = z["evicted_rows"][c]
        assert set(mgr.traces[-1].evicted_rows.tolist()) == set(ev[ev >= 0].tolist())
        if strategy == LFU:
            assert np.array_equal(mgr.freq_cnter, z["freq_cnter"][c])
    assert mgr.num_hits_history == z["hits"].tolist() and mgr.num_miss_history == z["misses"].tolist()
    mgr.flush()
    np.testing.assert_array_equal(mgr.weight, z["weight_after_flush"])


def test_protect_depth_extension():
    """protect_depth=1 (overlapped pipeline) never evicts a row of the previous call."""
    rng = np.random.default_rng(5)
    mgr = OracleCachedParamMgr(np.zeros((400, 4), np.float32), 60, LFU)
    mgr.protect_depth = 1
    prev = None
    for _ in range(30):
        ids = rng.integers(0, 400, size=25)
        mgr.prepare_ids(ids)
        if prev is not None:
            assert not set(mgr.traces[-1].evicted_rows.tolist()) & set(np.unique(prev).tolist())
            assert np.all(mgr.inverted_cached_idx[np.unique(prev)] >= 0)
        prev = ids


def test_power_law_generator_shape():
    rng = np.random.default_rng(0)
    ids = power_law_ids(rng, 10_000, 200_000, 0.25)
    assert ids.min() >= 0 and ids.max() < 10_000
    f = id_freq_map(ids, 10_000)
    assert f.sum() == 200_000 and f[0] > f[100] > f[5000]


# ---------------------------------------------------------------------------------------------------------------------
# second oracle: the op-sequence-literal torch restatement (oracle/cache_oracle_torch.py) against the set-based one


def _same_state(a, t, lfu):
    assert np.array_equal(a.cached_idx_map, t.cached_idx_map.numpy())
    assert np.array_equal(a.inverted_cached_idx, t.inverted_cached_idx.numpy())
    assert a.cuda_available_row_num == t._cuda_available_row_num
    if lfu:
        assert np.array_equal(a.freq_cnter, t.freq_cnter.numpy())
    np.testing.assert_array_equal(a.cuda_cached_weight, t.cuda_cached_weight.numpy())
    np.testing.assert_array_equal(a.weight, t.weight.numpy())


@pytest.mark.parametrize("name,strategy", [("cache_dataset_freq", DATASET), ("cache_dataset_nofreq", DATASET),
                                           ("cache_lfu_freq", LFU), ("cache_lfu_nofreq", LFU)])
def test_torch_literal_oracle_replays_golden_streams(name, strategy):
    from oracle.cache_oracle_torch import TorchCachedParamMgr
    z = np.load(GOLD / f"{name}.npz")
    N, C, D, n_ids, calls, warm = z["meta"]
    t = TorchCachedParamMgr(torch.from_numpy(z["weight"].copy()), int(C), strategy)
    t.reorder(z["freq"] if z["freq"].size else None, warm / 1000.0)
    assert np.array_equal(t.idx_map.numpy(), z["idx_map"])
    assert np.array_equal(t.cached_idx_map.numpy(), z["cached_idx_map_0"])
    for c in range(int(calls)):
        slots = t.prepare_ids(torch.from_numpy(z["ids"][c]))
        t.cuda_cached_weight[torch.unique(slots)] += 0.5
        assert np.array_equal(slots.numpy(), z["slots"][c])
        assert np.array_equal(t.cached_idx_map.numpy(), z["cached_idx_map"][c])
        ev = z["evicted_rows"][c]
        assert set(t.last_evicted_rows.tolist()) == set(ev[ev >= 0].tolist())
        if strategy == LFU:
            assert np.array_equal(t.freq_cnter.numpy(), z["freq_cnter"][c])
    assert t.num_hits_history == z["hits"].tolist() and t.num_miss_history == z["misses"].tolist()
    t.flush()
    np.testing.assert_array_equal(t.weight.numpy(), z["weight_after_flush"])


@pytest.mark.parametrize("strategy", [DATASET, LFU])
@pytest.mark.parametrize("depth", [0, 1])
@pytest.mark.parametrize("use_freq", [False, True])
@pytest.mark.parametrize("N,C,n_ids,s", [(3000, 400, 350, 1.05), (20000, 900, 700, 0.25), (257, 256, 200, 0.5),
                                         (5000, 5000, 3000, 0.25)])
def test_two_oracles_agree_on_random_streams(strategy, depth, use_freq, N, C, n_ids, s):
    """set-based numpy restatement == op-sequence-literal torch restatement: slots, maps, counters, payloads, host
    table, histories and totals after every call (ties, masking order and counter updates included)"""
    from oracle.cache_oracle_torch import TorchCachedParamMgr
    rng = np.random.default_rng(N * 31 + C + depth * 7 + int(use_freq))
    D = 8
    w = rng.standard_normal((N, D)).astype(np.float32)
    perm = rng.permutation(N)
    freq = id_freq_map(perm[power_law_ids(rng, N, 50000, s)], N) if use_freq else None
    a = OracleCachedParamMgr(w.copy(), C, strategy)
    t = TorchCachedParamMgr(torch.from_numpy(w.copy()), C, strategy)
    a.protect_depth = t.protect_depth = depth
    a.reorder(freq, 0.7)
    t.reorder(freq, 0.7)
    assert np.array_equal(a.idx_map, t.idx_map.numpy())
    _same_state(a, t, strategy == LFU)
    if depth:
        n_ids = n_ids // 3
    for c in range(25):
        ids = perm[power_law_ids(rng, N, n_ids, s)]
        sa = a.prepare_ids(ids)
        st = t.prepare_ids(torch.from_numpy(ids))
        assert np.array_equal(sa, st.numpy())
        assert set(a.traces[-1].evicted_rows.tolist()) == set(t.last_evicted_rows.tolist())
        u = np.unique(sa)
        a.cuda_cached_weight[u] *= np.float32(1.25)
        t.cuda_cached_weight[torch.from_numpy(u)] *= 1.25
        _same_state(a, t, strategy == LFU)
    assert a.num_hits_history == t.num_hits_history and a.num_miss_history == t.num_miss_history
    assert a.num_write_back_history == t.num_write_back_history
    assert (a.cache_miss, a.total_cache, a.cpu_to_cuda_numel, a.cuda_to_cpu_numel) == \
        (t._cache_miss, t._total_cache, t._cpu_to_cuda_numel, t._cuda_to_cpu_numel)
    a.flush()
    t.flush()
    _same_state(a, t, strategy == LFU)


@pytest.mark.parametrize("init_freq", [False, True])
def test_torch_literal_oracle_lfu_known_answer(init_freq):
    """upstream's only known-answer test, through the op-sequence-literal restatement"""
    from oracle.cache_oracle_torch import TorchCachedParamMgr
    t = TorchCachedParamMgr(torch.randn(5, 5), 3, LFU)
    t.reorder([4, 2, 1, 3, 1] if init_freq else None, warmup_ratio=1.0)
    for ids in LFU_SCRIPT:
        t.prepare_ids(torch.tensor(ids))
    assert t.num_hits_history[-6:] == [3, 0, 1, 0, 1, 1]


def _replay_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("replay_reference", GOLD / "replay_reference.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_parity_pin_kit_skips_cleanly_without_colossalai(capsys):
    """tests/golden/replay_reference.py: the one command that pins the cache path against the real upstream manager;
    without an importable ColossalAI (this container, the GPU box) it says so and exits 0."""
    rr = _replay_module()
    assert rr.upstream() is None
    assert rr.lfu_known_answer() is None
    assert rr.main([]) == 0
    assert "SKIPPED" in capsys.readouterr().out


@pytest.mark.parametrize("name,strategy", [("cache_dataset_freq", "dataset"), ("cache_lfu_freq", "lfu"),
                                           ("cache_lfu_nofreq", "lfu")])
def test_parity_pin_kit_comparison_logic(name, strategy):
    """the replay + diff logic on a manager with upstream's attribute names (the op-sequence-literal torch restatement):
    identical -> every call exact; victims swapped among LFU ties -> reported as tie order; a wrong victim -> a real
    difference"""
    from oracle.cache_oracle_torch import TorchCachedParamMgr
    rr = _replay_module()
    z = np.load(GOLD / f"{name}.npz")

    def factory(tamper=None):
        def make(weight, C, freq, warmup):
            m = TorchCachedParamMgr(torch.from_numpy(weight.copy()), C, LFU if strategy == "lfu" else DATASET)
            m.reorder(None if freq is None else torch.from_numpy(freq), warmup)
            m._calls = 0
            return m

        def prepare(m, ids):
            out = m.prepare_ids(torch.from_numpy(ids)).numpy()
            m._calls += 1
            if tamper is not None and m._calls == 6:
                tamper(m)
            return out

        def touch(m, slots):
            m.cuda_cached_weight[torch.from_numpy(np.unique(slots))] += 0.5

        return make, prepare, touch

    rep = rr.replay(strategy, z, factory())
    assert rep["exact"] == rep["calls"] and rep["different"] == 0 and rep["histories_equal"], rep

    def swap_two_slots(m):                       # same resident rows, other slots
        a, b = (m.cached_idx_map >= 0).nonzero().view(-1)[:2].tolist()
        ra, rb = int(m.cached_idx_map[a]), int(m.cached_idx_map[b])
        m.cached_idx_map[a], m.cached_idx_map[b] = rb, ra
        m.inverted_cached_idx[ra], m.inverted_cached_idx[rb] = b, a

    rep = rr.replay(strategy, z, factory(swap_two_slots))
    assert rep["exact"] == 5 and rep["slot_pairing_only"] >= 1 and rep["different"] == 0, rep

    def evict_a_wrong_row(m):                    # a resident row replaced by one that is not
        s = int((m.cached_idx_map >= 0).nonzero().view(-1)[0])
        old = int(m.cached_idx_map[s])
        new = int((m.inverted_cached_idx < 0).nonzero().view(-1)[0])
        m.cached_idx_map[s] = new
        m.inverted_cached_idx[old], m.inverted_cached_idx[new] = -1, s

    rep = rr.replay(strategy, z, factory(evict_a_wrong_row))
    assert rep["different"] == 1, rep


def _stand_in_colossalai():
    """module objects `colossalai`, `.nn`, `.nn.parallel`, `.nn.parallel.layers` holding upstream's three names with
    upstream's constructor signatures, backed by the op-sequence-literal restatement"""
    import enum
    import types
    from oracle.cache_oracle_torch import TorchCachedParamMgr

    class EvictionStrategy(enum.Enum):
        LFU = 1
        DATASET = 2

    class CachedParamMgr(TorchCachedParamMgr):
        def __init__(self, weight, cuda_row_num=0, buffer_size=0, pin_weight=True,
                     evict_strategy=EvictionStrategy.DATASET, async_copy=False):
            super().__init__(weight, cuda_row_num, "lfu" if evict_strategy is EvictionStrategy.LFU else "dataset")

    class CachedEmbeddingBag(torch.nn.Module):
        def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2.0,
                     scale_grad_by_freq=False, sparse=False, _weight=None, mode="mean", include_last_offset=False,
                     dtype=None, device=None, cache_ratio=0.01, ids_freq_mapping=None, warmup_ratio=0.7,
                     buffer_size=0, pin_weight=False, evict_strategy=EvictionStrategy.DATASET):
            super().__init__()
            rows = max(int(num_embeddings * cache_ratio), 1)
            self.cache_weight_mgr = CachedParamMgr(_weight, rows, buffer_size, pin_weight, evict_strategy)
            self.cache_weight_mgr.reorder(ids_freq_mapping, warmup_ratio)

        def forward(self, ids, offsets=None, per_sample_weights=None, shape_hook=None):
            with torch.no_grad():
                slots = self.cache_weight_mgr.prepare_ids(ids)
            return torch.nn.functional.embedding_bag(slots, self.cache_weight_mgr.cuda_cached_weight, offsets)

        @property
        def num_hits_history(self):
            return self.cache_weight_mgr.num_hits_history

    mods = {}
    for name in ("colossalai", "colossalai.nn", "colossalai.nn.parallel", "colossalai.nn.parallel.layers"):
        mods[name] = types.ModuleType(name)
        mods[name].__path__ = []                      # a package: `import a.b.c` walks through it
    layers = mods["colossalai.nn.parallel.layers"]
    layers.CachedParamMgr, layers.CachedEmbeddingBag, layers.EvictionStrategy = (CachedParamMgr, CachedEmbeddingBag,
                                                                                  EvictionStrategy)
    return mods


def test_parity_pin_kit_runs_end_to_end_through_a_stand_in_upstream_module(monkeypatch, capsys):
    """the part of tests/golden/replay_reference.py that will run on the day -- the
    `import colossalai.nn.parallel.layers` branch (the `.cache_embedding` sub-package absent, as in the pinned
    commit's neighbours), upstream's constructor signatures and attribute names, all FOUR streams and the LFU
    known-answer script through `main()` -- executed here with the restatement injected under upstream's module name.
    A wrong answer from the stand-in must come out as exit code 1."""
    rr = _replay_module()
    for name, mod in _stand_in_colossalai().items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(rr, "DEVICE", "cpu")
    up = rr.upstream()
    assert up is not None and up[0].__name__ == "CachedParamMgr"
    assert rr.lfu_known_answer() is True
    assert rr.main([]) == 0
    out = capsys.readouterr().out
    assert "SKIPPED" not in out and "IDENTICAL up to tie order" in out
    for name, _ in rr.STREAMS:
        assert f"{name}: " in out and "'different': 0" in out
    assert out.count("'histories_equal': True") == 4

    # a manager that evicts a wrong row on its sixth call: main() must say so and return 1
    layers = sys.modules["colossalai.nn.parallel.layers"]
    good = layers.CachedParamMgr

    class Wrong(good):
        def prepare_ids(self, ids):
            out = super().prepare_ids(ids)
            self._n = getattr(self, "_n", 0) + 1
            if self._n == 6:
                s = int((self.cached_idx_map >= 0).nonzero().view(-1)[0])
                old = int(self.cached_idx_map[s])
                new = int((self.inverted_cached_idx < 0).nonzero().view(-1)[0])
                self.cached_idx_map[s] = new
                self.inverted_cached_idx[old], self.inverted_cached_idx[new] = -1, s
            return out

    monkeypatch.setattr(layers, "CachedParamMgr", Wrong)
    assert rr.main([]) == 1
    assert "REAL DIFFERENCE" in capsys.readouterr().out
