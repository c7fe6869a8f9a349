"""Cache-manager checks that do not fit the small fixtures: oracle parity at medium size, size-independent
properties at the FULL Criteo-1TB index size (N = 177,944,275 rows, C = 1,779,442 slots, 3.4 M ids per call;
D = 4 keeps the host table at 2.8 GB -- the cache op does not depend on D), and hypothesis-driven random
streams against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ce():
    import cachedembedding_amd as ce
    return ce


@pytest.mark.parametrize("strategy", ["dataset", "lfu"])
def test_medium_size_exact_vs_oracle(strategy):
    ce = _ce()
    from oracle.cache_oracle import DATASET, LFU, OracleCachedParamMgr, id_freq_map, power_law_ids
    rng = np.random.default_rng(77)
    N, C, D, n_ids = 2_000_003, 150_000, 8, 400_000
    w = rng.standard_normal((N, D)).astype(np.float32)
    perm = rng.permutation(N)
    freq = id_freq_map(perm[power_law_ids(rng, N, 2_000_000, 0.25)], N)
    ora = OracleCachedParamMgr(w.copy(), C, LFU if strategy == "lfu" else DATASET)
    ora.reorder(freq, 0.7)
    mgr = ce.CachedParamMgr(torch.from_numpy(w.copy()), C,
                            evict_strategy=ce.EvictionStrategy.LFU if strategy == "lfu" else ce.EvictionStrategy.DATASET)
    mgr.reorder(freq, 0.7)
    for c in range(5):
        ids = perm[power_law_ids(rng, N, n_ids, 0.25)]
        exp = ora.prepare_ids(ids)
        got = mgr.prepare_ids(torch.from_numpy(ids).cuda())
        assert np.array_equal(got.cpu().numpy(), exp)
        assert np.array_equal(mgr.cached_idx_map.cpu().numpy().astype(np.int64), ora.cached_idx_map)
        if strategy == "lfu":
            assert np.array_equal(mgr.freq_cnter.cpu().numpy(), ora.freq_cnter)
    assert mgr.num_miss_history == ora.num_miss_history and mgr.num_write_back_history == ora.num_write_back_history
    assert sum(mgr.num_write_back_history) > 0, "the stream must exercise eviction"


@pytest.mark.parametrize("strategy", ["dataset", "lfu"])
def test_full_criteo1tb_index_size_properties(strategy):
    ce = _ce()
    from cachedembedding_amd import synthetic
    sizes = synthetic.CRITEO_1TB
    N, D, B, P = sum(sizes), 4, 16384, 8
    C = int(N * 0.01)
    assert N == 177_944_275 and C == 1_779_442
    gen = synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=11, device="cuda")
    freq = gen.id_freq_map(16)
    table = ce.HostTable.allocate(N, D)
    # payload pattern: row r holds (r, r+1, r+2, r+3) mod 2^20 -- lets every admitted row be verified
    t = table.tensor
    chunk = 1 << 24
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        t[s:e] = ((torch.arange(s, e).unsqueeze(1) + torch.arange(D)) % (1 << 20)).float()
    lfu = strategy == "lfu"
    mgr = ce.CachedParamMgr(table, C, evict_strategy=ce.EvictionStrategy.LFU if lfu else ce.EvictionStrategy.DATASET)
    mgr.reorder(freq, 0.7)
    id2row = mgr.idx_map.long()
    prev_resident = None
    for call in range(12):
        ids = gen.next_values(P).view(-1)                      # 3,407,872 ids
        slots = mgr.prepare_ids(ids)
        rows = id2row[ids]
        cim = mgr.cached_idx_map.long()
        inv = mgr.inverted_cached_idx.long()
        assert int(slots.min()) >= 0 and int(slots.max()) < C
        assert torch.equal(cim[slots], rows), "a lookup's slot must hold exactly the requested row"
        occ = torch.nonzero(cim >= 0).view(-1)
        assert torch.equal(inv[cim[occ]], occ), "cached_idx_map and inverted_cached_idx disagree"
        assert int((inv >= 0).sum()) == occ.numel() == C - mgr.cuda_available_row_num
        uniq = torch.unique(rows)
        assert mgr.num_hits_history[-1] + mgr.num_miss_history[-1] == uniq.numel()
        # payload: every resident row carries its own pattern (admit copied the right host row)
        samp = occ[torch.randint(0, occ.numel(), (200_000,), device="cuda")]
        exp = ((cim[samp].unsqueeze(1) + torch.arange(D, device="cuda")) % (1 << 20)).float()
        assert torch.equal(mgr.cuda_cached_weight.detach()[samp], exp)
        if prev_resident is not None and mgr.num_write_back_history[-1] > 0:
            evicted = prev_resident[inv[prev_resident] < 0]
            assert evicted.numel() == mgr.num_write_back_history[-1]
            assert not torch.isin(evicted, uniq).any(), "a row of the current call was evicted"
            if not lfu:
                # DATASET evicts the coldest (largest re-ranked row) among unprotected residents
                kept = prev_resident[inv[prev_resident] >= 0]
                kept_unprot = kept[~torch.isin(kept, uniq)]
                assert int(evicted.min()) > int(kept_unprot.max())
        prev_resident = cim[occ].clone()
    assert sum(mgr.num_write_back_history) > 0
    mgr.flush()
    assert mgr.cuda_available_row_num == C and int((mgr.inverted_cached_idx >= 0).sum()) == 0


def test_eviction_larger_than_the_staging_buffer_vs_oracle():
    """more victims in one call (> 262,144) than the HBM write-back staging holds: the overflow goes straight
    from the cache rows to the host table; state and host table must still match the oracle exactly"""
    ce = _ce()
    from oracle.cache_oracle import DATASET, OracleCachedParamMgr
    rng = np.random.default_rng(5)
    N, C, D = 1_500_000, 400_000, 4
    w = rng.standard_normal((N, D)).astype(np.float32)
    ora = OracleCachedParamMgr(w.copy(), C, DATASET)
    ora.reorder(None, 1.0)
    mgr = ce.CachedParamMgr(torch.from_numpy(w.copy()), C, evict_strategy=ce.EvictionStrategy.DATASET)
    mgr.reorder(None, 1.0)
    ora.cuda_cached_weight += np.float32(1.0)                 # "training" touched every resident row
    with torch.no_grad():
        mgr.cuda_cached_weight += 1.0
    ids = rng.permutation(np.arange(C, N))[:350_000]           # 350 k fresh rows -> 350 k evictions
    exp = ora.prepare_ids(ids)
    got = mgr.prepare_ids(torch.from_numpy(ids).cuda())
    assert mgr.num_write_back_history[-1] == 350_000 == ora.num_write_back_history[-1]
    assert np.array_equal(got.cpu().numpy(), exp)
    assert np.array_equal(mgr.cached_idx_map.cpu().numpy().astype(np.int64), ora.cached_idx_map)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(mgr.weight.numpy(), ora.weight)     # written-back payloads
    np.testing.assert_array_equal(mgr.cuda_cached_weight.detach().cpu().numpy(), ora.cuda_cached_weight)


@pytest.mark.parametrize("D", [7, 130])
def test_rows_not_multiple_of_16_bytes(D):
    ce = _ce()
    from oracle.cache_oracle import LFU, OracleCachedParamMgr
    rng = np.random.default_rng(D)
    N, C = 3000, 200
    w = rng.standard_normal((N, D)).astype(np.float32)
    ora = OracleCachedParamMgr(w.copy(), C, LFU)
    ora.reorder(None, 0.7)
    mgr = ce.CachedParamMgr(torch.from_numpy(w.copy()), C, evict_strategy=ce.EvictionStrategy.LFU)
    mgr.reorder(None, 0.7)
    for _ in range(10):
        ids = rng.integers(0, N, size=150)
        exp = ora.prepare_ids(ids)
        got = mgr.prepare_ids(torch.from_numpy(ids).cuda())
        assert np.array_equal(got.cpu().numpy(), exp)
        ora.cuda_cached_weight[np.unique(exp)] += np.float32(0.25)
        with torch.no_grad():
            mgr.cuda_cached_weight[torch.unique(got)] += 0.25
        np.testing.assert_array_equal(mgr.cuda_cached_weight.detach().cpu().numpy(), ora.cuda_cached_weight)
    mgr.flush(); ora.flush()
    np.testing.assert_array_equal(mgr.weight.numpy(), ora.weight)


def test_hypothesis_random_streams_vs_oracle():
    ce = _ce()
    hypothesis = pytest.importorskip("hypothesis")
    from hypothesis import HealthCheck, given, settings, strategies as st
    from oracle.cache_oracle import DATASET, LFU, OracleCachedParamMgr

    @settings(max_examples=30, deadline=None, suppress_health_check=list(HealthCheck))
    @given(st.integers(20, 400), st.integers(1, 40), st.sampled_from(["dataset", "lfu"]), st.booleans(),
           st.integers(0, 1), st.floats(0.0, 1.0), st.integers(0, 2 ** 31 - 1))
    def run(N, C, strategy, with_freq, depth, warm, seed):
        C = min(C, N)
        rng = np.random.default_rng(seed)
        w = rng.standard_normal((N, 4)).astype(np.float32)
        freq = rng.integers(0, 6, size=N) if with_freq else None
        ora = OracleCachedParamMgr(w.copy(), C, LFU if strategy == "lfu" else DATASET)
        ora.protect_depth = depth
        ora.reorder(freq, warm)
        mgr = ce.CachedParamMgr(torch.from_numpy(w.copy()), C, evict_strategy=ce.EvictionStrategy.LFU
                                if strategy == "lfu" else ce.EvictionStrategy.DATASET)
        mgr.reorder(freq, warm)
        mgr.set_protect_depth(depth)
        for _ in range(8):
            n = int(rng.integers(0, 3 * C + 2))
            ids = rng.integers(0, N, size=n)
            try:
                exp = ora.prepare_ids(ids)
            except AssertionError:
                with pytest.raises(AssertionError):
                    mgr.prepare_ids(torch.from_numpy(ids).cuda())
                if depth:          # after an overflow with a protected history the two may legitimately differ
                    return
                continue
            got = mgr.prepare_ids(torch.from_numpy(ids).cuda())
            assert np.array_equal(got.cpu().numpy(), exp)
            assert np.array_equal(mgr.cached_idx_map.cpu().numpy().astype(np.int64), ora.cached_idx_map)
            assert np.array_equal(mgr.inverted_cached_idx.cpu().numpy().astype(np.int64), ora.inverted_cached_idx)
            if strategy == "lfu":
                assert np.array_equal(mgr.freq_cnter.cpu().numpy(), ora.freq_cnter)
            np.testing.assert_array_equal(mgr.cuda_cached_weight.detach().cpu().numpy(), ora.cuda_cached_weight)

    run()


def test_lfu_counters_when_every_lookup_is_a_different_row():
    """k_slots_lfu counts in an 8192-entry LDS hash table per workgroup; 8192 distinct slots per workgroup fill it
    completely, so this stream drives the crowded-table path (direct atomics) as well as the table flush."""
    ce = _ce()
    from oracle.cache_oracle import LFU, OracleCachedParamMgr
    rng = np.random.default_rng(5)
    N, C, D, n_ids = 400_000, 200_000, 4, 120_000
    w = rng.standard_normal((N, D)).astype(np.float32)
    ora = OracleCachedParamMgr(w.copy(), C, LFU)
    ora.reorder(None, 0.0)
    mgr = ce.CachedParamMgr(torch.from_numpy(w.copy()), C, evict_strategy=ce.EvictionStrategy.LFU)
    mgr.reorder(None, 0.0)
    for c in range(4):
        ids = rng.permutation(N)[:n_ids]                     # all distinct
        if c == 3:
            ids = np.concatenate([ids[:60_000], ids[:60_000]])   # every row twice, far apart
        exp = ora.prepare_ids(ids)
        got = mgr.prepare_ids(torch.from_numpy(ids).cuda())
        assert np.array_equal(got.cpu().numpy(), exp)
        assert np.array_equal(mgr.freq_cnter.cpu().numpy(), ora.freq_cnter)
        assert np.array_equal(mgr.cached_idx_map.cpu().numpy().astype(np.int64), ora.cached_idx_map)
    assert sum(mgr.num_write_back_history) > 0


@pytest.mark.parametrize("strategy", ["dataset", "lfu"])
@pytest.mark.parametrize("workload,ratio,P,calls", [("criteo_kaggle", 0.05, 1, 40), ("avazu", 0.01, 1, 40),
                                                    ("criteo_1tb", 0.01, 8, 12)])
def test_baseline_configs_at_real_index_sizes_exact_vs_oracle(workload, ratio, P, calls, strategy):
    """BASELINE.json configs[1], [2], [4] (Criteo-Kaggle 5 % P=1, Criteo-1TB 1 % P=8, Avazu 1 % P=1) at their real
    table and cache sizes: slots, cached_idx_map, LFU counters and the hit / miss / write-back histories are identical
    to the oracle's, call by call."""
    ce = _ce()
    from cachedembedding_amd import synthetic
    from oracle.cache_oracle import DATASET, LFU, OracleCachedParamMgr
    sizes = synthetic.TABLES[workload]
    N, D, B = sum(sizes), 4, 16384
    C = int(N * ratio)
    assert (N, C) in ((33_762_577, 1_688_128), (9_445_823, 94_458), (177_944_275, 1_779_442))
    gen = synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=3, device="cuda")
    freq = gen.id_freq_map(8).cpu().numpy()
    w = (np.arange(N, dtype=np.float32) % 4099).reshape(N, 1).repeat(D, axis=1)
    lfu = strategy == "lfu"
    ora = OracleCachedParamMgr(w.copy(), C, LFU if lfu else DATASET)
    ora.reorder(freq, 0.98)         # nearly full after warm-up: evictions start within a few iterations
    mgr = ce.CachedParamMgr(torch.from_numpy(w), C,
                            evict_strategy=ce.EvictionStrategy.LFU if lfu else ce.EvictionStrategy.DATASET)
    mgr.reorder(freq, 0.98)
    assert np.array_equal(mgr.idx_map.cpu().numpy().astype(np.int64), ora.idx_map)
    for call in range(calls):
        ids = gen.next_values(P).view(-1)                      # P iterations' ids = one cache op
        got = mgr.prepare_ids(ids)
        exp = ora.prepare_ids(ids.cpu().numpy())
        assert np.array_equal(got.cpu().numpy(), exp)
        if call % 8 == 7 or call < 2 or call == calls - 1:
            assert np.array_equal(mgr.cached_idx_map.cpu().numpy().astype(np.int64), ora.cached_idx_map)
            if lfu:
                assert np.array_equal(mgr.freq_cnter.cpu().numpy(), ora.freq_cnter)
    assert mgr.num_hits_history == ora.num_hits_history and mgr.num_miss_history == ora.num_miss_history
    assert mgr.num_write_back_history == ora.num_write_back_history
    assert sum(ora.num_write_back_history) > 0, "the stream must reach eviction"


@pytest.mark.parametrize("transport", ["zerocopy", "worker"])
def test_config0_regime_whole_kaggle_table_resident_exact_vs_oracle(transport):
    """BASELINE.json configs[0]: Criteo-Kaggle, cache_ratio = 1.0 -- C = N = 33,762,577, every row fits, so after the
    warm-up (70 % preloaded) misses only ever fill free slots: the select never runs, nothing is ever written
    back ("K5: no select when free slots suffice").  Slots, maps and histories exact vs the oracle at
    the real index size (D = 4 keeps the tables at 0.5 GB; the cache op does not depend on D)."""
    ce = _ce()
    from cachedembedding_amd import synthetic
    from oracle.cache_oracle import DATASET, OracleCachedParamMgr
    sizes = synthetic.TABLES["criteo_kaggle"]
    N, D, B = sum(sizes), 4, 16384
    assert N == 33_762_577
    gen = synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=5, device="cuda")
    freq = gen.id_freq_map(8).cpu().numpy()
    rng = np.random.default_rng(0)
    w = rng.standard_normal((N, D)).astype(np.float32)
    ora = OracleCachedParamMgr(w.copy(), N, DATASET)
    ora.reorder(freq, 0.7)
    mgr = ce.CachedParamMgr(torch.from_numpy(w.copy()), N, evict_strategy=ce.EvictionStrategy.DATASET)
    mgr.reorder(freq, 0.7)
    mgr.set_transport(transport)
    assert np.array_equal(mgr.idx_map.cpu().numpy().astype(np.int64), ora.idx_map)
    for c in range(6):
        ids = gen.next_values(1).view(-1)                       # one batch: 425,984 ids (prefetch_num = 1)
        exp = ora.prepare_ids(ids.cpu().numpy())
        got = mgr.prepare_ids(ids)
        assert np.array_equal(got.cpu().numpy(), exp)
    assert np.array_equal(mgr.cached_idx_map.cpu().numpy().astype(np.int64), ora.cached_idx_map)
    assert mgr.num_hits_history == ora.num_hits_history and mgr.num_miss_history == ora.num_miss_history
    assert mgr.num_write_back_history == [0] * 6 == ora.num_write_back_history
    assert sum(mgr.num_miss_history) > 0
    assert mgr.cuda_available_row_num == ora.cuda_available_row_num > 0
    # admitted payloads are the host rows
    occ = np.nonzero(ora.cached_idx_map >= 0)[0]
    samp = torch.from_numpy(rng.choice(occ, size=200_000, replace=False)).cuda()
    np.testing.assert_array_equal(mgr.cuda_cached_weight.detach()[samp].cpu().numpy(),
                                  ora.cuda_cached_weight[samp.cpu().numpy()])


@pytest.mark.parametrize("strategy", ["lfu", "dataset"])
def test_micro_benchmark_shape_200_calls_exact_vs_oracle(strategy):
    """benchmark/benchmark_cache.py:58-72,83-95 in its own shape -- Avazu table (N = 9,445,823), B = 2048, F = 13,
    cache op in every iteration, 200 iterations -- ids exact vs the oracle on every call, counters at the end
    (BASELINE.json configs[4]: LFU evict-rate stress with power-law ids)."""
    ce = _ce()
    from cachedembedding_amd import synthetic
    from oracle.cache_oracle import DATASET, LFU, OracleCachedParamMgr
    sizes = synthetic.TABLES["avazu"]
    N, D, B = sum(sizes), 4, 2048
    C = int(N * 0.01)
    assert N == 9_445_823 and C == 94_458
    gen = synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=7, device="cuda")
    rng = np.random.default_rng(1)
    w = rng.standard_normal((N, D)).astype(np.float32)
    lfu = strategy == "lfu"
    ora = OracleCachedParamMgr(w.copy(), C, LFU if lfu else DATASET)
    ora.reorder(None, 0.7)                                       # benchmark_cache.py passes no frequency map (B#7)
    mgr = ce.CachedParamMgr(torch.from_numpy(w.copy()), C,
                            evict_strategy=ce.EvictionStrategy.LFU if lfu else ce.EvictionStrategy.DATASET)
    mgr.reorder(None, 0.7)
    for it in range(200):
        ids = gen.next_values(1).view(-1)                        # 26,624 ids
        exp = ora.prepare_ids(ids.cpu().numpy())
        got = mgr.prepare_ids(ids)
        assert np.array_equal(got.cpu().numpy(), exp), f"call {it}"
    assert np.array_equal(mgr.cached_idx_map.cpu().numpy().astype(np.int64), ora.cached_idx_map)
    if lfu:
        assert np.array_equal(mgr.freq_cnter.cpu().numpy(), ora.freq_cnter)
    assert mgr.num_hits_history == ora.num_hits_history and mgr.num_miss_history == ora.num_miss_history
    assert mgr.num_write_back_history == ora.num_write_back_history and sum(ora.num_write_back_history) > 0
    t = mgr.totals()
    assert (t["cache_miss"], t["total_cache"]) == (ora.cache_miss, ora.total_cache)
