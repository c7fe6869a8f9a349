"""Binary (npy) Criteo reader against a plain numpy restatement of the reference's batching rules
(recsys/datasets/criteo.py:38-249): contiguous rank shares, id % hash + table offset, batches that span files,
feature-major KJT values; and the id-frequency map (feature_counter.py:12-31)."""
import numpy as np
import pytest
import torch

from cachedembedding_amd.datasets import (BinaryCriteoNpy, CAT_FEATURE_COUNT, criteo_files, get_id_freq_map,
                                          rank_row_range)

HASHES = [7, 100, 13, 5, 1000, 3, 50, 11, 2, 977, 64, 31, 10, 9, 17, 128, 4, 33, 14, 555, 21, 77, 45, 12, 104, 35]


@pytest.fixture()
def criteo_dir(tmp_path):
    rng = np.random.default_rng(0)
    rows = [1000, 37, 513, 700]               # day_0..day_3; day_3 is the "final day" with days=4
    for d, n in enumerate(rows):
        np.save(tmp_path / f"day_{d}_dense.npy", rng.random((n, 13), dtype=np.float32))
        np.save(tmp_path / f"day_{d}_sparse.npy", rng.integers(0, 1 << 31, (n, 26), dtype=np.int64).astype(np.int32))
        np.save(tmp_path / f"day_{d}_labels.npy", rng.integers(0, 2, (n, 1), dtype=np.int32))
    return tmp_path, rows


def _expected(tmp_path, days, rank, world, B, tables):
    dense = np.concatenate([np.load(tmp_path / f"day_{d}_dense.npy") for d in days])
    sparse = np.concatenate([np.load(tmp_path / f"day_{d}_sparse.npy") for d in days]).astype(np.int64)
    labels = np.concatenate([np.load(tmp_path / f"day_{d}_labels.npy") for d in days]).reshape(-1)
    total = dense.shape[0]
    base, rem = divmod(total, world)
    lo = base * rank + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    h = np.array([HASHES[t] for t in tables])
    off = np.concatenate([[0], np.cumsum(h)[:-1]])
    sparse = sparse[lo:hi][:, tables] % h + off
    dense, labels = dense[lo:hi], labels[lo:hi]
    nb = (hi - lo) // B
    return [(dense[i * B:(i + 1) * B], sparse[i * B:(i + 1) * B].T.reshape(-1), labels[i * B:(i + 1) * B])
            for i in range(nb)]


@pytest.mark.parametrize("mmap", [False, True])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_batches_match_numpy_restatement(criteo_dir, mmap, world):
    tmp_path, _ = criteo_dir
    B = 64
    d, s, l = criteo_files(str(tmp_path), "train", days=4)
    assert len(d) == len(s) == len(l) == 3 and all("day_3" not in f for f in d + s + l)
    seen = 0
    for rank in range(world):
        ds = BinaryCriteoNpy(d, s, l, B, rank, world, mmap_mode=mmap, hashes=HASHES)
        exp = _expected(tmp_path, [0, 1, 2], rank, world, B, list(range(26)))
        assert len(ds) == len(exp)
        for got, (ed, es, el) in zip(ds, exp):
            vals, offs, stride = got["sparse"]
            assert stride == B and vals.dtype == torch.int64 and offs.dtype == torch.int32
            assert torch.equal(offs, torch.arange(26 * B + 1, dtype=torch.int32))
            np.testing.assert_array_equal(got["dense"].numpy(), ed)
            np.testing.assert_array_equal(vals.numpy(), es)
            np.testing.assert_array_equal(got["labels"].numpy(), el)
            assert int(vals.max()) < sum(HASHES)
        seen += len(ds)
    assert seen >= (1000 + 37 + 513) // B - world          # every rank drops < 1 batch of tail rows


def test_assigned_tables_and_val_test_halves(criteo_dir):
    tmp_path, _ = criteo_dir
    tables = [1, 4, 9, 25]
    d, s, l = criteo_files(str(tmp_path), "val", days=4)
    assert len(d) == 1 and "day_3" in d[0]
    # val = first half of the final day, test = second half: rank / rank + W of world 2W
    val = BinaryCriteoNpy(d, s, l, 50, 0, 2, hashes=HASHES, assigned_tables=tables)
    test = BinaryCriteoNpy(d, s, l, 50, 1, 2, hashes=HASHES, assigned_tables=tables)
    for ds, rank in ((val, 0), (test, 1)):
        exp = _expected(tmp_path, [3], rank, 2, 50, tables)
        assert len(ds) == len(exp) == 7
        for got, (ed, es, el) in zip(ds, exp):
            np.testing.assert_array_equal(got["sparse"][0].numpy(), es)
            assert got["sparse"][1].numel() == len(tables) * 50 + 1
            assert int(got["sparse"][0].max()) < sum(HASHES[t] for t in tables)   # offsets of the LOCAL table


def test_shuffle_keeps_rows_together(criteo_dir):
    tmp_path, _ = criteo_dir
    d, s, l = criteo_files(str(tmp_path), "train", days=4)
    plain = BinaryCriteoNpy(d, s, l, 32, hashes=HASHES)
    shuf = BinaryCriteoNpy(d, s, l, 32, hashes=HASHES, shuffle_batches=True, seed=3)
    moved = 0
    for a, b in zip(plain, shuf):
        ra = torch.cat([a["dense"], a["sparse"][0].view(26, 32).t().float(), a["labels"].view(-1, 1).float()], 1)
        rb = torch.cat([b["dense"], b["sparse"][0].view(26, 32).t().float(), b["labels"].view(-1, 1).float()], 1)
        moved += int(not torch.equal(ra, rb))
        key = lambda r: sorted(map(tuple, r.tolist()))
        assert key(ra) == key(rb)
    assert moved > 0


def test_rank_row_range_partitions_every_row():
    lengths = [5, 0, 17, 3, 100]
    for world in (1, 2, 3, 7, 125, 200):
        covered = []
        for r in range(world):
            for idx, (lo, hi) in rank_row_range(lengths, r, world).items():
                covered += [(idx, j) for j in range(lo, hi + 1)]
        assert covered == [(i, j) for i, n in enumerate(lengths) for j in range(n)]


def test_id_freq_map(criteo_dir, tmp_path):
    root, _ = criteo_dir
    _, s, _ = criteo_files(str(root), "train", days=4)
    cache = str(tmp_path / "id_freq_map.pt")
    freq = get_id_freq_map(s, HASHES, cache_path=cache)
    h = np.array(HASHES)
    off = np.concatenate([[0], np.cumsum(h)[:-1]])
    allrows = np.concatenate([np.load(p).astype(np.int64) for p in s]) % h + off
    np.testing.assert_array_equal(freq.numpy(), np.bincount(allrows.reshape(-1), minlength=h.sum()))
    assert freq.numel() == sum(HASHES) and int(freq.sum()) == allrows.size
    assert torch.equal(get_id_freq_map([], HASHES, cache_path=cache), freq)        # served from the cache file


def _write_parquet(root, rows_per_file, hashes, rg):
    import pyarrow as pa
    import pyarrow.parquet as pq
    rng = np.random.default_rng(1)
    (root / "train").mkdir()
    allcols = []
    for i, n in enumerate(rows_per_file):
        cols = {f"int_{k}": rng.random(n).astype(np.float32) for k in range(13)}
        cols.update({f"cat_{k}": rng.integers(0, hashes[k], n).astype(np.int64) for k in range(26)})
        cols["label"] = rng.integers(0, 2, n).astype(np.int32)
        pq.write_table(pa.table(cols), root / "train" / f"part_{i}.parquet", row_group_size=rg)
        allcols.append(cols)
    return {k: np.concatenate([c[k] for c in allcols]) for k in allcols[0]}


@pytest.mark.parametrize("drop_last", [True, False])
def test_parquet_reader_batches_span_row_groups_and_files(tmp_path, drop_last):
    from cachedembedding_amd.datasets import ParquetCriteo, parquet_files
    cols = _write_parquet(tmp_path, [130, 75, 201], HASHES, rg=48)
    files = parquet_files(str(tmp_path), "train")
    assert [f.rsplit("/", 1)[1] for f in files] == ["part_0.parquet", "part_1.parquet", "part_2.parquet"]
    B, tables = 64, [0, 3, 9, 25]
    ds = ParquetCriteo(files, B, hashes=HASHES, drop_last=drop_last, assigned_tables=tables)
    total = 130 + 75 + 201
    assert len(ds) == (total // B if drop_last else -(-total // B))
    h = np.array([HASHES[t] for t in tables])
    off = np.concatenate([[0], np.cumsum(h)[:-1]])
    got = list(ds)
    assert len(got) == len(ds)
    for i, b in enumerate(got):
        lo, hi = i * B, min(total, (i + 1) * B)
        n = hi - lo
        exp_sparse = np.stack([cols[f"cat_{t}"][lo:hi] + off[j] for j, t in enumerate(tables)]).reshape(-1)
        vals, offs, stride = b["sparse"]
        assert stride == n and offs.numel() == len(tables) * n + 1
        np.testing.assert_array_equal(vals.numpy(), exp_sparse)
        np.testing.assert_array_equal(b["dense"].numpy(), np.stack([cols[f"int_{k}"][lo:hi] for k in range(13)], 1))
        np.testing.assert_array_equal(b["labels"].numpy(), cols["label"][lo:hi])
    # seeded row-group shuffle: same multiset of rows, different order, reproducible per (seed, epoch)
    a = ParquetCriteo(files, B, hashes=HASHES, shuffle_row_groups=True, seed=7)
    b2 = ParquetCriteo(files, B, hashes=HASHES, shuffle_row_groups=True, seed=7)
    la, lb = list(a), list(b2)
    assert all(torch.equal(x["sparse"][0], y["sparse"][0]) for x, y in zip(la, lb))
    plain = torch.cat([x["labels"] for x in ParquetCriteo(files, B, hashes=HASHES, drop_last=False)])
    shuf = torch.cat([x["labels"] for x in ParquetCriteo(files, B, hashes=HASHES, drop_last=False,
                                                          shuffle_row_groups=True, seed=7)])
    assert plain.numel() == shuf.numel() == total and int(plain.sum()) == int(shuf.sum())
