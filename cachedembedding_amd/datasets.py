"""Criteo readers (binary npy day files, and the 1TB set's parquet parts below): the data formats on the input side
of the path.

What the reference feeds `_train` from for Criteo-Kaggle (recsys/datasets/criteo.py:38-249
`InMemoryBinaryCriteoIterDataPipe`, :377-413 file selection, :461-486 + recsys/datasets/feature_counter.py:12-31 for
the id-frequency map): per day three files `*dense*.npy` float [rows, 13], `*sparse*.npy` int [rows, 26],
`*labels*.npy` int [rows, 1]; every rank takes a contiguous share of the rows; categorical ids are folded with
`id % hash_size[table]` and offset into the concatenated table of the loader's (assigned) tables; a batch is
B consecutive rows, possibly spanning two files, with the sparse part as a feature-major KJT
(`values[f * B + b]`, offsets = arange).

Built for the box this runs on rather than as a torch DataLoader: the rank's share is folded/offset ONCE at load
time and stored batch-blocked and feature-major in pinned host memory, so an iteration is three zero-copy views
that `modules.FiniteDataIter` ships to HBM on its side stream (no per-batch transpose, no collate, no worker
processes).  `mmap_mode=True` keeps the files on disk and assembles each batch on demand instead.

A batch is `{"dense": f32[B, 13], "sparse": [values i64[F*B], offsets i32[F*B+1], B], "labels": i32[B]}` -- the
`[values, offsets, stride]` list is what `FusedSparseModules` / `examples/dlrm_main.py` take.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

CAT_FEATURE_COUNT = 26
INT_FEATURE_COUNT = 13


def rank_row_range(lengths: Sequence[int], rank: int, world_size: int) -> Dict[int, Tuple[int, int]]:
    """file index -> (first row, last row) inclusive, for the contiguous share of `rank`: total rows are split
    evenly, the first (total % world_size) ranks get one extra row."""
    total = int(sum(lengths))
    base, rem = divmod(total, world_size)
    left = base * rank + min(rank, rem)
    right = left + base + (1 if rank < rem else 0)          # exclusive
    out: Dict[int, Tuple[int, int]] = {}
    start = 0
    for idx, n in enumerate(lengths):
        lo, hi = max(left, start), min(right, start + n)
        if lo < hi:
            out[idx] = (lo - start, hi - start - 1)
        start += n
    return out


def _npy_rows(path: str) -> int:
    return int(np.load(path, mmap_mode="r").shape[0])


def criteo_files(dataset_dir: str, stage: str, days: int = 7) -> Tuple[List[str], List[str], List[str]]:
    """(dense, sparse, labels) file lists of a stage: train = every day but the last, val/test = the last day
    (the caller halves it by passing rank / rank + W with world 2W, as the reference does)."""
    stage = stage.lower()
    if stage not in ("train", "val", "test"):
        raise ValueError(f"Supplied stage was {stage}. Must be one of ['train', 'val', 'test'].")
    final = f"day_{days - 1}"
    names = [f for f in os.listdir(dataset_dir) if f.endswith(".npy")]
    names = [f for f in names if (final in f) != (stage == "train")]
    return tuple(sorted(os.path.join(dataset_dir, f) for f in names if kind in f)      # type: ignore[return-value]
                 for kind in ("dense", "sparse", "labels"))


def get_id_freq_map(sparse_paths: Sequence[str], hashes: Sequence[int], cache_path: Optional[str] = None
                    ) -> torch.Tensor:
    """occurrences of every row of the concatenated table over the given sparse files (int64[sum(hashes)])"""
    if cache_path and os.path.exists(cache_path):
        return torch.load(cache_path)
    h = np.asarray(hashes, dtype=np.int64).reshape(1, -1)
    off = np.concatenate([[0], np.cumsum(h[0])[:-1]]).reshape(1, -1)
    total = int(h.sum())
    freq = np.zeros(total, dtype=np.int64)
    for path in sparse_paths:
        arr = np.load(path, mmap_mode="r")
        for lo in range(0, arr.shape[0], 1 << 20):                 # bounded working set
            blk = np.asarray(arr[lo:lo + (1 << 20)], dtype=np.int64) % h + off
            freq += np.bincount(blk.reshape(-1), minlength=total)
    out = torch.from_numpy(freq)
    if cache_path and (not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0):
        torch.save(out, cache_path)
    return out


class BinaryCriteoNpy:
    """Iterable over the batches of this rank's share (see the module docstring)."""

    def __init__(self, dense_paths: Sequence[str], sparse_paths: Sequence[str], labels_paths: Sequence[str],
                 batch_size: int, rank: int = 0, world_size: int = 1, shuffle_batches: bool = False,
                 mmap_mode: bool = False, hashes: Optional[Sequence[int]] = None,
                 assigned_tables: Optional[Sequence[int]] = None, pin_memory: Optional[bool] = None, seed: int = 0):
        assert len(dense_paths) == len(sparse_paths) == len(labels_paths) and len(dense_paths) > 0
        self.batch_size, self.rank, self.world_size = int(batch_size), rank, world_size
        self.shuffle_batches, self.mmap_mode, self.seed = shuffle_batches, mmap_mode, seed
        self.assigned_tables = np.arange(CAT_FEATURE_COUNT) if assigned_tables is None else np.asarray(assigned_tables)
        F = len(self.assigned_tables)
        if hashes is not None:
            assert len(hashes) == CAT_FEATURE_COUNT
            self.hashes = np.asarray([hashes[t] for t in self.assigned_tables], dtype=np.int64).reshape(1, F)
            self.sparse_offsets = np.concatenate([[0], np.cumsum(self.hashes[0])[:-1]]).astype(np.int64).reshape(1, F)
        else:
            self.hashes = self.sparse_offsets = None
        self.pin = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        ranges = rank_row_range([_npy_rows(p) for p in dense_paths], rank, world_size)
        self._parts = []                                            # (dense, sparse, labels) array views per file
        for idx, (lo, hi) in ranges.items():
            self._parts.append(tuple(np.load(p[idx], mmap_mode="r")[lo:hi + 1]
                                     for p in (dense_paths, sparse_paths, labels_paths)))
        self.num_rows_per_file = [int(p[0].shape[0]) for p in self._parts]
        self.num_batches = sum(self.num_rows_per_file) // self.batch_size
        B = self.batch_size
        self.offsets = torch.arange(0, F * B + 1, dtype=torch.int32)
        self.stride = B
        self._epoch = 0
        if not mmap_mode:
            self._preload()

    # -- per-batch assembly ---------------------------------------------------------------------------------
    def _fold(self, sparse: np.ndarray) -> np.ndarray:
        sparse = np.asarray(sparse, dtype=np.int64)[:, self.assigned_tables]
        if self.hashes is not None:
            sparse = sparse % self.hashes + self.sparse_offsets
        return sparse

    def _rows(self, first: int, count: int):
        """rows [first, first+count) of the rank's share as (dense f32, folded sparse i64, labels i32) arrays"""
        d, s, l = [], [], []
        start = 0
        for (dense, sparse, labels), n in zip(self._parts, self.num_rows_per_file):
            lo, hi = max(first, start), min(first + count, start + n)
            if lo < hi:
                sl = slice(lo - start, hi - start)
                d.append(np.asarray(dense[sl], dtype=np.float32))
                s.append(self._fold(sparse[sl]))
                l.append(np.asarray(labels[sl], dtype=np.int32).reshape(-1))
            start += n
        return np.concatenate(d), np.concatenate(s), np.concatenate(l)

    def _alloc(self, shape, dtype) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, pin_memory=self.pin)

    def _preload(self) -> None:
        nb, B, F = self.num_batches, self.batch_size, len(self.assigned_tables)
        self._dense = self._alloc((nb, B, INT_FEATURE_COUNT), torch.float32)
        self._values = self._alloc((nb, F * B), torch.int64)
        self._labels = self._alloc((nb, B), torch.int32)
        step = max(1, (1 << 22) // B)                               # ~4 M rows per pass
        for b0 in range(0, nb, step):
            k = min(step, nb - b0)
            d, s, l = self._rows(b0 * B, k * B)
            self._dense[b0:b0 + k] = torch.from_numpy(d).view(k, B, -1)
            # feature-major inside each batch: [k, B, F] -> [k, F, B]
            self._values[b0:b0 + k] = torch.from_numpy(s).view(k, B, F).transpose(1, 2).reshape(k, F * B)
            self._labels[b0:b0 + k] = torch.from_numpy(l).view(k, B)
        self._parts = []                                            # release the file mappings

    def _batch(self, i: int, rng: Optional[np.random.Generator]):
        B, F = self.batch_size, len(self.assigned_tables)
        if self.mmap_mode:
            d, s, l = self._rows(i * B, B)
            dense, labels = torch.from_numpy(d), torch.from_numpy(l)
            values = torch.from_numpy(s).t().reshape(-1)
        else:
            dense, values, labels = self._dense[i], self._values[i], self._labels[i]
        if rng is not None:                                         # shuffle the rows of the batch in unison
            perm = torch.from_numpy(rng.permutation(B))
            dense, labels = dense[perm], labels[perm]
            values = values.view(F, B)[:, perm].reshape(-1)
        return {"dense": dense, "sparse": [values, self.offsets, self.stride], "labels": labels}

    def __len__(self) -> int:
        return self.num_batches

    def __iter__(self) -> Iterator[dict]:
        rng = np.random.default_rng(self.seed + self._epoch) if self.shuffle_batches else None
        self._epoch += 1
        for i in range(self.num_batches):
            yield self._batch(i, rng)


# ---------------------------------------------------------------------------------------------------------
# Parquet form of the Criteo-1TB set (NVTabular-preprocessed: `part_*.parquet` with columns int_0..int_12,
# cat_0..cat_25, label; categorical ids already folded into [0, hash_size)), recsys/datasets/criteo.py:252-376
# (`PetastormDataReader`) and :416-445 (split directories train / validation / test).
INT_NAMES = [f"int_{i}" for i in range(INT_FEATURE_COUNT)]
CAT_NAMES = [f"cat_{i}" for i in range(CAT_FEATURE_COUNT)]
LABEL_NAME = "label"


def parquet_files(dataset_dir: str, stage: str) -> List[str]:
    split = {"train": "train", "val": "validation", "test": "test"}[stage.lower()]
    d = os.path.join(dataset_dir, split)
    n = len([f for f in os.listdir(d) if f.endswith(".parquet")])
    return [os.path.join(d, f"part_{i}.parquet") for i in range(n)]


class ParquetCriteo:
    """Row groups are read one at a time with pyarrow (in file order, or in a per-epoch seeded order with
    shuffle_row_groups) and cut into batches of `batch_size` rows that may span row groups and files; ids get the
    offset of their table in the loader's concatenated table (no modulo: the files hold folded ids).  Batches have
    the same dict form as BinaryCriteoNpy.  Single reader per process (the reference raises for world_size > 1
    as well); `rank` / `world_size` split the FILE list round-robin when given."""

    def __init__(self, paths: Sequence[str], batch_size: int, rank: Optional[int] = None,
                 world_size: Optional[int] = None, shuffle_batches: bool = False,
                 hashes: Optional[Sequence[int]] = None, seed: int = 1024, drop_last: bool = True,
                 assigned_tables: Optional[Sequence[int]] = None, shuffle_row_groups: bool = False):
        import pyarrow.parquet as pq
        self._pq = pq
        self.paths = list(paths)
        if world_size is not None and world_size > 1:
            self.paths = self.paths[(rank or 0)::world_size]
        self.batch_size, self.shuffle_batches, self.seed, self.drop_last = int(batch_size), shuffle_batches, seed, drop_last
        self.shuffle_row_groups = shuffle_row_groups
        self.assigned_tables = np.arange(CAT_FEATURE_COUNT) if assigned_tables is None else np.asarray(assigned_tables)
        F = len(self.assigned_tables)
        self.keys = [CAT_NAMES[t] for t in self.assigned_tables]
        if hashes is not None:
            h = np.asarray([hashes[t] for t in self.assigned_tables], dtype=np.int64)
            self.sparse_offsets = np.concatenate([[0], np.cumsum(h)[:-1]]).astype(np.int64).reshape(F, 1)
        else:
            self.sparse_offsets = None
        self._groups = []                                   # (path, row group index, rows)
        for p in self.paths:
            md = pq.ParquetFile(p).metadata
            self._groups += [(p, g, md.row_group(g).num_rows) for g in range(md.num_row_groups)]
        rows = sum(g[2] for g in self._groups)
        self.num_batches = rows // self.batch_size if drop_last else (rows + self.batch_size - 1) // self.batch_size
        self.offsets = torch.arange(0, F * self.batch_size + 1, dtype=torch.int32)
        self.stride = self.batch_size
        self.epoch = 0

    def __len__(self) -> int:
        return self.num_batches

    def _emit(self, dense, sparse, labels, rng):
        n = dense.shape[0]
        if self.shuffle_batches:
            perm = rng.permutation(n)
            dense, sparse, labels = dense[perm], sparse[:, perm], labels[perm]
        offsets = self.offsets if n == self.batch_size else torch.arange(0, sparse.shape[0] * n + 1, dtype=torch.int32)
        return {"dense": torch.from_numpy(np.ascontiguousarray(dense)),
                "sparse": [torch.from_numpy(np.ascontiguousarray(sparse).reshape(-1)), offsets, n],
                "labels": torch.from_numpy(np.ascontiguousarray(labels))}

    def __iter__(self) -> Iterator[dict]:
        rng = np.random.default_rng(self.seed + self.epoch)
        self.epoch += 1
        order = rng.permutation(len(self._groups)) if self.shuffle_row_groups else range(len(self._groups))
        B = self.batch_size
        buf = None                                          # < B rows carried over from the previous row group
        files = {}
        cols = INT_NAMES + self.keys + [LABEL_NAME]
        for gi in order:
            path, g, _ = self._groups[gi]
            if path not in files:
                files[path] = self._pq.ParquetFile(path)
            t = files[path].read_row_group(g, columns=cols)
            dense = np.stack([t.column(c).to_numpy().astype(np.float32, copy=False) for c in INT_NAMES], axis=1)
            sparse = np.stack([t.column(c).to_numpy().astype(np.int64, copy=False) for c in self.keys], axis=0)
            if self.sparse_offsets is not None:
                sparse = sparse + self.sparse_offsets
            labels = t.column(LABEL_NAME).to_numpy().astype(np.int32, copy=False).reshape(-1)
            if buf is not None:
                dense = np.concatenate([buf[0], dense], axis=0)
                sparse = np.concatenate([buf[1], sparse], axis=1)
                labels = np.concatenate([buf[2], labels], axis=0)
            n, s = dense.shape[0], 0
            while n - s >= B:
                yield self._emit(dense[s:s + B], sparse[:, s:s + B], labels[s:s + B], rng)
                s += B
            buf = (dense[s:], sparse[:, s:], labels[s:]) if s < n else None
        if buf is not None and not self.drop_last:
            yield self._emit(*buf, rng)
