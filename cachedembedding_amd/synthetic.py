"""Synthetic KJT-shaped inputs for the hot path (no datasets exist on the GPU box).

Shapes follow the reference's data side (restated, not copied):
  * table sizes: recsys/datasets/criteo.py:30-35, recsys/datasets/avazu.py:34-35;
  * global id = per-table id + exclusive cumsum offset (recsys/datasets/criteo.py:118-119);
  * KJT layout: values = sparse[B, F].T.reshape(-1) (feature-major), lengths = ones,
    offsets = arange(F*B + 1) int32, stride = B (recsys/datasets/criteo.py:127-134,184-194);
  * long-tail id generator: baselines/data/custom.py:76-93 (u ~ U[(1/e)^s, 1],
    id = floor(u^(-1/s)) - 1, fp64, s = 0.25);
  * id frequency map = bincount of the ids (recsys/datasets/feature_counter.py:21-29).
Everything is generated with torch on the target device (plumbing, not the product).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

import torch

CRITEO_1TB = [45833188, 36746, 17245, 7413, 20243, 3, 7114, 1441, 62, 29275261, 1572176, 345138, 10, 2209, 11267,
              128, 4, 974, 14, 48937457, 11316796, 40094537, 452104, 12606, 104, 35]
CRITEO_KAGGLE = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
                 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
AVAZU = [7, 7, 4737, 7745, 26, 8552, 559, 36, 2686408, 6729486, 8251, 5, 4]
CUSTOM_POWER_LAW = [int(3e7), int(1e7), int(2e7), int(1e7), int(1e7), int(3e6), int(8e6), int(1e7), int(1e6),
                    int(1e6), int(1e6), int(1e6), int(5e6), 4000, 250, 250]

# 26 equal tables with the Criteo-1TB row total: with uniform ids a batch of 16384 samples then holds (almost) no row
# twice -- the no-reuse end of the reuse sweep (profiles/r04_reuse_sweep.md), where counted and algorithmic bytes agree
FLAT_178M = [6_844_011] * 25 + [6_844_000]

TABLES = {"criteo_1tb": CRITEO_1TB, "criteo_kaggle": CRITEO_KAGGLE, "avazu": AVAZU, "custom": CUSTOM_POWER_LAW,
          "flat_178m": FLAT_178M}


def scale_tables(sizes: Sequence[int], scale: float) -> List[int]:
    """Shrink every table by `scale` (for hosts that cannot pin the full table); >= 1 row each."""
    return [max(1, int(round(s * scale))) for s in sizes]


@dataclass
class SparseBatch:
    values: torch.Tensor    # int64 [F*B*L], feature-major
    offsets: torch.Tensor   # int32 [F*B + 1]
    stride: int             # B

    def as_list(self):
        """the [values, offsets, stride] list _train hands to the model (recsys/dlrm_main.py:253)"""
        return [self.values, self.offsets, self.stride]


class SyntheticKJT:
    def __init__(self, table_sizes: Sequence[int], batch_size: int, pooling: int = 1, dist: str = "power_law",
                 s: float = 0.25, seed: int = 1024, device="cuda", uniform_frac: float = 0.0):
        # uniform_frac (dist = "power_law"): this share of the lookups draws its id uniformly from the table instead --
        # a knob between the long-tail generator's ~9 % distinct rows per Criteo batch and the uniform one's ~54 %
        self.uniform_frac = float(uniform_frac)
        self.sizes = list(table_sizes)
        self.F = len(self.sizes)
        self.B = batch_size
        self.L = pooling
        self.dist = dist
        self.s = s
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.num_embeddings = sum(self.sizes)
        sz = torch.tensor(self.sizes, dtype=torch.float64, device=self.device)
        self._sizes_i = torch.tensor(self.sizes, dtype=torch.int64, device=self.device)
        self._lo = (1.0 / sz) ** s                                           # per-table lower bound of u
        self._table_off = torch.cumsum(self._sizes_i, 0) - self._sizes_i     # exclusive cumsum
        self.offsets = torch.arange(0, self.F * self.B + 1, dtype=torch.int32, device=self.device) * self.L

    def _local_ids(self, n_per_table: int) -> torch.Tensor:
        """[F, n] per-table ids."""
        u = torch.rand(self.F, n_per_table, dtype=torch.float64, device=self.device, generator=self.gen)
        if self.dist == "uniform":
            ids = torch.floor(u * self._sizes_i.unsqueeze(1).double()).long()
        else:
            lo = self._lo.unsqueeze(1)
            x = u * (1.0 - lo) + lo
            ids = torch.floor(1.0 / (x ** (1.0 / self.s))).long() - 1
            if self.uniform_frac > 0.0:
                pick = torch.rand(self.F, n_per_table, device=self.device, generator=self.gen) < self.uniform_frac
                u2 = torch.rand(self.F, n_per_table, dtype=torch.float64, device=self.device, generator=self.gen)
                ids = torch.where(pick, torch.floor(u2 * self._sizes_i.unsqueeze(1).double()).long(), ids)
        return torch.minimum(ids.clamp_(min=0), self._sizes_i.unsqueeze(1) - 1)

    def next_values(self, batches: int = 1) -> torch.Tensor:
        """`batches` batches of global ids, each feature-major [F*B*L]; returns [batches, F*B*L]."""
        n = batches * self.B * self.L
        ids = self._local_ids(n) + self._table_off.unsqueeze(1)          # [F, batches*B*L]
        return ids.view(self.F, batches, self.B * self.L).transpose(0, 1).reshape(batches, -1).contiguous()

    def next_batch(self) -> SparseBatch:
        return SparseBatch(self.next_values(1)[0], self.offsets, self.B)

    def id_freq_map(self, sample_batches: int = 64) -> torch.Tensor:
        """bincount over a sample of the same generator (GlobalFeatureCounter.compute restated)."""
        freq = torch.zeros(self.num_embeddings, dtype=torch.int64, device=self.device)
        done = 0
        while done < sample_batches:
            k = min(8, sample_batches - done)
            v = self.next_values(k).view(-1)
            freq += torch.bincount(v, minlength=self.num_embeddings)
            done += k
        return freq
