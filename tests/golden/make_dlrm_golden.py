"""Generates tests/golden/dlrm_toy.npz: a tiny DLRM trained 20 steps on the CPU with the calls the reference makes
around the hot path -- one fused nn.EmbeddingBag(mode='sum', include_last_offset=True, sparse=True) over the
concatenated tables (recsys/models/dlrm.py:70-81, 99-110 with cache_ratio = 1.0 is exactly that), the [F*B, D] ->
[B, F, D] shape hook (recsys/models/dlrm.py:26-27), BCE-with-logits, SGD on sparse + dense parameter groups
(recsys/dlrm_main.py:268-279, 455-461).  The dense part is examples/dlrm_main.py::DenseModules (stock torch.nn:
bottom MLP -> pairwise dots -> top MLP, the arch of recsys/models/dlrm.py:216-232).

Stored: initial table + dense parameters, the 20 batches (dense, ids, labels), and per step the pooled [B, F, D]
tensor the embedding produced and the loss.  tests/test_gpu_modules.py::test_toy_dlrm_matches_torch_cpu_trajectory replays
the batches through examples/dlrm_main.py's model + PrefetchWindow on the GPU and compares.
Run from the repo root:  python tests/golden/make_dlrm_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "examples"))
sys.path.insert(0, str(ROOT))

SIZES = [50, 30, 1000, 200]
B, D, NUM_DENSE, STEPS, LR = 64, 32, 13, 20, 0.1
DENSE_ARCH, OVER_ARCH = [64, 32], [64, 1]


def make_batches(g):
    """ids ~ a long tail per table (floor(u^-4) - 1 clipped: the reference generator's formula at s = 0.25,
    baselines/data/custom.py:76-93), global id = table id + exclusive cumsum offset, feature-major KJT values;
    the label is a learnable function of the inputs."""
    off = np.concatenate([[0], np.cumsum(SIZES)[:-1]])
    out = []
    for _ in range(STEPS):
        dense = torch.rand(B, NUM_DENSE, generator=g)
        ids = []
        for f, n in enumerate(SIZES):
            lo = (1.0 / n) ** 0.25
            u = torch.rand(B, generator=g, dtype=torch.float64) * (1 - lo) + lo
            local = torch.clamp(torch.floor(u ** -4.0).long() - 1, 0, n - 1)
            ids.append(local + int(off[f]))
        values = torch.stack(ids).reshape(-1)                      # [F*B], feature-major
        labels = ((dense[:, 0] + (values[:B] % 2).float() * 0.5) > 0.75).float()
        out.append((dense, values, labels))
    return out


def main():
    from dlrm_main import DenseModules
    torch.manual_seed(7)
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(1024)
    N, F = sum(SIZES), len(SIZES)
    table = (torch.rand(N, D, generator=g) - 0.5) * 0.2
    dense_mod = DenseModules(NUM_DENSE, F, D, DENSE_ARCH, OVER_ARCH)
    dense0 = {k: v.detach().clone().numpy() for k, v in dense_mod.state_dict().items()}
    batches = make_batches(g)
    emb = nn.EmbeddingBag.from_pretrained(table.clone(), freeze=False, mode="sum", include_last_offset=True, sparse=True)
    opt = torch.optim.SGD([{"params": emb.parameters(), "lr": LR}, {"params": dense_mod.parameters(), "lr": LR}])
    crit = nn.BCEWithLogitsLoss()
    offsets = torch.arange(F * B + 1, dtype=torch.int64)
    pooled, losses = [], []
    for dense, values, labels in batches:
        e = emb(values, offsets).view(F, B, D).transpose(0, 1)      # sparse_embedding_shape_hook
        pooled.append(e.detach().clone().numpy())
        loss = crit(dense_mod(dense, e).squeeze(-1), labels)
        losses.append(float(loss.detach()))
        opt.zero_grad()
        loss.backward()
        opt.step()
    rec = dict(sizes=np.array(SIZES), table=table.numpy(), lr=np.float32(LR), dense_arch=np.array(DENSE_ARCH),
               over_arch=np.array(OVER_ARCH),
               dense_x=np.stack([b[0].numpy() for b in batches]), values=np.stack([b[1].numpy() for b in batches]),
               labels=np.stack([b[2].numpy() for b in batches]), pooled=np.stack(pooled).astype(np.float32),
               losses=np.array(losses, np.float64), final_table=emb.weight.detach().numpy())
    for k, v in dense0.items():
        rec["dense." + k] = v
    np.savez_compressed(Path(__file__).resolve().parent / "dlrm_toy.npz", **rec)
    print("losses", np.round(losses, 4))


if __name__ == "__main__":
    main()
