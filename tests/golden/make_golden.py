"""Generates tests/golden/*.npz from the CPU oracle (oracle/cache_oracle.py, oracle/bag_oracle.py).

The reference tree holds no golden vectors for this path and none of its Python can be
imported (colossalai / torchrec absent) -- see SURVEY.md 8(c).  These files therefore pin
the *oracle itself* (regression vectors: any later edit of the oracle that changes a
result is caught) plus upstream ColossalAI's one known-answer test (LFU hit history).
Run from the repo root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import bag_oracle  # noqa: E402
from oracle.cache_oracle import DATASET, LFU, OracleCachedParamMgr, id_freq_map, power_law_ids  # noqa: E402

OUT = Path(__file__).resolve().parent


def cache_stream(strategy, with_freq, seed, N=1000, C=50, D=8, n_ids=64, calls=24, warmup=0.7, s=1.05):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((N, D)).astype(np.float32)
    perm = rng.permutation(N)                      # ids are not sorted by popularity
    sample = perm[power_law_ids(rng, N, 20000, s)]
    freq = id_freq_map(sample, N) if with_freq else None
    mgr = OracleCachedParamMgr(w.copy(), C, strategy)
    mgr.reorder(freq, warmup)
    rec = dict(weight=w, freq=np.zeros(0, np.int64) if freq is None else freq, idx_map=mgr.idx_map.copy(),
               cached_idx_map_0=mgr.cached_idx_map.copy())
    ids_all, slots_all, cim, frq, evs = [], [], [], [], []
    for c in range(calls):
        ids = perm[power_law_ids(rng, N, n_ids, s)]
        slots = mgr.prepare_ids(ids)
        # touch the cached rows like a training step would, so write-back is observable
        mgr.cuda_cached_weight[np.unique(slots)] += np.float32(0.5)
        ids_all.append(ids); slots_all.append(slots); cim.append(mgr.cached_idx_map.copy())
        frq.append(mgr.freq_cnter.copy() if mgr.freq_cnter is not None else np.zeros(0, np.int64))
        ev = np.sort(mgr.traces[-1].evicted_rows)
        evs.append(np.pad(ev, (0, n_ids - len(ev)), constant_values=-1))
    mgr.flush()
    rec.update(ids=np.stack(ids_all), slots=np.stack(slots_all), cached_idx_map=np.stack(cim),
               freq_cnter=np.stack(frq), evicted_rows=np.stack(evs), hits=np.array(mgr.num_hits_history),
               misses=np.array(mgr.num_miss_history), weight_after_flush=mgr.weight.copy(),
               meta=np.array([N, C, D, n_ids, calls, int(warmup * 1000)]))
    return rec


def bag_case(seed, N=97, D=12, nb=23, maxlen=5, weighted=False, mode="sum"):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, D, generator=g)
    lens = torch.randint(0, maxlen + 1, (nb,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)])
    nnz = int(offsets[-1])
    idx = torch.randint(0, N, (nnz,), generator=g)
    psw = torch.rand(nnz, generator=g) if weighted else None
    out = bag_oracle.bag_forward(w, idx, offsets, psw, mode, True)
    go = torch.randn(nb, D, generator=g)
    dw = bag_oracle.bag_backward_dense(N, idx, offsets, go, psw, mode, True)
    w1 = bag_oracle.sgd_step(w, idx, offsets, go, 0.5, psw, mode, True, sparse=True)
    return dict(weight=w.numpy(), indices=idx.numpy(), offsets=offsets.numpy(),
                psw=np.zeros(0, np.float32) if psw is None else psw.numpy(), out=out.numpy(),
                grad_out=go.numpy(), grad_weight=dw.numpy(), weight_after_sgd=w1.numpy())


def main():
    for strat, name in ((DATASET, "dataset"), (LFU, "lfu")):
        for wf in (False, True):
            rec = cache_stream(strat, wf, seed=1024 + (7 if wf else 0))
            np.savez_compressed(OUT / f"cache_{name}_{'freq' if wf else 'nofreq'}.npz", **rec)
    np.savez_compressed(OUT / "bag_sum.npz", **bag_case(1))
    np.savez_compressed(OUT / "bag_sum_weighted.npz", **bag_case(2, weighted=True))
    np.savez_compressed(OUT / "bag_mean.npz", **bag_case(3, mode="mean"))
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")))


if __name__ == "__main__":
    main()
