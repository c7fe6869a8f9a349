#!/usr/bin/env python
"""Counterpart of the reference's micro-benchmark (benchmark/benchmark_cache.py:21-75): a CachedEmbeddingBag
driven with cache_op=True every iteration (prepare_ids + forward), a fixed random upstream gradient,
backward, zero_grad, 200 iterations, then the manager's communication statistics.

Differences, all deliberate: the data is synthetic (no dataset on the box: per-table long-tail ids shaped like
the chosen dataset), and the constructor really receives cache_ratio / ids_freq_mapping / warmup_ratio (the
reference's call at :39-40 drops them).  Defaults follow the reference's driver block (:83-95):
batch 2048, dim 32, cache ratio 0.02, warm-up 0.7, Criteo-Kaggle tables.
"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import cachedembedding_amd as ce  # noqa: E402
from cachedembedding_amd import synthetic  # noqa: E402


def benchmark_cache_embedding(batch_size, embedding_dim, cache_ratio, id_freq_map=None, warmup_ratio=0.0,
                              use_lfu=False, tables="criteo_kaggle", iters=200, skew=0.25, fused_lr=None, seed=1024):
    sizes = synthetic.TABLES[tables]
    gen = synthetic.SyntheticKJT(sizes, batch_size, 1, "power_law", skew, seed=seed, device="cuda")
    num_embed = gen.num_embeddings
    cuda_row_num = int(cache_ratio * num_embed)
    print(f"batch size: {batch_size}, cached rows: {cuda_row_num},  cached_ratio {cuda_row_num / num_embed}")
    t0 = time.time()
    model = ce.CachedEmbeddingBag(num_embed, embedding_dim, sparse=True, include_last_offset=True, mode="sum",
                                  cache_ratio=cache_ratio, ids_freq_mapping=id_freq_map, warmup_ratio=warmup_ratio,
                                  evict_strategy=ce.EvictionStrategy.LFU if use_lfu else ce.EvictionStrategy.DATASET)
    print(f"model init: {time.time() - t0:.2f}s")
    if fused_lr is not None:
        model.set_fused_sgd(fused_lr)
    grad = None
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(iters + 1):
        sf = gen.next_batch()
        res = model(sf.values, sf.offsets)
        grad = torch.randn_like(res) if grad is None else grad
        res.backward(grad)
        model.zero_grad()
    torch.cuda.synchronize()
    dt = time.time() - t0
    lookups = (iters + 1) * sf.values.numel()
    print(f"{(iters + 1) / dt:.1f} it/s, {lookups / dt / 1e6:.1f} M lookups/s")
    model.cache_weight_mgr.print_comm_stats()
    hits, miss = sum(model.num_hits_history), sum(model.num_miss_history)
    print(f"unique-row hit rate {hits / max(1, hits + miss):.3f}")
    return (iters + 1) / dt


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=2048)
    ap.add_argument("--embedding_dim", type=int, default=32)
    ap.add_argument("--cache_ratio", type=float, default=0.02)
    ap.add_argument("--warmup_ratio", type=float, default=0.7)
    ap.add_argument("--tables", default="criteo_kaggle", choices=list(synthetic.TABLES))
    ap.add_argument("--use_lfu", action="store_true")
    ap.add_argument("--fused_lr", type=float, default=None)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    g = synthetic.SyntheticKJT(synthetic.TABLES[a.tables], a.batch_size, 1, "power_law", 0.25, seed=7, device="cuda")
    freq = g.id_freq_map(64)
    try:
        benchmark_cache_embedding(a.batch_size, a.embedding_dim, a.cache_ratio, freq, a.warmup_ratio, a.use_lfu,
                                  a.tables, a.iters, fused_lr=a.fused_lr)
    except AssertionError as ae:     # the overflow the reference catches at benchmark_cache.py:106-108
        print(f"batch size: {a.batch_size}, cache ratio: {a.cache_ratio}, raise error: {ae}")
