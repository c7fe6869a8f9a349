"""Workload for the rocprofv3 --pmc passes (HBM traffic per launch of the two dominant kernels).

  calib : k_bag_fwd over 425,984 DISTINCT rows of a 4M-row (2 GB) table -> known traffic
          (read 512 B + 8 B index + 4 B offset, write 512 B per lookup; no reuse, table >> 256 MB MALL).
          Used to calibrate FETCH_SIZE / WRITE_SIZE for this access pattern (MI355X_MICROARCH.md, HBM section:
          FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950; other patterns must be calibrated).
  bench : the kernels exactly as bench.py launches them at the default workload (Criteo-1TB ids, 1 % cache).
"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import cachedembedding_amd as ce  # noqa: E402
from cachedembedding_amd import synthetic  # noqa: E402
from cachedembedding_amd.functional import presort_window  # noqa: E402

mode = sys.argv[1]
# bench mode, optional: workload dist skew uniform_frac prefetch_num  (the reuse sweep: profiles/reuse_sweep.sh)
workload, dist_, skew, ufrac, P_arg = (sys.argv[2:7] + ["criteo_1tb", "power_law", "0.25", "0", "8"][len(sys.argv[2:7]):])
B, F, D = 16384, 26, 128
dev = torch.device("cuda", 0)
off = torch.arange(B * F + 1, dtype=torch.int32, device=dev)
if mode == "calib":
    C = 4_000_000
    w = torch.randn(C, D, device=dev).requires_grad_(True)
    grad = torch.randn(B, F, D, device=dev) * 1e-3
    fused = ce.FusedSGD(1.0)
    for it in range(12):
        idx = torch.randperm(C, device=dev)[:B * F]
        out = ce.embedding_bag(idx, w, off, mode="sum", include_last_offset=True, hook_features=F, sparse=True,
                               fused_sgd=fused)
        out.backward(grad)
    torch.cuda.synchronize()
else:
    sizes = synthetic.TABLES[workload]
    N = sum(sizes)
    gen = synthetic.SyntheticKJT(sizes, B, 1, dist_, float(skew), seed=1024, device=dev, uniform_frac=float(ufrac))
    freq = gen.id_freq_map(32)
    emb = ce.CachedEmbeddingBag(N, D, sparse=True, mode="sum", include_last_offset=True, cache_ratio=0.01,
                                ids_freq_mapping=freq, warmup_ratio=0.7, strict=False)
    emb.set_fused_sgd(1.0)
    emb.set_cache_op(False)
    grad = torch.randn(B, F, D, device=dev) * 1e-3
    P = int(P_arg)
    for win in range(max(14, 112 // P)):
        vals = gen.next_values(P)
        slots = emb.cache_weight_mgr.prepare_ids(vals.view(-1)).view(P, -1)
        # as bench.py does: source-row keys for the streaming backward
        keys = presort_window(slots.contiguous(), emb.cache_weight_mgr.cuda_row_num, offsets=off,
                              include_last_offset=True, hook_features=F, identity_bags=True)
        for i in range(P):
            out = emb(slots[i], off, hook_features=F, presorted=keys[i])
            out.backward(grad)
    torch.cuda.synchronize()
    st = emb.cache_weight_mgr.sync_stats()
    assert st.status == 0, "a cache op of the probe overflowed: lower prefetch_num"
print("done", mode)
