"""Per-step KERNEL terms of the row-wise sharded step for W = 2 / 4 / 8 ranks, measured on ONE GPU.

No multi-GPU node is available to this build, so the step time at W ranks is modelled; this script
replaces the model's kernel terms by measurements.  It builds what RANK 0 of a W-rank job holds for a window -- its
host shard (rows with frequency rank % W == 0), its C / W-slot cache, the fixed-capacity buckets all W ranks would
send it (every rank's batch drawn from its own generator, deduplicated and bucketed by owner with the library's own
ce_dedupe_bucket_rows_padded_window), the owner-side cache op over them, the local index over "cache + receive
buffer" and the window's keys -- and then times, back to back between hipEvents, exactly the launches
GraphedShardedWindow._step issues per step, in order, WITHOUT the two all-to-alls (their payload is printed:
(W - 1) x capacity rows of 4 D bytes per direction and exchange):

    owner gather of the rows the peers ask for   ->  [row all-to-all]  ->  pooling from keys over cache + buffer
    -> zero-fill of the buffer -> fused fold + SGD over cache + buffer -> [gradient all-to-all] -> owner axpy

Usage: python profiles/sharded_terms.py [W ...]   (default 1 2 4 8) -> markdown on stdout
"""
import ctypes
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from cachedembedding_amd import _lib, synthetic  # noqa: E402
from cachedembedding_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402
from cachedembedding_amd.cache_mgr import CachedParamMgr, EvictionStrategy, HostTable  # noqa: E402
from cachedembedding_amd.functional import presort_window  # noqa: E402

B, F, D, P, REPS = 16384, 26, 128, 4, 5
dev = torch.device("cuda", 0)
sizes = synthetic.TABLES["criteo_1tb"]
N = sum(sizes)
n = B * F
offsets = torch.arange(n + 1, dtype=torch.int32, device=dev)
fgen = synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=1024, device=dev)
freq = fgen.id_freq_map(32)
order = torch.argsort(freq, descending=True, stable=True)
idx_map = torch.empty(N, device=dev, dtype=torch.int32)
idx_map[order] = torch.arange(N, device=dev, dtype=torch.int32)
del freq, order, fgen
grad = torch.randn(B, F, D, device=dev) * 1e-3


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(0)
    torch.cuda.synchronize()
    e0.record()
    for r in range(reps):
        fn(r)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps           # us


rows_out = []
for W in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    n_local = (N + W - 1) // W
    C_local = int(N * 0.01) // W
    table = HostTable.allocate(n_local, D).fill_uniform_(-1.0 / N, 1.0 / N, 1024)
    mgr = CachedParamMgr(table, C_local, evict_strategy=EvictionStrategy.DATASET, device=dev)
    mgr.reorder(None, 0.7)
    gens = [synthetic.SyntheticKJT(sizes, B, 1, "power_law", 0.25, seed=1024 + 1000 * r, device=dev) for r in range(W)]
    ids = [g.next_values(P).contiguous() for g in gens]                           # [P, n] per rank
    stamp = torch.empty(P * N, dtype=torch.int32, device=dev)
    ws = torch.empty(P * (W + 1) * n, dtype=torch.int32, device=dev)
    # bucket sizes first (a generous capacity), then the capacity bench.py would choose, then the real buckets
    def dedupe(r, cap):
        req = torch.full((P, W, cap), -1, dtype=torch.int64, device=dev)
        pos = torch.full((P, n), -1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(P, W, dtype=torch.int64, device=dev)
        ovf = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.ce_dedupe_bucket_rows_padded_window(ptr(ids[r]), n, P, ptr(idx_map), N, W, cap, ptr(stamp), None, ptr(ws),
                                                      ptr(req), ptr(pos), ptr(cnt), ptr(ovf), stream_ptr()))
        return req, pos, cnt, int(ovf.item())
    cnts = torch.cat([dedupe(r, 65536)[2].view(-1) for r in range(W)]).double()
    want = max(float(cnts.max()), float(cnts.mean() + 4.5 * cnts.std(unbiased=False)))
    cap = (int(want) + 1 + 255) // 256 * 256
    plans = [dedupe(r, cap) for r in range(W)]
    assert all(p_[3] == 0 for p_ in plans)
    serve = torch.stack([plans[r][0][:, 0, :] for r in range(W)])                  # [W, P, cap]: what rank 0 serves
    slots = mgr.prepare_ids(serve.reshape(-1).contiguous(), padded=True).view(W, P, cap)
    slots_pwc = slots.permute(1, 0, 2).contiguous().view(P, W * cap)
    idx0 = torch.empty(P, n, dtype=torch.int64, device=dev)
    check(lib.ce_exchange_local_index(ptr(plans[0][1]), n, P, ptr(slots_pwc), W * cap, 0, cap, C_local, ptr(idx0),
                                      stream_ptr()))
    tail = mgr.reserve_tail(W * cap)
    tab = mgr.cache_with_tail[:C_local + W * cap]
    keys = presort_window(idx0, C_local + W * cap, offsets=offsets, include_last_offset=True, hook_features=F,
                          identity_bags=True)
    remote = slots_pwc.clone()
    remote.view(P, W, cap)[:, 0] = -1
    cache = mgr.cuda_cached_weight
    sent = torch.empty(W * cap, D, device=dev)
    recv = torch.randn(W * cap, D, device=dev) * 1e-4
    out = torch.empty(B, F, D, device=dev)
    arange = torch.arange(W * cap + 1, dtype=torch.int32, device=dev)
    sp = stream_ptr()
    t = {}
    if W > 1:
        t["owner gather"] = timed(lambda r: check(lib.ce_bag_forward(
            ptr(cache), C_local, D, ptr(remote[r % P]), W * cap, ptr(arange), 0, W * cap, 1, None, _lib.CE_MODE_SUM, 0,
            ptr(sent), sp)), REPS * P)
    t["pooling from keys (cache + buffer)"] = timed(lambda r: check(lib.ce_bag_forward_src_keys(
        ptr(tab), tab.shape[0], D, n, ptr(keys[r % P].keys), ptr(out), sp)), REPS * P)
    if W > 1:
        t["zero-fill of the buffer"] = timed(lambda r: tail.zero_(), REPS * P)
    t["fused fold + SGD (cache + buffer)"] = timed(lambda r: check(lib.ce_bag_backward_sgd_presorted_src(
        ptr(tab), tab.shape[0], D, n, ptr(grad), 1.0, ptr(keys[r % P].keys), sp)), REPS * P)
    if W > 1:
        t["owner axpy of the returned deltas"] = timed(lambda r: check(lib.ce_rows_axpy(
            ptr(cache), C_local, D, ptr(remote[r % P]), W * cap, ptr(recv), -1.0, sp)), REPS * P)
    own = float((plans[0][2][:, 0]).double().mean())
    rows_out.append(dict(W=W, capacity=cap, mean_bucket=float(cnts.mean()), own_rows_per_batch=own,
                         unique_rows_per_batch=float(plans[0][2].sum(dim=1).double().mean()),
                         wire_MB_per_exchange_and_direction=(W - 1) * cap * D * 4 / 1e6, kernels_us=t,
                         kernels_us_total=sum(t.values())))
    del mgr, table, stamp, ws, tab, tail, cache
    torch.cuda.empty_cache()

print("# Row-wise sharded step: rank 0's kernel terms at W ranks, measured on one MI355X (no collectives)\n")
print("`python profiles/sharded_terms.py` -- B = 16384 per rank, F = 26, D = 128, Criteo-1TB ids (every rank its own "
      "generator), 1 % cache split over the ranks, fixed-capacity buckets as bench.py chooses them.  UNMEASURED on "
      "hardware: the two all-to-alls per step (payload in the last column).\n")
names = list(rows_out[-1]["kernels_us"])
print("| W | capacity (rows) | mean bucket | unique rows / batch | of which own | " + " | ".join(f"{k} (us)" for k in names) +
      " | kernels total (us) | wire MB per exchange, per direction |")
print("|" + "---|" * (7 + len(names)))
for r in rows_out:
    print(f"| {r['W']} | {r['capacity']} | {r['mean_bucket']:.0f} | {r['unique_rows_per_batch']:.0f} | {r['own_rows_per_batch']:.0f} | " +
          " | ".join(f"{r['kernels_us'].get(k, 0.0):.1f}" for k in names) +
          f" | {r['kernels_us_total']:.1f} | {r['wire_MB_per_exchange_and_direction']:.1f} |")
print("\n```json\n" + json.dumps(rows_out) + "\n```")
