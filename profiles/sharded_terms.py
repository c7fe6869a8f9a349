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

Round 5: the same with the EARLY / LATE split of both exchanges (GraphedShardedWindow(split=True), the default at W > 1).
Every owner's classification is computed with the library's own ce_split_classify over what all W ranks ask it for,
rank 0's places with ce_split_places, its two indices with ce_exchange_local_index_split; reported per W: the early /
deferred fraction of the distinct rows on the bench id stream, the fitted capacities, the bytes each of the four
messages carries, and the kernel terms ON the step's critical path (late gather, pooling, fold + SGD, urgent axpy) next
to the ones that run beside the step (zero-fill of the delta buffers -- round 6: on the communication stream, behind the
late rows --, early gather, deferred axpy).

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
    split = None
    if W > 1:
        # ---- the split: every owner classifies what all ranks ask it for; rank 0 needs its own classification (what it
        # serves) and, for what it requests, the flags owner o computed for requester 0
        n_loc = (N + W - 1) // W
        mask = torch.zeros(n_loc, dtype=torch.int64, device=dev)
        flags_by_owner = []
        for o in range(W):
            serve_o = torch.stack([plans[r][0][:, o, :] for r in range(W)]).contiguous()      # [W, P, cap]
            fl = torch.zeros(W, P, cap, dtype=torch.uint8, device=dev)
            check(lib.ce_split_classify(ptr(serve_o), W, P, cap, n_loc, None, 0, ptr(mask), ptr(fl), stream_ptr()))
            flags_by_owner.append(fl)
        flags_o = flags_by_owner[0]                                                           # rank 0 as owner
        flags_r = torch.stack([flags_by_owner[o][0] for o in range(W)], dim=1).contiguous()   # [P, W(owner), cap]
        big = torch.tensor([[cap] * 4] * P, dtype=torch.int32, device=dev)
        cnt4 = torch.zeros(P, W, 4, dtype=torch.int32, device=dev)
        ovf = torch.zeros(1, dtype=torch.int32, device=dev)
        pf = torch.empty(P, W * cap, dtype=torch.int32, device=dev)
        pb = torch.empty_like(pf)
        serve_pwc = serve.permute(1, 0, 2).contiguous()
        flo_pwc = flags_o.permute(1, 0, 2).contiguous()
        check(lib.ce_split_places(ptr(serve_pwc), ptr(flo_pwc), P, W, cap, -1, ptr(big), ptr(pf), ptr(pb), ptr(cnt4), ptr(ovf), stream_ptr()))
        c4 = cnt4.double().cpu()
        fwd_c, bwd_c = c4[1:], c4[:-1]              # batch 0 is all early, the last batch all urgent: not representative
        need = lambda x: max(float(x.max()), float(x.mean() + 4.5 * x.std(unbiased=False)))
        r128 = lambda v: min(cap, (int(v) + 1 + 127) // 128 * 128)
        cl, cu = r128(need(fwd_c[..., 1])), r128(need(bwd_c[..., 3]))
        ce_ = max(r128(need(fwd_c[..., 0])), min(cap, cap - cl + 128))
        cd = max(r128(need(bwd_c[..., 2])), min(cap, cap - cu + 128))
        caps = torch.tensor([[ce_, cl, cd, cu]] * P, dtype=torch.int32)
        caps[P - 1, 2], caps[P - 1, 3] = 1, cap
        caps_d = caps.to(dev)
        ne, nl, nd, nu = W * ce_, W * cl, W * cd, W * cap
        T = 2 * ne + nl + 2 * nd + nu
        pf_s, pb_s = torch.empty_like(pf), torch.empty_like(pf)
        check(lib.ce_split_places(ptr(serve_pwc), ptr(flo_pwc), P, W, cap, 0, ptr(caps_d), ptr(pf_s), ptr(pb_s), None, ptr(ovf), stream_ptr()))
        pf_r, pb_r = torch.empty_like(pf), torch.empty_like(pf)
        req0 = plans[0][0].contiguous()                                                       # [P, W, cap]
        check(lib.ce_split_places(ptr(req0), ptr(flags_r), P, W, cap, 0, ptr(caps_d), ptr(pf_r), ptr(pb_r), None, ptr(ovf), stream_ptr()))
        assert int(ovf.item()) == 0, "a class did not fit its fitted capacity on this window"
        tail = mgr.reserve_tail(T)
        tab = mgr.cache_with_tail[:C_local + T]
        cache = mgr.cuda_cached_weight
        idx_f = torch.empty(P, n, dtype=torch.int64, device=dev)
        idx_b = torch.empty_like(idx_f)
        check(lib.ce_exchange_local_index_split(ptr(plans[0][1]), n, P, ptr(slots_pwc), ptr(pf_r), ptr(pb_r), W * cap, 0, cap,
                                                C_local, 2 * ne + nl, ptr(caps_d), W, ne, nu, nd, ptr(idx_f), ptr(idx_b),
                                                stream_ptr()))
        keys_f = presort_window(idx_f, C_local + T, offsets=offsets, include_last_offset=True, hook_features=F, identity_bags=True)
        keys_b = presort_window(idx_b, C_local + T, offsets=offsets, include_last_offset=True, hook_features=F, identity_bags=True)
        lists = {}
        capl = caps_d.long()
        for name, pl, first, col in (("f", pf_s, ne, 0), ("b", pb_s, nd, 2)):
            width = (ne + nl + 1) if name == "f" else (nd + nu + 1)
            L_ = torch.full((P, width), -1, dtype=torch.int64, device=dev)
            wfirst = (W * capl[:, col]).unsqueeze(1)
            p64 = pl.long()
            dest = torch.where(p64 < wfirst, p64, first + p64 - wfirst)
            dest = torch.where(p64 >= 0, dest, torch.full_like(dest, width - 1))
            L_.scatter_(1, dest, slots_pwc)
            L_[:, -1] = -1
            lists[name] = L_
        sendE, sendL = torch.empty(ne, D, device=dev), torch.empty(nl, D, device=dev)
        recvU, recvD = torch.randn(nu, D, device=dev) * 1e-4, torch.randn(nd, D, device=dev) * 1e-4
        steps = [i for i in range(P - 1)] or [0]                  # (the window's last step returns everything at once)
        ar = torch.arange(max(ne, nl) + 1, dtype=torch.int32, device=dev)
        def gather(lst, nrows, out_):
            return lambda r: check(lib.ce_bag_forward(ptr(cache), C_local, D, ptr(lst[steps[r % len(steps)]]), nrows, ptr(ar), 0, nrows, 1,
                                                      None, _lib.CE_MODE_SUM, 0, ptr(out_), sp))
        st = {}
        st["late gather (critical)"] = timed(gather([lists["f"][i, ne:ne + nl] for i in range(P)], nl, sendL), REPS * P)
        st["pooling from keys (critical)"] = timed(lambda r: check(lib.ce_bag_forward_src_keys(
            ptr(tab), tab.shape[0], D, n, ptr(keys_f[steps[r % len(steps)]].keys), ptr(out), sp)), REPS * P)
        b0 = 2 * ne + nl
        st["zero-fill of the delta buffers (beside the step: behind the late rows on the communication stream, round 6)"] = timed(lambda r: tail[b0 + (r & 1) * nd:b0 + (r & 1) * nd + nd + nu].zero_(), REPS * P)
        st["fused fold + SGD (critical)"] = timed(lambda r: check(lib.ce_bag_backward_sgd_presorted_src(
            ptr(tab), tab.shape[0], D, n, ptr(grad), 1.0, ptr(keys_b[steps[r % len(steps)]].keys), sp)), REPS * P)
        st["urgent axpy (critical)"] = timed(lambda r: check(lib.ce_rows_axpy(
            ptr(cache), C_local, D, ptr(lists["b"][steps[r % len(steps)], nd:nd + W * cu]), W * cu, ptr(recvU), -1.0, sp)), REPS * P)
        st["early gather (beside the step)"] = timed(gather([lists["f"][i, :ne] for i in range(P)], ne, sendE), REPS * P)
        st["deferred axpy (beside the step)"] = timed(lambda r: check(lib.ce_rows_axpy(
            ptr(cache), C_local, D, ptr(lists["b"][steps[r % len(steps)], :nd]), nd, ptr(recvD), -1.0, sp)), REPS * P)
        crit = sum(v for k, v in st.items() if "critical" in k)
        row_b = D * 4
        split = dict(early_frac=float(fwd_c[..., 0].sum() / fwd_c[..., :2].sum()), deferred_frac=float(bwd_c[..., 2].sum() / bwd_c[..., 2:].sum()),
                     caps=dict(early=ce_, late=cl, deferred=cd, urgent=cu), kernels_us=st, critical_kernels_us=crit,
                     wire_MB=dict(late=(W - 1) * cl * row_b / 1e6, urgent=(W - 1) * cu * row_b / 1e6,
                                  early=(W - 1) * ce_ * row_b / 1e6, deferred=(W - 1) * cd * row_b / 1e6,
                                  unsplit=(W - 1) * cap * row_b / 1e6))
    own = float((plans[0][2][:, 0]).double().mean())
    rows_out.append(dict(W=W, capacity=cap, mean_bucket=float(cnts.mean()), own_rows_per_batch=own,
                         unique_rows_per_batch=float(plans[0][2].sum(dim=1).double().mean()),
                         wire_MB_per_exchange_and_direction=(W - 1) * cap * D * 4 / 1e6, kernels_us=t,
                         kernels_us_total=sum(t.values()), split=split))
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
print("\n## With the early / late split (round 5; round 6: the delta buffers are zeroed on the communication stream)\n")
print("early = share of a step's distinct remote rows that no rank looked up in the step before (they leave the owner while "
      "that step computes); deferred = share whose gradient no rank needs in the step after.  Capacities: mean + 4.5 sigma of "
      "each class over the window's (batch, peer) chunks, rounded to 128 rows.  Critical = between two steps' pooling; the "
      "early gather and the deferred axpy run on the communication stream beside the step.\n")
print("| W | early | deferred | cap early / late / deferred / urgent | wire MB late + urgent (critical) | early + deferred (beside) | "
      "unsplit, per exchange | critical kernels (us): late gather + pooling + fold + urgent axpy | beside: zero-fill, early gather, deferred axpy (us) |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows_out:
    sp_ = r.get("split")
    if not sp_:
        continue
    k = sp_["kernels_us"]
    crit = [v for kk, v in k.items() if "critical" in kk]
    off = [v for kk, v in k.items() if "beside" in kk]
    w_ = sp_["wire_MB"]
    c_ = sp_["caps"]
    print(f"| {r['W']} | {sp_['early_frac']:.3f} | {sp_['deferred_frac']:.3f} | {c_['early']} / {c_['late']} / {c_['deferred']} / {c_['urgent']} | "
          f"{w_['late']:.1f} + {w_['urgent']:.1f} | {w_['early']:.1f} + {w_['deferred']:.1f} | {w_['unsplit']:.1f} | "
          + " + ".join(f"{v:.1f}" for v in crit) + f" = {sum(crit):.1f} | " + ", ".join(f"{v:.1f}" for v in off) + " |")
print("\n```json\n" + json.dumps(rows_out) + "\n```")
