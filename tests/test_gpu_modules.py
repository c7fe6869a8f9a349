"""GPU tests of the glue around the operator: side-stream data iterator, FusedSparseModules wiring,
table-wise sharding, synthetic KJT generator, prefetch window (sequential, overlapped, graphed)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_finite_data_iter_side_stream():
    from cachedembedding_amd.modules import CudaStreamDataIter, FiniteDataIter
    batches = [dict(dense=torch.full((4, 3), float(i)).pin_memory(), sparse=[torch.arange(6) + i, torch.arange(7), 2])
               for i in range(5)]
    it = FiniteDataIter(batches)
    seen = []
    for b in it:
        assert b["dense"].is_cuda and b["sparse"][0].is_cuda and b["sparse"][2] == 2
        seen.append(int(b["dense"][0, 0].item()))
        assert torch.equal(b["sparse"][0].cpu(), torch.arange(6) + seen[-1])
    assert seen == [0, 1, 2, 3, 4]
    inf = CudaStreamDataIter(batches)
    got = [int(next(inf)["dense"][0, 0].item()) for _ in range(12)]
    assert got == [0, 1, 2, 3, 4] * 2 + [0, 1]


def test_synthetic_kjt_layout_and_generator():
    from cachedembedding_amd import synthetic
    sizes = [1000, 3, 50000, 17]
    gen = synthetic.SyntheticKJT(sizes, 64, 1, "power_law", 0.25, seed=3, device="cuda")
    b = gen.next_batch()
    F, B = 4, 64
    assert b.values.shape == (F * B,) and b.offsets.dtype == torch.int32 and b.stride == B
    assert torch.equal(b.offsets.cpu(), torch.arange(F * B + 1, dtype=torch.int32))
    off = np.concatenate([[0], np.cumsum(sizes)])
    v = b.values.view(F, B).cpu().numpy()                    # feature-major (criteo.py:127-134)
    for f in range(F):
        assert (v[f] >= off[f]).all() and (v[f] < off[f + 1]).all()
    freq = gen.id_freq_map(8)
    assert freq.shape == (sum(sizes),) and int(freq.sum()) == 8 * F * B
    # long tail: the first id of a big table is by far the most frequent
    assert freq[off[2]] > 20 * freq[off[2] + 100]


def test_fused_sparse_modules_matches_reference_wiring():
    from cachedembedding_amd.modules import FusedSparseModules
    torch.manual_seed(0)
    sizes, D, B = [50, 7, 300], 32, 16
    m = FusedSparseModules(sizes, D, reduction_mode="sum", sparse=True, use_cache=True, cache_ratio=0.5,
                           is_dist_dataloader=False)
    table = m.embed.weight.clone()                           # host table before training
    F = len(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    ids = torch.stack([torch.randint(int(off[f]), int(off[f + 1]), (B,)) for f in range(F)]).view(-1)
    offsets = torch.arange(F * B + 1, dtype=torch.int32)
    out = m([ids.cuda(), offsets.cuda(), B], cache_op=True)
    assert out.shape == (B, F, D)
    exp = table[ids].view(F, B, D).transpose(0, 1)
    assert torch.equal(out.cpu(), exp)
    m2 = FusedSparseModules(sizes, D, reduction_mode="sum", sparse=True, use_cache=True, cache_ratio=0.5,
                            is_dist_dataloader=False, fold_hook=True)
    out2 = m2([ids.cuda(), offsets.cuda(), B])
    assert out2.is_contiguous() and out2.shape == (B, F, D)
    with pytest.raises(NotImplementedError):
        FusedSparseModules(sizes, D, use_cache=False)
    with pytest.raises(TypeError):
        m(torch.zeros(3))


def test_tablewise_arrangement_balances_rows():
    from cachedembedding_amd import synthetic
    from cachedembedding_amd.tablewise import get_tablewise_rank_arrange, prepare_tablewise_config
    for sizes in (synthetic.CRITEO_KAGGLE, synthetic.CRITEO_1TB, synthetic.AVAZU):
        for W in (1, 2, 4, 8):
            arr = get_tablewise_rank_arrange(sizes, W)
            loads = [sum(s for s, r in zip(sizes, arr) if r == k) for k in range(W)]
            assert len(arr) == len(sizes) and max(loads) <= sum(sizes) / W + max(sizes)
    cfg = prepare_tablewise_config([100, 100000], 0.01, None, "criteo_kaggle", 2)
    assert [c.cuda_row_num for c in cfg] == [100, 3000] and {c.assigned_rank for c in cfg} == {0, 1}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tablewise(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from cachedembedding_amd.tablewise import ParallelCachedEmbeddingBagTablewise, TablewiseEmbeddingBagConfig
        torch.manual_seed(0)
        sizes, D, B = [40, 300, 25, 120], 16, 8
        tables = [torch.randn(n, D) for n in sizes]
        arrange = [0, 1, 1, 0] if world == 2 else [0, 0, 0, 0]
        cfgs = [TablewiseEmbeddingBagConfig(n, max(8, n // 4), arrange[i], initial_weight=tables[i])
                for i, n in enumerate(sizes)]
        m = ParallelCachedEmbeddingBagTablewise(cfgs, D, sparse=True, mode="sum", include_last_offset=True)
        mine = [i for i, r in enumerate(arrange) if r == rank]
        ids_per_table = [torch.randint(0, n, (B,)) for n in sizes]               # global batch, same on all ranks
        loc_off = np.concatenate([[0], np.cumsum([sizes[i] for i in mine])])
        values = torch.cat([ids_per_table[t] + int(loc_off[k]) for k, t in enumerate(mine)])
        offsets = torch.arange(len(mine) * B + 1, dtype=torch.int32)
        Fq = len(sizes)
        out = m(values.cuda(), offsets.cuda(), shape_hook=lambda x: x.view(x.shape[0], Fq, -1))
        order = [t for r in range(world) for t in range(Fq) if arrange[t] == r]    # rank-major table order
        exp = torch.stack([tables[t][ids_per_table[t]] for t in order], dim=1)      # [B, F, D]
        exp = torch.tensor_split(exp, world, dim=0)[rank]
        torch.testing.assert_close(out.cpu(), exp, rtol=1e-6, atol=1e-6)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "fail", traceback.format_exc()))


@pytest.mark.parametrize("world", [1, 2])
def test_tablewise_parallel(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_tablewise, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    res = [q.get() for _ in range(world) if not q.empty()]
    assert len(res) == world and all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("presort", [False, True, "src"])
@pytest.mark.parametrize("mode", ["sequential", "overlap", "overlap_interleaved", "overlap_auto", "graph",
                                  "graph_interleaved", "graph_switching", "graph_auto"])
def test_prefetch_window_modes_train_identically_to_plain_embedding_bag(mode, presort):
    """_train's window block in its three forms gives the same training trajectory as a plain full-table
    EmbeddingBag with SGD (each window's unique rows fit the cache even when two windows are protected)."""
    import cachedembedding_amd as ce
    from cachedembedding_amd.pipeline import GraphedWindow, PrefetchWindow
    torch.manual_seed(0)
    N, D, F, B, P, lr, nwin = 20000, 64, 4, 64, 4, 0.5, (10 if mode.endswith("_auto") else 6)
    # *_auto: the library's own trial (pipeline.ArrangementTrial) switches the arrangement while these windows train --
    # blocks of 2 windows, one per arrangement, then its verdict
    trial_args = dict(block_windows=2, rounds=1, settle=0)
    w0 = torch.randn(N, D)
    emb = ce.CachedEmbeddingBag(N, D, sparse=True, _weight=w0.clone(), mode="sum", include_last_offset=True,
                                cuda_row_num=4 * F * B * P, warmup_ratio=0.5, strict=False)
    emb.set_fused_sgd(lr)
    emb.set_cache_op(False)
    off = torch.arange(F * B + 1, dtype=torch.int32, device="cuda")
    # "src": the window's keys carry the grad_out row of every lookup (streaming backward)
    layout = (off, True, F) if presort == "src" else None
    presort = bool(presort)
    grad = (torch.randn(B, F, D) * 0.1).cuda()
    g = torch.Generator().manual_seed(5)
    windows = [[(torch.rand(F * B, generator=g) ** 3 * N).long().clamp_(0, N - 1) for _ in range(P)] for _ in range(nwin)]
    ref = w0.clone()

    def step(slots, i, keys=None):
        out = emb(slots, off, hook_features=F, presorted=keys)
        out.backward(grad)

    if mode in ("graph", "graph_interleaved", "graph_switching", "graph_auto"):
        il = mode == "graph_interleaved"        # the cache op in two halves around the previous window's steps, one stream
        # graph_switching: one object, the arrangement changed between windows by the caller; graph_auto: by the library
        plan = ["overlap", "interleaved", "interleaved", "overlap", "interleaved", "overlap", "overlap"]
        gw = GraphedWindow(emb, P, F * B, step, overlap=not il, warmup_values=[v.cuda() for v in windows[0]],
                           presort=presort, transport="worker", bag_layout=layout, interleaved=il,
                           arrangement=None if il else ("auto" if mode == "graph_auto" else "overlap"),
                           arrangement_trial=trial_args if mode == "graph_auto" else None)
        assert gw.arrangement == ("overlap" if mode in ("graph", "graph_switching") else "interleaved")
        # the capture warm-up trained on window 0 twice over (eager pass + nothing else): replay that on the ref
        for v in windows[0]:
            ref.index_add_(0, v, grad.cpu().transpose(0, 1).reshape(-1, D), alpha=-lr)
        if mode == "graph_switching":
            assert gw.switchable
            gw.set_arrangement(plan[0])
        gw.submit([v.cuda() for v in windows[0]], 0)
        for w in range(nwin):
            if w + 1 < nwin:
                if mode == "graph_switching":
                    gw.set_arrangement(plan[w + 1])
                    assert gw.arrangement == plan[w + 1]
                gw.submit([v.cuda() for v in windows[w + 1]], (w + 1) % 2)
            gw.run(w % 2)
        if mode == "graph_auto":
            assert gw.settle_arrangement(wait=True) in ("overlap", "interleaved") and gw.trial.trials == 1
            rep = gw.trial.report()
            assert set(rep["trial_ms_per_window"]) == {"overlap", "interleaved"}
            other = "overlap" if rep["mode"] == "interleaved" else "interleaved"
            assert max(rep["trial_ms_per_window"][rep["mode"]]) <= max(rep["trial_ms_per_window"][other])
            assert gw.arrangement == rep["mode"]
    else:
        ov = mode.startswith("overlap")
        win = PrefetchWindow(emb, P, overlap=ov, presort=presort, transport="worker", bag_layout=layout,
                             arrangement={"overlap": "overlap", "overlap_interleaved": "interleaved",
                                          "overlap_auto": "auto"}.get(mode),
                             arrangement_trial=trial_args if mode == "overlap_auto" else None)
        if ov:
            win.submit([v.cuda() for v in windows[0]])
        modes_seen = []
        for w in range(nwin):
            if ov:
                modes_seen.append("interleaved" if win._pending[0] is None else "overlap")
                slots = win.collect()
                if w + 1 < nwin:
                    win.submit([v.cuda() for v in windows[w + 1]])
            else:
                slots = win.prepare([v.cuda() for v in windows[w]])
            for i in range(P):
                step(slots[i], i, win.keys[i] if presort else None)
        if mode == "overlap_interleaved":
            assert set(modes_seen) == {"interleaved"}
        if mode == "overlap_auto":           # both arrangements trained real windows, then the trial settled
            assert {"interleaved", "overlap"} <= set(modes_seen)
            torch.cuda.synchronize()
            assert win.trial.poll() in ("overlap", "interleaved")
    for w in range(nwin):
        for v in windows[w]:
            ref.index_add_(0, v, grad.cpu().transpose(0, 1).reshape(-1, D), alpha=-lr)
    torch.cuda.synchronize()
    assert emb.cache_weight_mgr.sync_stats().status == 0
    emb.flush()
    torch.testing.assert_close(emb.weight, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("arr", ["overlap", "interleaved", "switching", "auto"])
@pytest.mark.parametrize("P,lfu,presort,transport", [(1, False, False, "worker"), (1, True, False, "zerocopy"),
                                                    (3, False, "src", "worker")])
def test_cache_op_two_windows_ahead(P, lfu, presort, transport, arr):
    """GraphedWindow(plan_ahead=2): three slot buffers, protect_depth 2, the cache op of window k+2 only waits for the
    training of window k-1 (the last reader of its buffer).  Same training trajectory as a plain full-table
    EmbeddingBag with SGD, and no row of a window that still trains is evicted (the table after flush says so).
    arr: where those cache ops run -- on the side stream ('overlap'), in two halves on the training stream around the
    steps of the window in training ('interleaved': what a prefetch_num = 1 pipeline on Kaggle 5 % prefers), changed
    by the caller before every window ('switching') or by the library's own trial ('auto', the default)."""
    import cachedembedding_amd as ce
    from cachedembedding_amd.pipeline import GraphedWindow
    torch.manual_seed(0)
    N, D, F, B, lr, nwin = 20000, 64, 4, 64, 0.5, 30
    w0 = torch.randn(N, D)
    off = torch.arange(F * B + 1, dtype=torch.int32, device="cuda")
    layout = (off, True, F) if presort == "src" else None
    grad = (torch.randn(B, F, D) * 0.1).cuda()
    g = torch.Generator().manual_seed(7)
    windows = [[(torch.rand(F * B, generator=g) ** 3 * N).long().clamp_(0, N - 1).cuda() for _ in range(P)]
               for _ in range(nwin)]
    emb = ce.CachedEmbeddingBag(N, D, sparse=True, _weight=w0.clone(), mode="sum", include_last_offset=True,
                                cuda_row_num=4 * F * B * P, warmup_ratio=0.5, strict=False,
                                evict_strategy=ce.EvictionStrategy.LFU if lfu else ce.EvictionStrategy.DATASET)
    emb.set_fused_sgd(lr)
    emb.set_cache_op(False)

    def step(slots, i, keys=None):
        out = emb(slots, off, hook_features=F, presorted=keys)
        out.backward(grad)

    gw = GraphedWindow(emb, P, F * B, step, overlap=True, warmup_values=windows[0], presort=bool(presort),
                       transport=transport, bag_layout=layout, plan_ahead=2,
                       arrangement=None if arr == "auto" else ("overlap" if arr == "switching" else arr),
                       arrangement_trial=dict(block_windows=4, rounds=1, settle=0) if arr == "auto" else None)
    assert gw.nbuf == 3 and gw.switchable
    plan = ["interleaved", "interleaved", "overlap", "interleaved", "overlap", "overlap", "overlap", "interleaved"]
    torch.cuda.synchronize()
    submitted = -1
    seen = set()
    for w in range(nwin):
        if arr == "switching":
            gw.set_arrangement(plan[w % len(plan)])
        seen.add(gw.arrangement)
        for w2 in range(submitted + 1, min(nwin, w + 3)):
            gw.submit(windows[w2], w2 % 3)
            submitted = w2
        gw.run(w % 3)
    if arr in ("switching", "auto"):
        assert seen == {"overlap", "interleaved"}
    else:
        assert seen == {arr}
    if arr == "auto":
        assert gw.settle_arrangement(wait=True) in ("overlap", "interleaved") and gw.trial.trials == 1
    torch.cuda.synchronize()
    assert emb.cache_weight_mgr.sync_stats().status == 0
    emb.flush()
    ref = w0.clone()
    for v in windows[0]:                     # the capture warm-up trained on window 0 once (eager pass)
        ref.index_add_(0, v.cpu(), grad.cpu().transpose(0, 1).reshape(-1, D), alpha=-lr)
    for w in range(nwin):
        for v in windows[w]:
            ref.index_add_(0, v.cpu(), grad.cpu().transpose(0, 1).reshape(-1, D), alpha=-lr)
    torch.testing.assert_close(emb.weight, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("P,lfu,presort", [(4, False, "src"), (1, False, False), (1, True, False), (2, True, "src")])
def test_cache_op_captured_in_the_window_graph(P, lfu, presort):
    """GraphedWindow(graph_cache_op=True): the next window's cache op is replayed from a hipGraph of its own beside
    the graph that trains the current one (zero-copy transport; the call number is counted on the device for
    replayed calls).  Same training
    trajectory as a plain full-table EmbeddingBag with SGD, and the same per-call records (hits / misses /
    write-backs, totals, number of calls) as the kernel-by-kernel pipeline on the same windows."""
    import cachedembedding_amd as ce
    from cachedembedding_amd.pipeline import GraphedWindow
    torch.manual_seed(0)
    N, D, F, B, lr, nwin = 20000, 64, 4, 64, 0.5, 40
    w0 = torch.randn(N, D)
    off = torch.arange(F * B + 1, dtype=torch.int32, device="cuda")
    layout = (off, True, F) if presort == "src" else None
    grad = (torch.randn(B, F, D) * 0.1).cuda()
    g = torch.Generator().manual_seed(5)
    windows = [[(torch.rand(F * B, generator=g) ** 3 * N).long().clamp_(0, N - 1) for _ in range(P)] for _ in range(nwin)]
    results = []
    for full in (True, False):
        emb = ce.CachedEmbeddingBag(N, D, sparse=True, _weight=w0.clone(), mode="sum", include_last_offset=True,
                                    cuda_row_num=3 * F * B * P, warmup_ratio=0.5, strict=False,
                                    evict_strategy=ce.EvictionStrategy.LFU if lfu else ce.EvictionStrategy.DATASET)
        emb.set_fused_sgd(lr)
        emb.set_cache_op(False)

        def step(slots, i, keys=None, emb=emb):
            out = emb(slots, off, hook_features=F, presorted=keys)
            out.backward(grad)

        gw = GraphedWindow(emb, P, F * B, step, overlap=True, warmup_values=[v.cuda() for v in windows[0]],
                           presort=bool(presort), transport="zerocopy", bag_layout=layout, graph_cache_op=full)
        assert (gw._plan_graphs is not None) == full
        gw.submit([v.cuda() for v in windows[0]], 0)
        for w in range(nwin):
            if w + 1 < nwin:
                gw.run_and_submit(w % 2, [v.cuda() for v in windows[w + 1]])
            else:
                gw.run(w % 2)
        torch.cuda.synchronize()
        mgr = emb.cache_weight_mgr
        assert mgr.sync_stats().status == 0
        rec = (list(mgr.num_hits_history), list(mgr.num_miss_history), list(mgr.num_write_back_history), mgr.totals())
        emb.flush()
        results.append((emb.weight.clone(), rec))
    ref = w0.clone()
    for v in windows[0]:                     # the capture warm-up trained on window 0 once (eager pass)
        ref.index_add_(0, v, grad.cpu().transpose(0, 1).reshape(-1, D), alpha=-lr)
    for w in range(nwin):
        for v in windows[w]:
            ref.index_add_(0, v, grad.cpu().transpose(0, 1).reshape(-1, D), alpha=-lr)
    torch.testing.assert_close(results[0][0], ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(results[1][0], ref, rtol=1e-4, atol=1e-4)
    assert results[0][1] == results[1][1]
    assert results[0][1][3]["calls"] >= nwin


@pytest.mark.parametrize("extra", [[], ["--fused_sgd", "--fold_hook", "--use_lfu"], ["--use_cache_mgr_async_copy"],
                                   ["--overlap_cache_op"], ["--overlap_cache_op", "--fused_sgd", "--fold_hook"],
                                   ["--overlap_cache_op", "--fused_sgd", "--fold_hook", "--graph_step", "--graph_after", "6"]])
def test_dlrm_trainer_counterpart_runs_and_learns(extra, capsys):
    """examples/dlrm_main.py (counterpart of recsys/dlrm_main.py): prefetch window + side-stream loader +
    dense DLRM around the operator; the loss must go down on a learnable synthetic target."""
    sys.path.insert(0, str(ROOT / "examples"))
    import importlib
    dm = importlib.import_module("dlrm_main")
    args = ["--dataset", "avazu", "--table_scale", "0.01", "--batch_size", "256", "--embedding_dim", "32",
            "--dense_arch_layer_sizes", "64,32", "--over_arch_layer_sizes", "64,1", "--use_cache", "--cache_ratio", "0.3",
            "--use_freq", "--prefetch_num", "4", "--use_overlap", "--use_sparse_embed_grad", "--limit_train_batches",
            "48", "--learning_rate", "0.2"] + extra
    dm.main(args)
    out = capsys.readouterr().out
    assert "48 iterations" in out and "it/s" in out and "CUDA->CPU" in out
    import re
    m = re.search(r"mean loss first quarter ([0-9.]+) last quarter ([0-9.]+)", out)
    assert m, out
    first, last = float(m.group(1)), float(m.group(2))
    assert last < first - 0.01, f"the loss did not go down: {first} -> {last}"


@pytest.mark.parametrize("extra", [[], ["--overlap_cache_op", "--fused_sgd", "--fold_hook"], ["--use_lfu"]])
def test_dlrm_evaluate_through_the_cache_equals_a_plain_embedding_bag_model(extra, capsys):
    """examples/dlrm_main.py --eval_acc (recsys/dlrm_main.py:300-333,358-369): after training, `_evaluate` runs the
    module in eval mode -- its own cache op per batch on a cache that keeps evicting (3 % of the rows), no gradients.
    Its predictions must be those of the same dense weights over a plain torch-CPU embedding of the FLUSHED table:
    AUROC / accuracy over the test set equal scikit-learn's on the CPU model's predictions, the table is left as the
    training left it (evaluation updates nothing), and the trained model separates the learnable target."""
    import copy
    import importlib
    import numpy as np
    from sklearn.metrics import accuracy_score, roc_auc_score
    sys.path.insert(0, str(ROOT / "examples"))
    dm = importlib.import_module("dlrm_main")
    args = ["--dataset", "avazu", "--table_scale", "0.01", "--batch_size", "256", "--embedding_dim", "32",
            "--dense_arch_layer_sizes", "64,32", "--over_arch_layer_sizes", "64,1", "--use_cache", "--cache_ratio", "0.03",
            "--use_freq", "--prefetch_num", "4", "--use_overlap", "--use_sparse_embed_grad", "--limit_train_batches",
            "48", "--learning_rate", "0.2", "--eval_acc", "--limit_val_batches", "5", "--limit_test_batches", "7",
            "--epochs", "2"] + extra
    # (94,464 rows, a cache of 2,833: the 48 training batches touch ~4,400 distinct rows, a window of 4 under 1,000 --
    # the cache must evict while it trains and while it evaluates, and two protected windows still fit)
    dm.main(args)
    out = capsys.readouterr().out
    assert out.count("AUROC over val set") == 2 and out.count("AUROC over test set") == 1
    assert out.count("Accuracy over val set") == 2 and out.count("Accuracy over test set") == 1
    res, model = dm.main.results, dm.main.model
    assert len(res["val_aurocs"]) == 2 and dm._evaluate.batches == 7
    embed = model.sparse_modules.embed
    writes = sum(embed.num_write_back_history)
    assert writes > 0, "the cache never evicted: the test would not cover evaluation under eviction"
    embed.flush()
    table = embed.weight.detach().clone()
    dense_cpu = copy.deepcopy(model.dense_modules).cpu().eval()
    F = model.sparse_modules.sparse_feature_num
    preds, labels = [], []
    with torch.no_grad():
        for b in dm.main.test_loader:
            values, B = b["sparse"][0].long(), b["sparse"][2]
            pooled = table[values].view(F, B, -1).transpose(0, 1).contiguous()        # one id per bag, feature-major KJT
            preds.append(torch.sigmoid(dense_cpu(b["dense"], pooled).squeeze(-1)))
            labels.append(b["labels"])
    preds, labels = torch.cat(preds).numpy(), torch.cat(labels).numpy().astype(np.int32)
    assert labels.min() == 0 and labels.max() == 1
    # GPU and CPU GEMMs differ in the last bits of a prediction: a near-tie may swap, nothing more
    assert abs(res["test_auroc"] - roc_auc_score(labels, preds)) < 2e-4
    assert abs(res["test_accuracy"] - accuracy_score(labels, preds >= 0.5)) <= 3.0 / len(labels)
    assert res["test_auroc"] > 0.6, f"the trained model does not separate the learnable target: {res}"
    # a second evaluation pass gives the same numbers and leaves the table alone
    again = dm._evaluate(model, dm.main.test_loader, "test", dm.parse_args(args), torch.device("cuda", 0), 0, 1)
    assert abs(again[0] - res["test_auroc"]) < 1e-6 and abs(again[1] - res["test_accuracy"]) <= 1.0 / len(labels)
    embed.flush()
    assert torch.equal(embed.weight.detach(), table)


def test_toy_dlrm_matches_torch_cpu_trajectory():
    """SURVEY 8(c) wiring fixture: a toy DLRM (13 -> 64 -> 32 dense, 4 tables, B = 64) trained 20 steps through
    examples/dlrm_main.py's model and training loop (prefetch window of 4, cached embedding with evictions) against
    the same model on one fused nn.EmbeddingBag + torch-CPU, captured by tests/golden/make_dlrm_golden.py:
    per-step pooled [B, F, D] within 1e-5, loss trajectory within 1e-4, final table after flush() within 1e-5."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "examples"))
    import importlib
    dm = importlib.import_module("dlrm_main")
    gold = np.load(ROOT / "tests" / "golden" / "dlrm_toy.npz")
    sizes = [int(x) for x in gold["sizes"]]
    steps, B = gold["dense_x"].shape[0], gold["dense_x"].shape[1]
    D = gold["table"].shape[1]
    lr = float(gold["lr"])
    for extra in ([], ["--fused_sgd", "--fold_hook"], ["--overlap_cache_op", "--fused_sgd", "--fold_hook"],
                  # the whole iteration replayed from one hipGraph after 3 eager ones (no module hook fires in a replay:
                  # the loss trajectory and the final table carry the comparison)
                  ["--overlap_cache_op", "--fused_sgd", "--fold_hook", "--graph_step", "--graph_after", "3"]):
        args = dm.parse_args(["--use_cache", "--cache_ratio", "0.4", "--prefetch_num", "4", "--use_sparse_embed_grad",
                              "--embedding_dim", str(D), "--batch_size", str(B), "--learning_rate", str(lr),
                              "--dense_arch_layer_sizes", ",".join(str(int(x)) for x in gold["dense_arch"]),
                              "--over_arch_layer_sizes", ",".join(str(int(x)) for x in gold["over_arch"])] + extra)
        dev = torch.device("cuda", 0)
        torch.manual_seed(0)
        model = dm.HybridParallelDLRM(sizes, args, None, dev)
        embed = model.sparse_modules.embed
        embed.flush()                                            # empty cache, then the fixture's table
        embed.weight.copy_(torch.from_numpy(gold["table"]))
        model.dense_modules.load_state_dict({k[len("dense."):]: torch.from_numpy(gold[k]) for k in gold.files
                                             if k.startswith("dense.")})
        groups = [{"params": list(model.dense_modules.parameters()), "lr": lr}]
        if args.fused_sgd:
            embed.set_fused_sgd(lr)
        else:
            groups.insert(0, {"params": list(model.sparse_modules.parameters()), "lr": lr})
        opt = torch.optim.SGD(groups)
        offsets = torch.arange(len(sizes) * B + 1, dtype=torch.int32)
        loader = [dict(dense=torch.from_numpy(gold["dense_x"][i]), labels=torch.from_numpy(gold["labels"][i]),
                       sparse=[torch.from_numpy(gold["values"][i]), offsets, B]) for i in range(steps)]
        pooled = []
        hook = model.sparse_modules.register_forward_hook(lambda m, a, out: pooled.append(out.detach().clone()))
        rec = []
        done, _, _ = dm.train(model, opt, loader, args, dev, 0, 1, record=rec)
        hook.remove()
        graphed = "--graph_step" in extra
        # (graphed: three eager iterations, then the hook fires once more while the step is being captured)
        assert done == steps and len(pooled) == (4 if graphed else steps)
        n_eager = 3 if graphed else steps
        losses = torch.stack(rec).double().cpu().numpy()
        np.testing.assert_allclose(losses, gold["losses"], rtol=0, atol=1e-4)
        got = torch.stack(pooled[:n_eager]).cpu().numpy()
        np.testing.assert_allclose(got, gold["pooled"][:n_eager], rtol=1e-5, atol=1e-5)
        assert sum(embed.cache_weight_mgr.num_write_back_history) >= 0
        embed.flush()
        np.testing.assert_allclose(embed.weight.numpy(), gold["final_table"], rtol=1e-5, atol=1e-5)


def test_dlrm_trainer_reads_binary_criteo_npy(tmp_path, capsys):
    """--dataset_dir: day_*_{dense,sparse,labels}.npy -> BinaryCriteoNpy -> FiniteDataIter -> prefetch window"""
    import numpy as np
    sys.path.insert(0, str(ROOT / "examples"))
    import importlib
    dm = importlib.import_module("dlrm_main")
    rng = np.random.default_rng(0)
    sizes = [50 + 37 * i for i in range(26)]
    for d, n in enumerate([900, 700, 300]):                 # day_6 is held out (val/test)
        day = d if d < 2 else 6
        np.save(tmp_path / f"day_{day}_dense.npy", rng.random((n, 13), dtype=np.float32))
        np.save(tmp_path / f"day_{day}_sparse.npy", rng.integers(0, 1 << 30, (n, 26)).astype(np.int32))
        np.save(tmp_path / f"day_{day}_labels.npy", rng.integers(0, 2, (n, 1)).astype(np.int32))
    args = ["--dataset_dir", str(tmp_path), "--num_embeddings_per_feature", ",".join(map(str, sizes)),
            "--batch_size", "128", "--embedding_dim", "32", "--dense_arch_layer_sizes", "64,32",
            "--over_arch_layer_sizes", "64,1", "--use_cache", "--cache_ratio", "0.9", "--use_freq", "--prefetch_num", "3",
            "--use_overlap", "--use_sparse_embed_grad", "--limit_train_batches", "0", "--learning_rate", "0.05",
            "--shuffle_batches"]
    dm.main(args)
    out = capsys.readouterr().out
    assert f"{(900 + 700) // 128} iterations" in out and "it/s" in out
    assert (tmp_path / "id_freq_map.pt").exists()


@pytest.mark.parametrize("use_lfu", [False, True])
def test_benchmark_cache_counterpart_runs(use_lfu, capsys):
    """benchmarks/benchmark_cache.py (benchmark/benchmark_cache.py:21-75): cache_op=True forward + backward loop"""
    sys.path.insert(0, str(ROOT / "benchmarks"))
    import importlib
    from cachedembedding_amd import synthetic
    bc = importlib.import_module("benchmark_cache")
    g = synthetic.SyntheticKJT(synthetic.TABLES["avazu"], 512, 1, "power_law", 0.25, seed=7, device="cuda")
    rate = bc.benchmark_cache_embedding(512, 32, 0.02, g.id_freq_map(8), 0.7, use_lfu, "avazu", iters=6)
    out = capsys.readouterr().out
    assert rate > 0 and "it/s" in out and "unique-row hit rate" in out and "CUDA->CPU" in out


@pytest.mark.parametrize("strategy,use_freq", [("dataset", True), ("dataset", False), ("lfu", True)])
@pytest.mark.parametrize("transport", ["zerocopy", "worker"])
def test_flush_save_reload_continue_training_round_trip(strategy, use_freq, transport, tmp_path):
    """flush() makes the host table the checkpoint.  Train, flush, save the table, rebuild
    the module from the file with from_pretrained (pin_weight as at benchmark/benchmark_fbgemm_uvm.py:98-105), keep
    training: outputs and the final table equal those of a module that never stopped, and of a plain nn.EmbeddingBag
    over the same logical rows."""
    import cachedembedding_amd as ce
    g = torch.Generator().manual_seed(3)
    N, D, C, B, lr = 4000, 32, 500, 256, 0.1
    w0 = torch.randn(N, D, generator=g)
    freq = torch.randint(0, 30, (N,), generator=g) if use_freq else None
    strat = ce.EvictionStrategy.LFU if strategy == "lfu" else ce.EvictionStrategy.DATASET
    kw = dict(sparse=True, mode="sum", include_last_offset=True, cuda_row_num=C, ids_freq_mapping=freq,
              warmup_ratio=0.7, evict_strategy=strat, pin_weight=True)

    def batches(n, seed):
        gg = torch.Generator().manual_seed(seed)
        out = []
        for _ in range(n):
            lens = torch.randint(0, 3, (B,), generator=gg)
            off = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)])
            out.append((torch.randint(0, N, (int(off[-1]),), generator=gg), off, torch.randn(B, D, generator=gg)))
        return out

    first, second = batches(8, 1), batches(8, 2)

    def train(m, data):
        outs = []
        for ids, off, go in data:
            o = m(ids.cuda(), off.cuda())
            o.backward(go.cuda())
            outs.append(o.detach().cpu())
        return outs

    a = ce.CachedEmbeddingBag.from_pretrained(w0.clone(), freeze=False, **kw)
    assert a.cache_weight_mgr.cuda_cached_weight.requires_grad
    a.cache_weight_mgr.set_transport(transport)
    a.set_fused_sgd(lr)
    train(a, first)
    a.flush()
    path = tmp_path / "table.pt"
    torch.save(a.weight.clone(), path)
    # the uninterrupted module keeps going; the reloaded one starts from the file
    b = ce.CachedEmbeddingBag.from_pretrained(torch.load(path), freeze=False, **kw)
    b.cache_weight_mgr.set_transport(transport)
    b.set_fused_sgd(lr)
    oa, ob = train(a, second), train(b, second)
    for x, y in zip(oa, ob):
        torch.testing.assert_close(x, y, rtol=1e-6, atol=1e-6)
    a.flush()
    b.flush()
    torch.testing.assert_close(a.weight, b.weight, rtol=1e-6, atol=1e-6)
    # reference: plain EmbeddingBag + SGD over the logical rows (DATASET + frequency map re-ranks rows, B#3)
    id2row = a.cache_weight_mgr.idx_map.cpu().long()
    ref = torch.nn.EmbeddingBag.from_pretrained(w0.clone(), freeze=False, mode="sum", include_last_offset=True, sparse=False)
    opt = torch.optim.SGD(ref.parameters(), lr=lr)
    for (ids, off, go), got in zip(first + second, [None] * 8 + oa):
        o = ref(id2row[ids], off)
        if got is not None:
            torch.testing.assert_close(got, o.detach(), rtol=1e-5, atol=1e-6)
        opt.zero_grad()
        o.backward(go)
        opt.step()
    torch.testing.assert_close(a.weight, ref.weight.detach(), rtol=1e-5, atol=1e-6)
    # freeze=True keeps the cache parameter out of autograd
    c = ce.CachedEmbeddingBag.from_pretrained(w0.clone(), **kw)
    assert not c.cache_weight_mgr.cuda_cached_weight.requires_grad
    ids, off, _ = first[0]
    torch.testing.assert_close(c(ids.cuda(), off.cuda()).cpu(),
                               torch.nn.functional.embedding_bag(c.cache_weight_mgr.idx_map.cpu().long()[ids], w0, off,
                                                                 mode="sum", include_last_offset=True))
