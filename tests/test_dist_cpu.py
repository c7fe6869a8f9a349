"""world_size-2 gloo tests (CPU) of the multi-GPU exchange logic: row-wise sharded lookup
(bucketise -> counts/ids all-to-all -> owner cache op -> rows all-to-all -> pool; grads the reverse way),
the KJT all-gather ordering and dual_all_to_all.  Local compute is the oracle-backed TorchShardOps."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q, args)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    res = []
    while not q.empty():
        res.append(q.get())
    assert len(res) == world, f"only {len(res)} of {world} ranks reported"
    for r in res:
        assert r[1] == "ok", r
    return res


def _entry(fn, rank, world, port, q, args):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        fn(rank, world, *args)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc()))


def _rowwise(rank, world, mode, hook):
    from cachedembedding_amd.parallel import RowwiseExchange
    from shard_ops_ref import TorchShardOps
    torch.manual_seed(0)                      # same global problem on every rank
    N, D, F, B_loc, P, lr = 997, 16, 3, 8, 2, 0.5
    w_full = torch.randn(N, D)
    idx_map = torch.randperm(N).int()         # the frequency re-rank (any permutation)
    shard = w_full[rank::world].numpy().copy()
    ops = TorchShardOps(shard, 200, idx_map, world, rank)
    ex = RowwiseExchange(ops)
    g = torch.Generator().manual_seed(100 + rank)      # different LOCAL batch per rank
    if hook:
        lens = torch.ones(F * B_loc, dtype=torch.long)
    else:
        lens = torch.randint(0, 4, (F * B_loc,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)])
    ids_list = [torch.randint(0, N, (int(offsets[-1]),), generator=g) for _ in range(P)]
    plans = ex.plan_window(ids_list)
    ref_w = w_full.clone()                    # global table indexed by re-ranked row
    for ids, plan in zip(ids_list, plans):
        assert sum(plan.send_splits) == plan.n == len(torch.unique(ids)) and plan.recv_rows.numel() == sum(plan.recv_splits)
        rows = ex.fetch_rows(plan)
        out = ops.pool(rows, plan.perm, offsets, None, mode, True, F if hook else 0)
        # expected: plain EmbeddingBag over the full table (row = idx_map[id]); all ranks' updates of the
        # previous batches are already in ref_w because every rank replays every rank's batch below
        exp = torch.nn.functional.embedding_bag(idx_map[ids].long(), ref_w, offsets, mode=mode,
                                                include_last_offset=True)
        if hook:
            exp = exp.view(F, B_loc, D).transpose(0, 1)
        torch.testing.assert_close(out, exp, rtol=1e-5, atol=1e-6)
        go = torch.randn(out.shape, generator=g)
        grows = ops.grad_rows(go, plan.perm, offsets, None, mode, True, F if hook else 0, plan.n)
        ops.owner_update(plan.slots, ex.return_grads(plan, grows), lr)
        # replay every rank's update on the replicated reference table
        packs = [None] * world
        dist.all_gather_object(packs, (ids, go, offsets))
        for pids, pgo, poff in packs:
            wr = ref_w.clone().requires_grad_(True)
            o = torch.nn.functional.embedding_bag(idx_map[pids].long(), wr, poff, mode=mode, include_last_offset=True)
            o.backward(pgo.transpose(0, 1).reshape(-1, D) if hook else pgo)
            ref_w = ref_w - lr * wr.grad
    ops.mgr.flush()
    torch.testing.assert_close(torch.from_numpy(ops.mgr.weight), ref_w[rank::world], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode,hook", [("sum", False), ("mean", False), ("sum", True)])
def test_rowwise_exchange_world2(mode, hook):
    _spawn(_rowwise, 2, mode, hook)


def _kjt(rank, world):
    from cachedembedding_amd.parallel import KJTAllToAll, dual_all_to_all
    K, b = 3, 4
    g = torch.Generator().manual_seed(rank)
    lengths = torch.randint(0, 3, (K * b,), generator=g, dtype=torch.int32)
    values = torch.arange(int(lengths.sum())) + 1000 * rank
    av, al = KJTAllToAll().all_to_all(values, lengths, K)
    # expected ordering [key][rank][sample] (recsys/datasets/utils.py:43-49)
    all_l, all_v = [None] * world, [None] * world
    dist.all_gather_object(all_l, lengths)
    dist.all_gather_object(all_v, values)
    exp_v, exp_l = [], []
    for k in range(K):
        for r in range(world):
            lk = all_l[r].view(K, b)
            start = int(lk[:k].sum())
            exp_v.append(all_v[r][start:start + int(lk[k].sum())])
            exp_l.append(lk[k])
    assert torch.equal(av, torch.cat(exp_v)) and torch.equal(al, torch.cat(exp_l))
    # dual_all_to_all: [B_glob, F, D_loc] -> [B_glob/W, F, D] and its autograd transpose
    Bg, Fq, D = 6, 2, 5
    full = torch.arange(Bg * Fq * D, dtype=torch.float32).view(Bg, Fq, D)
    cols = torch.tensor_split(full, world, dim=2)[rank].clone().requires_grad_(True)
    out = dual_all_to_all(cols, scatter_dim=0, gather_dim=-1)
    assert torch.equal(out.detach(), torch.tensor_split(full, world, dim=0)[rank])
    out.backward(torch.ones_like(out) * (rank + 1))
    exp_g = torch.cat([torch.full((Bg // world, Fq, cols.shape[2]), float(r + 1)) for r in range(world)])
    assert torch.equal(cols.grad, exp_g)


def test_kjt_allgather_and_dual_all_to_all_world2():
    _spawn(_kjt, 2)


def test_get_partition_matches_tensor_split():
    from cachedembedding_amd.parallel import get_partition
    for D in (128, 130, 7):
        for W in (1, 2, 3, 4):
            if D < W:
                continue
            sizes = [t.shape[0] for t in torch.tensor_split(torch.zeros(D), W)]
            off = 0
            for r in range(W):
                lo, hi, _ = get_partition(D, r, W)
                assert (lo, hi) == (off, off + sizes[r])
                off += sizes[r]


class _ClockEvent:
    """stand-in for torch.cuda.Event on a rank-local fake clock (tests/test_pipeline_cpu.py has the single-rank form)"""
    clock = [0.0]

    def __init__(self):
        self.at = None

    def record(self, stream=None):
        self.at = _ClockEvent.clock[0]

    def query(self):
        return True

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return other.at - self.at


def _collective_trial(rank, world):
    """VERDICT r5 #2a: the arrangement trial at W > 1.  Every rank times its own blocks -- here rank 0 finds 'interleaved'
    faster and the others 'overlap' -- but the verdict comes from the MAX over the ranks, reduced in the same window on
    every rank: all of them must run every window in the same arrangement and switch in the same window."""
    from cachedembedding_amd.pipeline import ArrangementTrial

    def reduce_max(values):
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    cost = {"interleaved": 1.0, "overlap": 1.3} if rank == 0 else {"interleaved": 1.6, "overlap": 1.2}
    tr = ArrangementTrial(8, block_windows=4, rounds=3, settle=1, retrial_every=40, reduce_fn=reduce_max, decide_lag=2,
                          event_factory=_ClockEvent)
    modes = []
    for _ in range(120):                       # two trials (retrial_every = 40 windows after the first verdict)
        m = tr.mode
        _ClockEvent.clock[0] += cost[m]
        modes.append(m)
        tr.window_done(None)
    mine = torch.tensor([0 if m == "interleaved" else 1 for m in modes])
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    for g in gathered:
        assert torch.equal(g, mine), "the ranks trained a window in different arrangements"
    # MAX over the ranks: interleaved 1.6, overlap 1.3 per window -> overlap, whatever this rank measured itself
    assert tr.decided == "overlap" and tr.trials == 2
    assert tr.history[-1]["ms_per_window"] == {"overlap": [1.3] * 3, "interleaved": [1.6] * 3}
    first_switch = modes.index("overlap", 6 * 4 + 1)           # the first window after the trial's blocks
    assert all(m == "overlap" for m in modes[first_switch:first_switch + 30])


@pytest.mark.parametrize("world", [2, 3])
def test_arrangement_trial_reaches_one_verdict_on_all_ranks(world):
    _spawn(_collective_trial, world)


def _eval_meter(rank, world):
    sys.path.insert(0, str(ROOT / "examples"))
    import importlib
    from sklearn.metrics import accuracy_score, roc_auc_score
    dm = importlib.import_module("dlrm_main")
    rng = np.random.default_rng(5)                    # the same global problem on every rank ...
    n_per = [700, 123, 400][:world]                   # ... of which the ranks hold shares of different sizes
    labels = rng.integers(0, 2, sum(n_per)).astype(np.int32)
    scores = (np.round(rng.random(sum(n_per)) * 50) / 50 * 0.7 + labels * 0.2).astype(np.float32)   # with ties
    lo = sum(n_per[:rank])
    mine = slice(lo, lo + n_per[rank])
    meter = dm.BinaryMetrics()
    for a in range(mine.start, mine.stop, 97):
        b = min(a + 97, mine.stop)
        meter(torch.from_numpy(scores[a:b]), torch.from_numpy(labels[a:b]))
    auroc, acc = meter.compute()                      # collective: every rank gets the whole set's numbers
    assert abs(auroc - roc_auc_score(labels, scores)) < 1e-12
    assert abs(acc - accuracy_score(labels, scores >= 0.5)) < 1e-12


@pytest.mark.parametrize("world", [2, 3])
def test_evaluation_meter_gathers_every_ranks_share(world):
    """examples/dlrm_main.py::_evaluate at W > 1 (recsys/dlrm_main.py:300-333: torchmetrics syncs across ranks at
    compute()): ranks hold shares of different lengths, all of them report the set's AUROC / accuracy"""
    _spawn(_eval_meter, world)
