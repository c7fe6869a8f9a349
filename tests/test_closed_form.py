"""oracle/closed_form.py (the end-of-run checker of bench.py and of the full-shape pipeline test) on the CPU:
it must accept what torch's own SGD leaves in a table and reject a table with one lost or one doubled update."""
import numpy as np
import pytest
import torch

from oracle.closed_form import SgdLedger


def _train(N, D, steps, n, lr, seed, idx_map=None, hot=0):
    g = torch.Generator().manual_seed(seed)
    w0 = (torch.rand(N, D, generator=g) - 0.5) * 0.1
    w = w0.clone()
    led = SgdLedger(N, D, lr, idx_map)
    batches = []
    for s in range(steps):
        ids = torch.randint(0, N, (n,), generator=g)
        if hot:
            ids[: n // 2] = torch.randint(0, hot, (n // 2,), generator=g)      # a few rows take half of the lookups
        grad = torch.randn(n, D, generator=g) * 0.01
        rows = ids if idx_map is None else idx_map[ids].long()
        # the reference's arithmetic: coalesce the step's gradient (fp32), then SGD.step
        sp = torch.sparse_coo_tensor(rows.view(1, -1), grad, (N, D)).coalesce()
        w.index_add_(0, sp.indices()[0], sp.values(), alpha=-lr)
        led.record(ids, grad)
        batches.append((ids, grad))
    return w0, w, led, batches


@pytest.mark.parametrize("with_map", [False, True])
def test_accepts_torch_sgd(with_map):
    N, D = 5000, 16
    idx_map = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(torch.int32) if with_map else None
    w0, w, led, _ = _train(N, D, steps=12, n=700, lr=0.5, seed=3, idx_map=idx_map, hot=3)
    res = led.check(lambda r: w0[r], lambda r: w[r], hot_rows=64, untouched_sample=4096)
    assert res["bound_violations"] == 0 and res["untouched_mismatch"] == 0
    assert res["steps"] == 12 and res["lookups"] == 12 * 700
    assert res["hot_torch_fp32_max_err_over_bound"] <= 1.0 and res["hot_table_max_err_over_bound"] <= 1.0
    assert res["hot_table_vs_torch_fp32_max_diff_rel_to_row_max"] < 1e-5
    # the same through several row chunks
    res2 = led.check(lambda r: w0[r], lambda r: w[r], hot_rows=0, untouched_sample=0, max_chunk_rows=777)
    assert res2["chunks"] > 1 and res2["bound_violations"] == 0 and res2["rows"] == res["rows"]


def test_rejects_a_lost_and_a_doubled_update():
    N, D, lr = 4000, 8, 0.5
    w0, w, led, batches = _train(N, D, steps=6, n=500, lr=lr, seed=5)
    ids, grad = batches[2]
    row = int(ids[17])
    lost = w.clone()
    lost[row] += lr * grad[17]                     # as if lookup 17 of step 2 had never been applied
    res = led.check(lambda r: w0[r], lambda r: lost[r], hot_rows=0, untouched_sample=0)
    assert res["bound_violations"] > 0 and res["rows_violating"] == 1 and res["worst"]["row"] == row
    twice = w.clone()
    twice[row] -= lr * grad[17]
    assert led.check(lambda r: w0[r], lambda r: twice[r], hot_rows=0, untouched_sample=0)["rows_violating"] == 1


def test_rejects_a_stray_write_to_an_untouched_row():
    N, D = 3000, 8
    w0, w, led, batches = _train(N, D, steps=2, n=100, lr=0.1, seed=9)
    touched = set(torch.cat([b[0] for b in batches]).tolist())
    victim = next(r for r in range(N) if r not in touched)
    w = w.clone()
    w[victim, 3] = float(np.nextafter(np.float32(w[victim, 3]), np.float32(1)))       # one ulp
    res = led.check(lambda r: w0[r], lambda r: w[r], hot_rows=0, untouched_sample=1 << 16)
    assert res["bound_violations"] == 0 and res["untouched_mismatch"] >= 1


def test_padding_ids_are_no_lookups():
    N, D = 1000, 4
    led = SgdLedger(N, D, 1.0)
    w0 = torch.zeros(N, D)
    ids = torch.tensor([5, -1, 7, 5])
    grad = torch.ones(4, D)
    led.record(ids, grad)
    w = w0.clone()
    w[5] -= 2
    w[7] -= 1
    res = led.check(lambda r: w0[r], lambda r: w[r], hot_rows=4, untouched_sample=512)
    assert res["rows"] == 2 and res["lookups"] == 3 and res["bound_violations"] == 0 and res["untouched_mismatch"] == 0


def test_single_lookup_rows_are_checked_bit_for_bit():
    """a row that was looked up exactly once saw ONE fp32 update: one ulp off is a violation"""
    N, D, lr = 6000, 8, 0.5
    w0, w, led, batches = _train(N, D, steps=3, n=300, lr=lr, seed=21)
    res = led.check(lambda r: w0[r], lambda r: w[r], hot_rows=0, untouched_sample=0)
    assert res["single_lookup_rows"] > 500 and res["single_lookup_mismatch"] == 0 and res["bound_violations"] == 0
    assert res["cold_rows"] >= res["single_lookup_rows"] and res["max_rel_err_cold"] < 1e-6
    counts = torch.bincount(torch.cat([b[0] for b in batches]), minlength=N)
    row = int((counts == 1).nonzero()[0])
    off = w.clone()
    off[row, 5] = float(np.nextafter(np.float32(off[row, 5]), np.float32(10)))       # one ulp
    res = led.check(lambda r: w0[r], lambda r: off[r], hot_rows=0, untouched_sample=0)
    assert res["single_lookup_mismatch"] == 1 and res["rows_violating"] == 1 and res["bound_violations"] >= 1


def test_cold_rows_are_held_to_1e_5_relative_without_an_absolute_floor():
    """a row with 2-4 lookups at |w| ~ 1e-3: an error of 1e-4 relative (1e-7 absolute -- far inside the round-4 floor
    of 2e-6) is rejected, an error of 1e-6 relative is accepted"""
    N, D, lr = 6000, 8, 0.5
    g = torch.Generator().manual_seed(4)
    w0 = (torch.rand(N, D, generator=g) - 0.5) * 2e-3
    ids = torch.randint(0, N, (900,), generator=g)
    grad = torch.randn(900, D, generator=g) * 1e-3
    w = w0.clone().index_add_(0, ids, grad, alpha=-lr)
    led = SgdLedger(N, D, lr)
    led.record(ids, grad)
    counts = torch.bincount(ids, minlength=N)
    row = int((counts == 2).nonzero()[0])
    assert led.check(lambda r: w0[r], lambda r: w[r], hot_rows=0, untouched_sample=0)["bound_violations"] == 0
    for rel, violates in ((1e-6, False), (1e-4, True)):
        off = w.clone()
        off[row, 2] = float(np.float32(off[row, 2]) * np.float32(1.0 + rel))
        res = led.check(lambda r: w0[r], lambda r: off[r], hot_rows=0, untouched_sample=0)
        assert (res["rows_violating"] == 1) == violates, (rel, res)
