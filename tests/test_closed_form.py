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
