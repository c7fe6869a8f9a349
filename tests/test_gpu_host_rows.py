"""Rows of the pinned host table by number (ce_host_fill_uniform_rows, ce_host_rows_gather) and the box probe:
the pieces bench.py's end-of-run verification reads the 91 GB table through."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,D", [(100_003, 128), (5_000, 30), (70_001, 4), (1_000, 200)])
def test_rows_regenerated_from_the_seed_equal_the_filled_table(N, D):
    from cachedembedding_amd import _lib
    from cachedembedding_amd.cache_mgr import HostTable
    lo, hi, seed = -1.0 / N, 1.0 / N, 1024 + D
    table = HostTable.allocate(N, D).fill_uniform_(lo, hi, seed)
    g = torch.Generator().manual_seed(N)
    rows = torch.randint(0, N, (20_000,), generator=g)
    rows[:3] = torch.tensor([0, N - 1, N // 2])
    rd = rows.cuda()
    regen = torch.empty(rows.numel(), D, device="cuda")
    _lib.check(_lib.lib.ce_host_fill_uniform_rows(rd.data_ptr(), rd.numel(), D, lo, hi, seed, regen.data_ptr(),
                                                  _lib.stream_ptr()))
    read = torch.empty(rows.numel(), D, device="cuda")
    _lib.check(_lib.lib.ce_host_rows_gather(table.dev_ptr, N, D, rd.data_ptr(), rd.numel(), read.data_ptr(),
                                            _lib.stream_ptr()))
    want = table.tensor[rows]
    assert torch.equal(read.cpu().view(torch.int32), want.view(torch.int32))          # the gather, bit for bit
    assert torch.equal(regen.cpu().view(torch.int32), want.view(torch.int32))         # the generator, bit for bit
    assert float(want.abs().max()) <= 1.0 / N and float(want.std()) > 0.2 / N


def test_gather_returns_zero_rows_for_rows_outside_the_table():
    from cachedembedding_amd import _lib
    from cachedembedding_amd.cache_mgr import HostTable
    N, D = 4096, 64
    table = HostTable.allocate(N, D).fill_uniform_(-1.0, 1.0, 5)
    rows = torch.tensor([3, -1, N, 10, 2 ** 40], device="cuda")
    out = torch.full((5, D), 7.0, device="cuda")
    _lib.check(_lib.lib.ce_host_rows_gather(table.dev_ptr, N, D, rows.data_ptr(), 5, out.data_ptr(), _lib.stream_ptr()))
    o = out.cpu()
    assert torch.equal(o[0], table.tensor[3]) and torch.equal(o[3], table.tensor[10])
    assert not o[1].any() and not o[2].any() and not o[4].any()


def test_box_probe_reports_plausible_rates():
    from cachedembedding_amd import _lib
    buf = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
    rd, fl = ctypes.c_double(), ctypes.c_double()
    _lib.check(_lib.lib.ce_box_probe(buf.data_ptr(), buf.numel(), 10, ctypes.byref(rd), ctypes.byref(fl),
                                     _lib.stream_ptr()))
    assert 1000.0 < rd.value < 20000.0 and 1000.0 < fl.value < 20000.0           # GB/s on an MI355X (HBM ~8 TB/s)
    with pytest.raises(_lib.CeError):
        _lib.check(_lib.lib.ce_box_probe(buf.data_ptr(), 1024, 10, ctypes.byref(rd), ctypes.byref(fl), _lib.stream_ptr()))
