"""Host logic of the prefetch-window objects that needs no GPU: the arrangement trial's state machine (with stand-in
events) and the copy-free concatenation of a window's ids."""
import torch

from cachedembedding_amd import pipeline as pl


class _FakeEvent:
    clock = [0.0]
    made = [0]

    def __init__(self, enable_timing=True):
        self.at = None
        _FakeEvent.made[0] += 1

    def record(self, stream=None):
        self.at = _FakeEvent.clock[0]

    def query(self):
        return True

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return other.at - self.at


def _run_trial(monkeypatch, cost, windows, **kw):
    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    _FakeEvent.clock[0] = 0.0
    tr = pl.ArrangementTrial(8, **kw)
    modes = []
    for w in range(windows):
        m = tr.mode
        c = cost[m]
        _FakeEvent.clock[0] += c(w) if callable(c) else c
        modes.append(m)
        tr.window_done(None)
    return tr, modes


def test_trial_alternates_blocks_then_keeps_the_faster_arrangement(monkeypatch):
    tr, modes = _run_trial(monkeypatch, {"interleaved": 1.3, "overlap": 1.1}, 40, block_windows=4, rounds=2, settle=1,
                           retrial_every=0)
    # first block: block_windows + 1 windows (the first event only opens it), then blocks of 4 in turn
    assert "".join(m[0] for m in modes[:17]) == "iiiiiooooiiiioooo"
    assert tr.decided == "overlap" and set(modes[18:]) == {"overlap"} and tr.trials == 1
    rep = tr.report()
    assert rep["mode"] == "overlap" and rep["trial_ms_per_window"] == {"overlap": [1.1, 1.1], "interleaved": [1.3, 1.3]}


def test_trial_records_two_events_per_block_not_one_per_window(monkeypatch):
    """a mark per window cost a prefetch_num = 1 pipeline whose launch thread is the bottleneck 10 us of every step (and
    the trial then measured its own marks): only the marks a block reads are recorded -- behind its `settle`-th window
    and behind its last one"""
    _FakeEvent.made[0] = 0
    tr, modes = _run_trial(monkeypatch, {"interleaved": 1.0, "overlap": 1.2}, 700, block_windows=100, rounds=3, settle=4,
                           retrial_every=0)
    assert tr.decided == "interleaved" and tr.trials == 1
    assert _FakeEvent.made[0] == 2 * 6                     # six blocks, two marks each -- not 600
    ms = tr.history[0]["ms_per_window"]
    assert ms == {"overlap": [1.2, 1.2, 1.2], "interleaved": [1.0, 1.0, 1.0]}


def test_trial_with_three_rounds_decides_by_the_median_block(monkeypatch):
    # one outlier block of the steady arrangement (windows 18-21: its second block) must not hand the verdict to the other
    def il(w):
        return 3.0 if 9 <= w < 13 else 1.0
    tr, modes = _run_trial(monkeypatch, {"interleaved": il, "overlap": 1.2}, 40, block_windows=4, rounds=3, settle=1,
                           retrial_every=0)
    ms = tr.history[0]["ms_per_window"]
    assert len(ms["interleaved"]) == 3 and max(ms["interleaved"]) > 2.0
    assert tr.decided == "interleaved"


def test_trial_runs_again_after_retrial_every_windows(monkeypatch):
    tr, modes = _run_trial(monkeypatch, {"interleaved": 1.0, "overlap": 1.5}, 60, block_windows=3, rounds=1, settle=0,
                           retrial_every=10)
    assert tr.trials >= 2 and tr.decided in (None, "interleaved")
    assert modes.count("overlap") >= 6          # two trials' overlap blocks


def test_reset_block_discards_a_partly_trained_block(monkeypatch):
    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    tr = pl.ArrangementTrial(8, block_windows=3, rounds=1, settle=0, retrial_every=0)
    tr.window_done(None)
    tr.window_done(None)
    tr.reset_block()                               # a window trained in part: the block starts over
    for _ in range(3):
        assert tr.mode == "interleaved"
        tr.window_done(None)
    assert tr.mode == "interleaved"
    tr.window_done(None)
    assert tr.mode == "overlap"


def test_cat_window_returns_a_view_for_consecutive_pieces_and_a_copy_otherwise():
    w = torch.arange(24).view(4, 6)
    v = pl.cat_window([w[i] for i in range(4)])
    assert v.data_ptr() == w.data_ptr() and torch.equal(v, w.view(-1))
    v = pl.cat_window([w[1], w[2]])
    assert v.data_ptr() == w[1].data_ptr() and torch.equal(v, w[1:3].reshape(-1))
    v = pl.cat_window([w[0], w[2]])                                    # a gap: copied
    assert torch.equal(v, torch.cat([w[0], w[2]])) and v.data_ptr() != w.data_ptr()
    v = pl.cat_window([w[0].clone(), w[1].clone()])                    # two storages: copied
    assert torch.equal(v, w[:2].reshape(-1))
    assert pl.cat_window([w[3]]).data_ptr() == w[3].data_ptr()
    i32 = w.int()
    assert torch.equal(pl.cat_window([i32[0], i32[1]]), i32[:2].reshape(-1))


def test_arrangement_arguments_are_checked():
    import pytest
    with pytest.raises(ValueError):
        pl._resolve_arrangement("sideways", True)
    with pytest.raises(ValueError):
        pl._resolve_arrangement("overlap", False)
    assert pl._resolve_arrangement(None, True) == pl.DEFAULT_ARRANGEMENT
    assert pl._resolve_arrangement(None, False) is None
