"""bench.py's id windows (IdWindows): a window that was given back to the allocator is drawn again bit for bit from the
generator state it came from, the self-test on the first window does not disturb the sequence of draws (the workload of
a line is the same ids as ever), and nothing is given back where a re-draw could not be confirmed.  CPU generator here;
on the GPU the same torch.Generator get_state / set_state calls act on the Philox seed + offset."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _gen(seed=1024, uniform_frac=0.0):
    from cachedembedding_amd import synthetic
    sizes = synthetic.scale_tables(synthetic.TABLES["criteo_kaggle"], 0.001)
    return synthetic.SyntheticKJT(sizes, 64, 1, "power_law", 0.25, seed=seed, device="cpu", uniform_frac=uniform_frac)


def test_windows_are_the_uninterrupted_sequence_and_come_back_bit_for_bit():
    import bench
    for uf in (0.0, 0.3):
        P = 4
        ref = _gen(uniform_frac=uf)
        expected = [ref.next_values(P) for _ in range(9)]
        after_ref = ref.next_values(1)
        g = _gen(uniform_frac=uf)
        wins = bench.IdWindows(g, P)
        for _ in range(5):
            wins.draw()
        assert wins.redraw_ok is True
        wins.release(3)
        assert wins[0] is None and wins[2] is None and wins[3] is not None
        for _ in range(4):
            wins.draw()                                   # drawing goes on where it was
        wins.release(7)
        assert all(torch.equal(wins[w], expected[w]) for w in (7, 8))
        state = g.gen.get_state()
        for w in (5, 0, 8, 2, 6, 1, 3, 4, 7):             # any order, resident ones included
            assert torch.equal(wins.again(w), expected[w]), w
        g.gen.set_state(state)                            # what bench.py does around the re-draws
        assert torch.equal(g.next_values(1), after_ref)   # the generator carries on as if nothing had been drawn again


def test_no_selftest_means_nothing_is_confirmed_and_a_broken_generator_keeps_everything():
    import bench
    g = _gen()
    wins = bench.IdWindows(g, 2, selftest=False)
    wins.draw()
    assert wins.redraw_ok is False                        # --no_verify / --keep_windows: bench.py decides by the flag alone

    class NoState:                                        # a generator whose state cannot be read
        def __init__(self, inner):
            self.inner, self.gen = inner, self

        def get_state(self):
            raise RuntimeError("no state")

        def set_state(self, s):
            raise RuntimeError("no state")

        def next_values(self, P):
            return self.inner.next_values(P)
    wins = bench.IdWindows(NoState(_gen()), 2)
    wins.draw()
    wins.draw()
    assert wins.redraw_ok is False and wins[0] is not None and wins[1] is not None

    class Drifting:                                       # a generator that does not repeat itself from a state
        def __init__(self, inner):
            self.inner, self.gen, self.k = inner, inner.gen, 0

        def next_values(self, P):
            self.k += 1
            return self.inner.next_values(P) + (self.k == 2)
    wins = bench.IdWindows(Drifting(_gen()), 2)
    wins.draw()
    assert wins.redraw_ok is False
