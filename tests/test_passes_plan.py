"""Host logic of the memory plan (megahit_amd/passes.py): contiguous bucket ranges under an item budget."""
import numpy as np

from megahit_amd import passes


def test_plan_ranges_cover_all_buckets_within_budget():
    rng = np.random.default_rng(5)
    hist = rng.integers(0, 50, size=65536).astype(np.uint64)
    hist[1234] = 900  # one bucket above the budget gets a range of its own
    ranges = passes.plan_ranges(hist, 500)
    assert ranges[0][0] == 0 and ranges[-1][1] == 65536
    for (lo, hi, n), nxt in zip(ranges, ranges[1:] + [None]):
        assert lo < hi and n == int(hist[lo:hi].sum())
        assert n <= 500 or hi - lo == 1
        if nxt is not None:
            assert nxt[0] == hi
    assert sum(r[2] for r in ranges) == int(hist.sum())
    assert any(r[0] == 1234 and r[1] == 1235 for r in ranges)


def test_plan_ranges_equal_passes_and_single_pass():
    hist = np.full(65536, 10, dtype=np.uint64)
    assert passes.plan_ranges(hist, 10 ** 9) == [(0, 65536, 655360)]
    r = passes.plan_ranges(hist, -4)
    assert len(r) == 4 and max(x[2] for x in r) - min(x[2] for x in r) <= 10
    assert passes.plan_ranges(np.zeros(65536, dtype=np.uint64), 5) == [(0, 65536, 0)]
    assert passes.items_budget(3 * 16 * 1000 / 0.8, 16) == 1000
