"""GPU: the chained-scan pass with unit-wide runs (sort_kernels.h: k_radix_onesweep_u — all tiles of a unit ranked first, the
LDS stage a window sliding over the unit's digit-sorted sequence) orders records exactly as the tile-by-tile pass and as
the oracle's stable sort do: the job of kmlib::kmsort behind SelectSortingFunc (reference src/kmlib/kmsort.h:45-122,
src/sorting/kmsort_selector.cpp:39-63), every record width that takes the chained scan, every form of the digit
(a bit field of key word 0 / 1, the generic form for later words), partial last units, skewed digits, and the engines on
top of it (stage 1 with its generated first pass, stage 2, count) against the oracle."""
import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_count import load, make_reads
from test_gpu_sdbg import check_sdbg

pytestmark = pytest.mark.gpu


# unit-wide runs (the default: the prefix plans of stage 1 rank with an LDS atomic wherever all records of a wavefront
# instruction agree on the bits sorted so far, include/mhx.h: sort_rank_uniform), the tile-by-tile pass, unit-wide runs with
# the match-any ballots everywhere
@pytest.fixture(params=[(1, 1), (0, 1), (1, 0)], ids=["unit_runs", "tile_runs", "unit_runs_ballots"])
def unit_runs(engine, request):
    engine.set_option("sort_unit_runs", request.param[0])
    engine.set_option("sort_rank_uniform", request.param[1])
    engine.set_option("sort_hybrid", 0)  # every key bit by LSD passes
    yield request.param
    engine.set_option("sort_unit_runs", 1)
    engine.set_option("sort_rank_uniform", 1)
    engine.set_option("sort_hybrid", 1)


# (key words, aux words, records): widths 2, 3, 4, 6, 8 words take the chained scan; 6144 / 4096 / 2048 records per unit
@pytest.mark.parametrize("kw,aux,n", [(2, 0, 1500001), (1, 1, 6144 * 5), (2, 1, 1200007), (1, 2, 70001), (3, 1, 900001), (2, 2, 4096 * 3 + 1),
                                      (5, 1, 300001), (4, 2, 2048 * 7 - 1), (7, 1, 250001), (6, 2, 1), (2, 1, 6143), (2, 1, 6145)])
def test_sort_records(engine, unit_runs, kw, aux, n):
    rng = np.random.default_rng(kw * 100 + aux * 10 + n % 97)
    items = rng.integers(0, 2 ** 32, size=(n, kw + aux), dtype=np.uint64).astype(np.uint32)
    dup = rng.integers(0, n, size=n // 3)
    items[: n // 3, :kw] = items[dup, :kw]  # equal keys, aux differs: stability shows
    want = ob.sort_items(items, kw, kmsort=False)
    got = engine.sort_records(items.copy(), kw)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kw,aux", [(2, 0), (2, 1), (3, 1)])
def test_sort_records_skewed_digits(engine, unit_runs, kw, aux):
    """a few digit values take nearly every record (runs far longer than a window), the rest are singletons"""
    n = 400003
    rng = np.random.default_rng(kw + aux)
    items = rng.integers(0, 2 ** 32, size=(n, kw + aux), dtype=np.uint64).astype(np.uint32)
    hot = rng.random(n) < 0.9
    for w in range(kw):
        items[hot, w] = np.where(rng.random(int(hot.sum())) < 0.5, np.uint32(0x07070707), np.uint32(0xC8C8C807))
    want = ob.sort_items(items, kw, kmsort=False)
    assert np.array_equal(engine.sort_records(items.copy(), kw), want)


@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("fixed", 21, 1), ("var", 27, 2), ("lowcomplex", 21, 2), ("fixed", 33, 2)])
def test_read2sdbg_and_count(engine, unit_runs, kind, k, m):
    from megahit_amd import lib
    reads = make_reads(kind, seed=k + m)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    want1 = ob.s1(pkg, k, m, tie_stable=True)
    r1 = engine.read2sdbg_s1(k, m)
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert r1.n_items == want1["n_items"] and np.array_equal(solid, want1["is_solid"][: solid.size])
    want2 = ob.s2(pkg, k, m, want1["is_solid"])
    r2 = engine.read2sdbg_s2(k, m)
    check_sdbg(engine, r2, want2)
    wantc = ob.count(pkg, k, m)
    rc = engine.count(k, m)
    assert np.array_equal(engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, rc.words_per_edge), wantc["edges"])
