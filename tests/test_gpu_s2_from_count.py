"""GPU: stage 2's solid items from a COUNT of the (k+1)-mers (round 6; s2.hip s2_agg_from_count) — min count 1 (stage 1 skipped, every
occurrence solid: main_sdbg_build.cpp:139-147; the meta presets) at k <= 27, and min count >= 2 at k = 23..27 behind stage 1 — against the
oracle's per-occurrence Read2SdbgS2 (reference src/sorting/read_to_sdbg_s2.cpp:271-440,521-614): SdBG bytes, per-bucket tables, W
counts; multiplicities beyond one item's count field (k = 27: 6 bits) and beyond the 65 535 cap; palindromic (k+1)-mers; giant buckets;
reads of several lengths; and the per-occurrence path beside it (s2_agg_from_count = 0), byte for byte the same."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib
from test_gpu_count import load, make_reads
from test_gpu_round3_knobs import fixed_library
from test_gpu_sdbg import check_sdbg
from test_gpu_tile_tails import giant_library

pytestmark = pytest.mark.gpu


def run(engine, reads, k, m, opts=None, expect=True):
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    opts = opts or {}
    try:
        for n, v in opts.items():
            engine.set_option(n, v)
        if m > 1:
            w1 = ob.s1(pkg, k, m, tie_stable=True)
            engine.read2sdbg_s1(k, m)
            want = ob.s2(pkg, k, m, w1["is_solid"])
        else:
            want = ob.s2(pkg, k, 1, None)
        engine.profile(True)
        engine.profile_reset()
        r2 = engine.read2sdbg_s2(k, m)
        stats = engine.profile_get()
        engine.profile(False)
        assert ("s2_edges_to_items" in stats) == expect, sorted(stats)
        check_sdbg(engine, r2, want)
        bytes_a = engine.fetch(lib.BUF_SDBG_BYTES, np.uint8).copy()
        if expect:  # the per-occurrence path on the same input: the same bytes
            engine.set_option("s2_agg_from_count", 0)
            if m > 1:
                engine.read2sdbg_s1(k, m)
            check_sdbg(engine, engine.read2sdbg_s2(k, m), want)
            assert np.array_equal(engine.fetch(lib.BUF_SDBG_BYTES, np.uint8), bytes_a)
    finally:
        engine.profile(False)
        engine.set_option("s2_agg_from_count", 1)
        for n in opts:
            engine.set_option(n, {"s1_giant_min": 262144, "s1_stream_fill": 7168, "s1_var_min_fill": 50}.get(n, 1))


@pytest.mark.parametrize("kind,k", [("pe100", 21), ("repeats100", 21), ("pe100", 27), ("repeats100", 27), ("pe100", 23), ("repeats100", 25), ("short30", 21), ("pe100", 15)])
def test_min_count_1(engine, kind, k):
    run(engine, fixed_library(kind, seed=k + 3), k, 1)


@pytest.mark.parametrize("kind,k,m", [("pe100", 27, 2), ("repeats100", 27, 2), ("pe100", 23, 2), ("repeats100", 25, 3), ("pe100", 26, 2)])
def test_behind_stage_1_at_k_23_to_27(engine, kind, k, m):
    run(engine, fixed_library(kind, seed=k + m), k, m)


@pytest.mark.parametrize("k,m", [(27, 1), (27, 2), (21, 1), (24, 1)])
def test_multiplicities_beyond_the_count_field_and_the_cap(engine, k, m):
    """~10^5 occurrences of poly-A (k+1)-mers: thousands of items of one edge at k = 27 (63 per item), the 65 535 cap; palindromes ((AC)n)"""
    run(engine, giant_library(k + m, n_poly=1500), k, m)


@pytest.mark.parametrize("opts", [dict(s1_giant_min=64), dict(s1_stream_fill=40)], ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()))
def test_with_giant_buckets_and_overflowing_tables(engine, opts):
    run(engine, fixed_library("repeats100", seed=12), 27, 1, opts)


def test_shapes_it_does_not_take(engine):
    run(engine, make_reads("var", 5), 27, 1, expect=False)               # k = 23..27: reads of one length only
    run(engine, fixed_library("pe100", seed=2), 29, 1, expect=False)     # the count's records hold (k+1)-mers up to k = 27
    run(engine, fixed_library("pe100", seed=2), 21, 2, expect=False)     # k <= 22, min count >= 2: stage 1 made the aggregated items itself
