"""GPU: `count` in bucket-range passes (the memory plan; reference src/sorting/base_engine.cpp:176-201 drives ONE KmerCounter through
lv1 bucket ranges, kmer_counter.cpp:158-381) ON THE STAGE-1 DESIGN (round 6): the bucket filter sits inside the histogram pre-pass and
the generating first sort pass (CountGenT<true> / CountGenVarT<true>), first_0_out / last_0_in / the histogram accumulate over the
passes.  Asserted: every pass ran the streaming form (its kernels are in the profile, the tile path's are not), the lv1 histogram came
from the packed reads, and the concatenated result equals the oracle's single run — fixed- and variable-length libraries, low-complexity
reads, position tags, tables that overflow, and a pass whose streaming form gives up (the tile path then continues the accumulation)."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib, passes
from test_gpu_count import load, make_reads
from test_gpu_round3_knobs import fixed_library

pytestmark = pytest.mark.gpu

RESET = dict(count_stream=1, s1_stream_fill=7168, s1_pos_bits=0, s1_stream_bits=0, s1_stream_sub0=-1, s1_stream_probes=1024, s1_var_min_fill=50,
             s1_filter_in_gen=1)


def run_passes(engine, pkg, k, m, n_passes, opts, per_pass_opts=None):
    load(engine, pkg)
    try:
        for n, v in opts.items():
            engine.set_option(n, v)
        engine.profile(True)
        engine.profile_reset()
        hist = engine.bucket_histogram(lib.STAGE_COUNT, k, m)
        stats_h = engine.profile_get()
        ranges = passes.plan_ranges(hist, -n_passes)
        edges, bcount, per_pass = [], np.zeros(65536, dtype=np.uint64), []
        try:
            for i, (lo, hi, n) in enumerate(ranges):
                for nm, v in (per_pass_opts or {}).get(i, {}).items():
                    engine.set_option(nm, v)
                engine.set_bucket_filter(passes._mask(lo, hi), n, 0, accumulate=i > 0)
                engine.profile_reset()
                r = engine.count(k, m)
                per_pass.append((engine.profile_get(), r))
                edges.append(engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge))
                bcount += engine.fetch(lib.BUF_BUCKET_COUNT, np.uint64)
        finally:
            engine.set_bucket_filter(None)
        got = dict(edges=np.concatenate(edges), bucket_count=bcount, hist=engine.fetch(lib.BUF_MUL_HIST, np.int64),
                   first_0_out=engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), last_0_in=engine.fetch(lib.BUF_LAST_0_IN, np.uint32))
    finally:
        engine.profile(False)
        for n, v in RESET.items():
            engine.set_option(n, v)
    return got, stats_h, per_pass, ranges


def check(got, want):
    assert got["edges"].shape == want["edges"].shape and np.array_equal(got["edges"], want["edges"])
    assert np.array_equal(got["bucket_count"], want["bucket_count"])
    assert np.array_equal(got["hist"], want["hist"])
    assert np.array_equal(got["first_0_out"], want["first_0_out"])
    assert np.array_equal(got["last_0_in"], want["last_0_in"])


@pytest.mark.parametrize("opts", [dict(), dict(s1_stream_fill=40), dict(s1_pos_bits=12), dict(s1_stream_bits=19), dict(s1_stream_sub0=2)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()) or "default")
@pytest.mark.parametrize("kind,k,m,n_passes", [("pe100", 21, 2, 3), ("repeats100", 21, 2, 4), ("short30", 21, 2, 2), ("pe100", 17, 1, 3), ("repeats100", 22, 2, 5), ("repeats100", 21, 3, 3), ("pe100", 27, 2, 3), ("repeats100", 24, 3, 2)])
def test_count_passes_on_the_streaming_design(engine, kind, k, m, n_passes, opts):
    if k > 22 and "s1_pos_bits" in opts:
        pytest.skip("k = 23..27: the two key words have no room for a position tag (the tile path serves tagged read sets there)")
    pkg = ob.Package(fixed_library(kind, seed=k * 5 + m), reverse=True)
    want = ob.count(pkg, k, m)
    got, stats_h, per_pass, ranges = run_passes(engine, pkg, k, m, n_passes, opts)
    assert len(ranges) >= 2
    assert "count_bucket_hist" in stats_h, sorted(stats_h)  # the plan's lv1 histogram: from the packed reads, no items extracted
    assert sum(r.n_items for _, r in per_pass) == want["n_items"]
    for st, r in per_pass:
        if r.n_items:  # (a range without a single edge has nothing to stream)
            assert "count_digit_hist" in st and "count_groups" in st and "count_extract" not in st and "count_runs" not in st, sorted(st)
            assert r.item_words == 3
    check(got, want)


@pytest.mark.parametrize("kind,k,m", [("var", 21, 2), ("lowcomplex", 21, 2), ("var", 17, 1)])
def test_count_passes_of_reads_of_several_lengths(engine, kind, k, m):
    pkg = ob.Package(make_reads(kind, 17), reverse=True)
    want = ob.count(pkg, k, m)
    got, stats_h, per_pass, ranges = run_passes(engine, pkg, k, m, 3, dict(s1_var_min_fill=10))
    assert "count_bucket_hist" in stats_h, sorted(stats_h)
    for st, r in per_pass:
        if r.n_items:
            assert "count_digit_hist" in st and "count_extract" not in st, sorted(st)
    check(got, want)


def test_a_pass_that_gives_up_continues_on_the_tile_path(engine):
    """pass 1 of 3 runs with a probe limit of 0: its streaming form raises the error word half-way, the state of pass 0 is restored and
    the tile path redoes the pass on top of it; pass 2 streams again — the three kinds of pass accumulate into one result"""
    k, m = 21, 2
    pkg = ob.Package(fixed_library("repeats100", seed=99), reverse=True)
    want = ob.count(pkg, k, m)
    got, _, per_pass, ranges = run_passes(engine, pkg, k, m, 3, {}, per_pass_opts={1: dict(s1_stream_probes=0), 2: dict(s1_stream_probes=1024)})
    assert len(ranges) >= 3
    assert "count_runs" in per_pass[1][0] or "count_groups" in per_pass[1][0]
    assert "count_extract" in per_pass[1][0] and "count_extract" not in per_pass[0][0] and "count_extract" not in per_pass[2][0]
    check(got, want)


def test_shapes_outside_the_streaming_design_still_take_the_tile_path_in_passes(engine):
    pkg = ob.Package(fixed_library("pe100", seed=5), reverse=True)
    want = ob.count(pkg, 29, 3)
    got, stats_h, per_pass, _ = run_passes(engine, pkg, 29, 3, 3, {})
    assert "count_bucket_hist" not in stats_h
    assert all("count_digit_hist" not in st for st, _ in per_pass)
    check(got, want)
