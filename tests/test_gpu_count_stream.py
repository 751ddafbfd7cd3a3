"""GPU: `count` on the design of stage 1 (CountGenT + k_s1_stream<COUNT>, s1.hip / count.hip: the first sort pass makes 12-byte
records, prefix passes, LDS group-by per bucket, first_0_out / last_0_in from a second look at the buckets that need it) against the
oracle's KmerCounter (reference src/sorting/kmer_counter.cpp:208-381): edges, per-bucket counts, multiplicity histogram,
first_0_out / last_0_in — on fixed-length libraries incl. low-complexity reads, k up to 22, min count 1 and 2, with sub-rounds,
overflowing tables, position tags and wider prefixes; and the tile path (count_stream = 0) beside it."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib
from test_gpu_count import load
from test_gpu_round3_knobs import fixed_library

pytestmark = pytest.mark.gpu

RESET = dict(count_stream=1, s1_stream_fill=7168, s1_pos_bits=0, s1_stream_bits=0, s1_stream_sub0=-1, s1_stream_probes=1024, s1_giant_min=262144, count_giant=1, count_stream_wide=1)


def check_count(engine, pkg, k, m, opts, expect_stream):
    want = ob.count(pkg, k, m)
    load(engine, pkg)
    try:
        for n, v in opts.items():
            engine.set_option(n, v)
        engine.profile(True)
        engine.profile_reset()
        r = engine.count(k, m)
        stats = engine.profile_get()
        engine.profile(False)
    finally:
        engine.profile(False)
        for n, v in RESET.items():
            engine.set_option(n, v)
    assert ("count_digit_hist" in stats) == expect_stream, sorted(stats)
    assert r.n_items == want["n_items"] and r.words_per_edge == want["wpe"]
    edges = engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge)
    assert edges.shape == want["edges"].shape
    assert np.array_equal(edges, want["edges"])
    assert np.array_equal(engine.fetch(lib.BUF_BUCKET_COUNT, np.uint64), want["bucket_count"])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want["hist"])
    assert np.array_equal(engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), want["first_0_out"])
    assert np.array_equal(engine.fetch(lib.BUF_LAST_0_IN, np.uint32), want["last_0_in"])


@pytest.mark.parametrize("opts", [dict(), dict(count_stream=0), dict(s1_stream_fill=40), dict(s1_pos_bits=12), dict(s1_stream_bits=19), dict(s1_stream_sub0=2),
                                  dict(s1_stream_probes=0)], ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()) or "default")
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 21, 2), ("tiny60", 21, 2), ("short30", 21, 2), ("pe100", 17, 1), ("repeats100", 22, 2),
                                      ("pe100", 13, 2), ("repeats100", 21, 1),
                                      # round 6: min count 3..15 — per-char 4-bit counters that stop at m instead of the seen-once / seen-twice bits
                                      ("pe100", 21, 3), ("repeats100", 21, 3), ("repeats100", 22, 5), ("pe100", 19, 15), ("short30", 21, 4)])
def test_count_on_the_bucket_streaming(engine, kind, k, m, opts):
    reads = fixed_library(kind, seed=k * 7 + m)
    pkg = ob.Package(reads, reverse=True)
    # (k = 13: a full-sort plan, not the stream plan — the tile path; a probe limit of 0 makes the stream form give up -> the tile path)
    expect_stream = opts.get("count_stream", 1) == 1 and k >= 17
    check_count(engine, pkg, k, m, opts, expect_stream)


@pytest.mark.parametrize("opts", [dict(), dict(s1_stream_fill=40), dict(s1_stream_bits=19), dict(s1_stream_sub0=2), dict(s1_giant_min=64), dict(count_stream_wide=0)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()) or "default")
@pytest.mark.parametrize("kind,k,m", [("pe100", 23, 2), ("repeats100", 24, 2), ("pe100", 25, 3), ("repeats100", 27, 2), ("pe100", 27, 1), ("pe100", 26, 2)])
def test_count_at_k_23_to_27_on_the_bucket_streaming(engine, kind, k, m, opts):
    """round 6: a window per item (CountGenWideT), 64-bit table keys, three-word edges from k = 24 (16-byte region entries); reads of one
    length, no position tags"""
    reads = fixed_library(kind, seed=k * 7 + m)
    pkg = ob.Package(reads, reverse=True)
    try:
        check_count(engine, pkg, k, m, opts, expect_stream=opts.get("count_stream_wide", 1) == 1)
    finally:
        engine.set_option("count_stream_wide", 1)


@pytest.mark.parametrize("k,m", [(28, 2), (21, 16)])
def test_shapes_the_stream_form_does_not_take(engine, k, m):
    reads = fixed_library("pe100", seed=3)
    check_count(engine, ob.Package(reads, reverse=True), k, m, {}, False)


@pytest.mark.parametrize("opts", [dict(), dict(s1_var_fast=0), dict(s1_stream_fill=40), dict(s1_pos_bits=12)], ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()) or "default")
@pytest.mark.parametrize("kind,k,m", [("trim", 21, 2), ("few", 21, 2), ("edge", 21, 2), ("trim", 17, 1), ("few", 22, 2), ("lowcomplex", 21, 2), ("var", 21, 2)])
def test_count_of_reads_of_several_lengths(engine, kind, k, m, opts):
    """CountGenVarT: item slots padded to the longest read's — libraries with every read trimmed, with 2 % trimmed, with reads of
    length 0, 1, k - 1, k, k + 1 and an empty first read, and the low-complexity / random-length libraries of the other count tests"""
    from test_gpu_count import make_reads
    from test_gpu_round5_knobs import var_library
    reads = make_reads(kind, 11) if kind in ("lowcomplex", "var") else var_library(kind, seed=k + m)
    pkg = ob.Package(reads, reverse=True)
    engine.set_option("s1_var_min_fill", 10)
    try:
        check_count(engine, pkg, k, m, opts, expect_stream=opts.get("s1_var_fast", 1) == 1)
    finally:
        engine.set_option("s1_var_min_fill", 50)
        engine.set_option("s1_var_fast", 1)


@pytest.mark.parametrize("opts", [dict(s1_giant_min=64), dict(s1_giant_min=1000), dict(s1_giant_min=64, s1_stream_fill=40), dict(s1_giant_min=64, s1_pos_bits=12),
                                  dict(s1_giant_min=64, s1_stream_bits=19), dict(s1_giant_min=300, s1_stream_sub0=2), dict(s1_giant_min=64, count_giant=0)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()))
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 21, 2), ("repeats100", 22, 1), ("repeats100", 21, 3), ("pe100", 17, 5)])
def test_giant_buckets_of_count(engine, kind, k, m, opts):
    """round 6: a bucket of >= s1_giant_min records is cut into slices (k_s1_giant_reduce<., COUNT>: per key its count and per-char counters
    in the slice), skipped by the streaming launch and finished by a second launch on the partial entries — the threshold scaled down so that
    ordinary buckets take the path, incl. buckets with solid keys without an in- or out-edge (the second look at the bucket's records)"""
    reads = fixed_library(kind, seed=k * 11 + m)
    pkg = ob.Package(reads, reverse=True)
    want = ob.count(pkg, k, m)
    load(engine, pkg)
    try:
        for n, v in opts.items():
            engine.set_option(n, v)
        engine.profile(True)
        engine.profile_reset()
        r = engine.count(k, m)
        stats = engine.profile_get()
    finally:
        engine.profile(False)
        for n, v in RESET.items():
            engine.set_option(n, v)
    assert ("count_giant_groups" in stats) == (opts.get("count_giant", 1) == 1), sorted(stats)
    edges = engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge)
    assert edges.shape == want["edges"].shape and np.array_equal(edges, want["edges"])
    assert np.array_equal(engine.fetch(lib.BUF_BUCKET_COUNT, np.uint64), want["bucket_count"])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want["hist"])
    assert np.array_equal(engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), want["first_0_out"])
    assert np.array_equal(engine.fetch(lib.BUF_LAST_0_IN, np.uint32), want["last_0_in"])


@pytest.mark.parametrize("m", [2, 3])
def test_a_million_records_of_one_key_in_count(engine, m):
    """poly-A / poly-C / (AC)n reads: >= 10^6 records of ONE key in each of three buckets, default threshold — and reads that END in the
    repeat (solid keys without an out-edge inside a giant bucket: the second launch looks at the bucket's records)"""
    rng = np.random.default_rng(8)
    reads = [x for x in fixed_library("pe100", seed=33)]
    for pat in ([0], [1], [0, 1]):
        reads += [np.tile(np.array(pat, dtype=np.uint8), 150 // len(pat)) for _ in range(9000)]
    for _ in range(40):  # random sequence running into a poly-A tail, several copies each: solid edges at the junction
        head = rng.integers(0, 4, size=60, dtype=np.uint8)
        for _c in range(3):
            reads.append(np.concatenate([head, np.zeros(90, dtype=np.uint8)]))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    pkg = ob.Package(reads, reverse=True)
    want = ob.count(pkg, 21, m)
    load(engine, pkg)
    engine.set_option("s1_var_min_fill", 10)
    try:
        r = engine.count(21, m)
        plan = engine.last_s1_plan()
    finally:
        engine.set_option("s1_var_min_fill", 50)
    assert "giant buckets in slices" in plan, plan
    edges = engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge)
    assert np.array_equal(edges, want["edges"])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want["hist"])
    assert np.array_equal(engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), want["first_0_out"])
    assert np.array_equal(engine.fetch(lib.BUF_LAST_0_IN, np.uint32), want["last_0_in"])
