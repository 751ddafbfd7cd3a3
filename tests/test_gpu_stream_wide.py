"""GPU: stage 1's bucket streaming at k = 23..29 (round 6): the local key — the (k-1)-mer below the plan's prefix + head/tail — no longer
fits 32 bits, the LDS table takes 64-bit keys (k_s1_stream<..., K64>, k_s1_giant_reduce<true>); no aggregated stage-2 items beyond
k = 22, stage 2 runs per occurrence.  Against the oracle's Read2SdbgS1 / Read2SdbgS2 (reference src/sorting/read_to_sdbg_s1.cpp:368-464,
read_to_sdbg_s2.cpp:521-614): every k of the range, overflowing tables that split themselves, sub-rounds, wider prefixes, position tags
(k <= 26: eight spare key bits), giant buckets (reads of one base, tandem repeats), bucket-range passes with the filter inside the
generating pass, variable-length libraries, several ranks; and the round-5 path (k_s1_seg) beside it with s1_stream_wide = 0."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib, passes
from test_gpu_count import load, make_reads
from test_gpu_passes import _check_sdbg
from test_gpu_round3_knobs import fixed_library
from test_gpu_sdbg import check_sdbg

pytestmark = pytest.mark.gpu

RESET = dict(s1_stream_wide=1, s1_stream_fill=7168, s1_pos_bits=0, s1_stream_bits=0, s1_stream_sub0=-1, s1_giant_min=262144, s1_giant=1, s1_stream_direct=1,
             s1_stream_probes=1024, s1_var_min_fill=50)


def run(engine, reads, k, m, opts, want_plan="stream", want_kernels=()):
    pkg = ob.Package(reads, reverse=True)
    want1 = ob.s1(pkg, k, m, tie_stable=True)
    want2 = ob.s2(pkg, k, m, want1["is_solid"])
    load(engine, pkg)
    try:
        for n, v in opts.items():
            engine.set_option(n, v)
        engine.profile(True)
        engine.profile_reset()
        r1 = engine.read2sdbg_s1(k, m)
        stats = engine.profile_get()
        engine.profile(False)
        assert engine.last_s1_plan().startswith(want_plan), engine.last_s1_plan()
        for kn in want_kernels:
            assert kn in stats, sorted(stats)
        solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert r1.n_items == want1["n_items"]
        assert np.array_equal(solid, want1["is_solid"][: solid.size])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want1["hist"])
        check_sdbg(engine, engine.read2sdbg_s2(k, m), want2)
    finally:
        engine.profile(False)
        for n, v in RESET.items():
            engine.set_option(n, v)


@pytest.mark.parametrize("opts", [dict(), dict(s1_stream_fill=40), dict(s1_stream_bits=19), dict(s1_stream_sub0=2), dict(s1_stream_direct=0)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()) or "default")
@pytest.mark.parametrize("kind,k,m", [("pe100", 23, 2), ("pe100", 25, 2), ("repeats100", 27, 2), ("pe100", 29, 2), ("repeats100", 24, 3), ("short30", 23, 2)])
def test_wide_keys_on_the_bucket_streaming(engine, kind, k, m, opts):
    run(engine, fixed_library(kind, seed=k * 3 + m), k, m, opts, want_kernels=("s1_groups",))


@pytest.mark.parametrize("kind,k", [("pe100", 23), ("repeats100", 26)])
def test_position_tags_with_wide_keys(engine, kind, k):
    run(engine, fixed_library(kind, seed=k), k, 2, dict(s1_pos_bits=12))


@pytest.mark.parametrize("opts", [dict(s1_giant_min=64), dict(s1_giant_min=64, s1_stream_fill=40), dict(s1_giant_min=200, s1_stream_bits=18)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()))
@pytest.mark.parametrize("kind,k", [("repeats100", 27), ("lowcomplex", 25), ("pe100", 23)])
def test_giant_buckets_with_wide_keys(engine, kind, k, opts, monkeypatch):
    monkeypatch.setenv("MHX_S1_MARK", "nonsolid")  # (the giant path rides on the marks of the non-solid occurrences taken from the table)
    reads = make_reads(kind, 7) if kind == "lowcomplex" else fixed_library(kind, seed=k)
    if kind == "lowcomplex":
        opts = dict(opts, s1_var_min_fill=10)
    run(engine, reads, k, 2, opts, want_kernels=("s1_giant_groups",))


def test_a_million_records_of_one_key_at_k27(engine):
    """poly-A / poly-C / (AC)n reads: >= 10^6 records of ONE key in each of three buckets of the 16-bit prefix — found, cut into slices and
    reduced on the device with 64-bit slice tables"""
    rng = np.random.default_rng(5)
    reads = [x for x in fixed_library("pe100", seed=27)]
    for pat in ([0], [1], [0, 1]):
        reads += [np.tile(np.array(pat, dtype=np.uint8), 150 // len(pat)) for _ in range(9000)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    r1 = engine.read2sdbg_s1(27, 2)
    assert "giant buckets in slices" in engine.last_s1_plan(), engine.last_s1_plan()
    want1 = ob.s1(pkg, 27, 2, tie_stable=True)
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert r1.n_items == want1["n_items"] and np.array_equal(solid, want1["is_solid"][: solid.size])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want1["hist"])


@pytest.mark.parametrize("kind,k", [("var", 23), ("var", 27)])
def test_reads_of_several_lengths_with_wide_keys(engine, kind, k):
    """k = 23: the padded-slot generator (S1GenVarT: one window per run, k <= 23); k = 27: the extraction kernel + loaded passes"""
    run(engine, make_reads(kind, 9), k, 2, dict(s1_var_min_fill=10))


def test_the_round5_path_beside_it(engine):
    run(engine, fixed_library("pe100", seed=4), 27, 2, dict(s1_stream_wide=0), want_plan="seg")


@pytest.mark.parametrize("kind,k,m", [("pe100", 27, 2), ("repeats100", 23, 2)])
def test_bucket_range_passes_with_wide_keys(engine, kind, k, m):
    pkg = ob.Package(fixed_library(kind, seed=k + 1), reverse=True)
    load(engine, pkg)
    w1 = ob.s1(pkg, k, m, tie_stable=True)
    want = ob.s2(pkg, k, m, w1["is_solid"])
    s1, got = passes.read2sdbg_in_passes(engine, k, m, max_items_s1=-3, max_items_s2=-3, need_mercy=0, batch_bytes=0)
    assert s1["n_passes"] >= 3 and s1["n_items"] == w1["n_items"]
    assert engine.last_s1_plan().startswith("stream"), engine.last_s1_plan()
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
    bits = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(bits, w1["is_solid"][: bits.size])
    _check_sdbg(got, want)


@pytest.mark.parametrize("world,k", [(2, 27), (3, 23)])
def test_wide_keys_on_several_ranks(world, k):
    """the pre-sorted exchange with 64-bit local keys at the owners: a bucket's records arrive as one sub-range per sender"""
    from test_gpu_comm import run_ranks, load_fixed_reads, sdbg_of, check_sdbg as check_ranks
    reads = [None] * world

    def load_r(r, e):
        reads[r] = load_fixed_reads(r, e)

    def body(r, e, cm):
        cm.setup(1, k, 2)
        cm.read2sdbg(k, 2)
        return sdbg_of(e) + (e.fetch(lib.BUF_MUL_HIST, np.int64), e.last_s1_plan())

    outs = run_ranks(world, load_r, body)
    allr = []
    for r in range(world):
        allr += reads[r]
    pkg = ob.Package(allr, reverse=True)
    s1 = ob.s1(pkg, k, 2)
    assert np.array_equal(sum(o[4] for o in outs), s1["hist"])
    for o in outs:
        assert o[5].startswith("stream") and "pre-sorted exchange" in o[5], o[5]
    check_ranks(outs, ob.s2(pkg, k, 2, s1["is_solid"]))
