"""GPU: the round-5 forms of the stage-1 front (one 64-bit window and ONE reverse complement per run of a thread's eight consecutive
items: S1GenRollT, k_s1_digit_hist_roll, k_s1_bucket_hist_fast<.., true>; s1.hip) against the oracle and against the forms they
replace, on fixed-length libraries incl. k = 23 (the widest window the run form takes) and k = 24 (falls back to the blocked form),
reads of 13 slots (a read boundary inside most runs) and a library smaller than a tile.
Reference: Read2SdbgS1 (src/sorting/read_to_sdbg_s1.cpp:145-366)."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib
from test_gpu_count import load
from test_gpu_round3_knobs import fixed_library
from test_gpu_sdbg import check_sdbg

pytestmark = pytest.mark.gpu

SETTINGS = [dict(s1_gen_roll=1, s1_digit_hist_roll=1), dict(s1_gen_roll=0, s1_digit_hist_roll=1), dict(s1_gen_roll=1, s1_digit_hist_roll=0),
            dict(s1_gen_roll=0, s1_digit_hist_roll=0)]


@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s_: ",".join("%s=%d" % kv for kv in s_.items()))
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 21, 2), ("tiny60", 21, 2), ("short30", 21, 2), ("pe100", 17, 3), ("repeats100", 22, 2),
                                      ("pe100", 23, 2), ("short30", 23, 2), ("pe100", 24, 2), ("pe100", 13, 2)])
def test_stage1_front_forms(engine, kind, k, m, setting):
    reads = fixed_library(kind, seed=k * 10 + m)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    want1 = ob.s1(pkg, k, m, tie_stable=True)
    want2 = ob.s2(pkg, k, m, want1["is_solid"])
    try:
        engine.set_option("s1_gen_blocked", 1)
        for name, v in setting.items():
            engine.set_option(name, v)
        # the lv1 histogram a memory plan asks for (k_s1_bucket_hist_fast) shares the arithmetic
        hist = engine.bucket_histogram(lib.STAGE_S1, k, m)
        assert int(hist.sum()) == want1["n_items"]
        r1 = engine.read2sdbg_s1(k, m)
        solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert r1.n_items == want1["n_items"]
        assert np.array_equal(solid, want1["is_solid"][: solid.size])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want1["hist"])
        check_sdbg(engine, engine.read2sdbg_s2(k, m), want2)
    finally:
        from test_gpu_round3_knobs import engine_default
        engine.set_option("s1_gen_blocked", engine_default(engine, "s1_gen_blocked"))
        engine.set_option("s1_gen_roll", 1)
        engine.set_option("s1_digit_hist_roll", 1)


@pytest.mark.parametrize("roll", [0, 1])
def test_bucket_histogram_forms_agree(engine, roll):
    """the lv1 histogram from the packed reads, both forms, bucket by bucket against the items the oracle enumerates"""
    reads = fixed_library("repeats100", seed=5)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    try:
        engine.set_option("s1_digit_hist_roll", roll)
        fast = engine.bucket_histogram(lib.STAGE_S1, 21, 2)
        engine.set_option("s1_bucket_hist_fast", 0)
        slow = engine.bucket_histogram(lib.STAGE_S1, 21, 2)
    finally:
        engine.set_option("s1_bucket_hist_fast", 1)
        engine.set_option("s1_digit_hist_roll", 1)
    assert np.array_equal(fast, slow)


def var_library(kind, seed):
    """libraries whose reads are not of one length: `trim` = reads of 100 bases, every one cut to U[60, 100]; `few` = 2 % of the reads
    cut (N-trimmed reads of a real library); `edge` = reads of length 0, 1, k - 1, k, k + 1, k + 4 and long ones mixed, an empty read first"""
    rng = np.random.default_rng(seed)
    reads = fixed_library("repeats100" if kind == "few" else "pe100", seed)
    if kind == "trim":
        return [r[: int(rng.integers(60, 101))] for r in reads]
    if kind == "few":
        return [r[: int(rng.integers(30, 100))] if rng.random() < 0.02 else r for r in reads]
    if kind == "edge":
        out = [np.zeros(0, dtype=np.uint8)]
        for i, r in enumerate(reads[:1500]):
            out.append(r[: [0, 1, 20, 21, 22, 25, 100, 100, 100, 100][i % 10]])
        return out
    raise ValueError(kind)


@pytest.mark.parametrize("var_fast", [1, 0])
@pytest.mark.parametrize("kind,k,m", [("trim", 21, 2), ("few", 21, 2), ("edge", 21, 2), ("trim", 17, 3), ("few", 23, 2), ("trim", 24, 2), ("edge", 13, 2)])
def test_reads_of_several_lengths_on_the_generating_pass(engine, kind, k, m, var_fast):
    """S1GenVarT: item slots padded to the longest read's, the slots a read does not fill declined — against the oracle and against
    the extraction kernel (s1_var_fast = 0); k = 24 is beyond the run form and takes the extraction kernel either way"""
    reads = var_library(kind, seed=k + m)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    want1 = ob.s1(pkg, k, m, tie_stable=True)
    want2 = ob.s2(pkg, k, m, want1["is_solid"])
    try:
        engine.set_option("s1_gen_blocked", 1)
        engine.set_option("s1_var_fast", var_fast)
        engine.set_option("s1_var_min_fill", 10)  # (the edge library is mostly short reads: take the padded form anyway)
        hist = engine.bucket_histogram(lib.STAGE_S1, k, m)
        assert int(hist.sum()) == want1["n_items"]
        r1 = engine.read2sdbg_s1(k, m)
        assert ("item slots per read on the generating pass" in engine.last_s1_plan()) == bool(var_fast and 17 <= k <= 23), engine.last_s1_plan()  # (k = 13: a full-sort plan, whose digits leave the first key word)
        solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert r1.n_items == want1["n_items"]
        assert np.array_equal(solid, want1["is_solid"][: solid.size])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want1["hist"])
        check_sdbg(engine, engine.read2sdbg_s2(k, m), want2)
    finally:
        from test_gpu_round3_knobs import engine_default
        engine.set_option("s1_gen_blocked", engine_default(engine, "s1_gen_blocked"))
        engine.set_option("s1_var_fast", 1)
        engine.set_option("s1_var_min_fill", 50)


@pytest.mark.parametrize("in_gen", [1, 0])
@pytest.mark.parametrize("kind,k,m,opts", [("trim", 21, 2, {}), ("few", 21, 2, dict(s1_stream_bits=19)), ("edge", 21, 2, dict(s1_var_min_fill=10)),
                                           ("trim", 17, 2, dict(s1_pos_bits=12))])
def test_bucket_filter_on_reads_of_several_lengths(engine, kind, k, m, in_gen, opts):
    """stage 1 in bucket-range passes (what mhx_core's memory plan does) on a library whose reads are not of one length: with
    s1_filter_in_gen the generating pass of each range makes only the records of the kept buckets AND declines the slots a read does
    not fill (S1GenVarT<true>), the lv1 histogram comes from the packed reads (k_s1_bucket_hist_var); without it extraction batches"""
    from megahit_amd import passes
    from test_gpu_passes import _check_sdbg
    reads = var_library(kind, seed=k + m)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    w1 = ob.s1(pkg, k, m, tie_stable=True)
    want = ob.s2(pkg, k, m, w1["is_solid"])
    try:
        engine.set_option("s1_gen_blocked", 1)
        engine.set_option("s1_filter_in_gen", in_gen)
        for n, v in opts.items():
            engine.set_option(n, v)
        s1, got = passes.read2sdbg_in_passes(engine, k, m, max_items_s1=-3, max_items_s2=-4, need_mercy=0, batch_bytes=8 << 20)
    finally:
        from test_gpu_round3_knobs import engine_default
        engine.set_option("s1_gen_blocked", engine_default(engine, "s1_gen_blocked"))
        engine.set_option("s1_filter_in_gen", 1)
        for n, v in dict(s1_stream_bits=0, s1_var_min_fill=50, s1_pos_bits=0).items():
            engine.set_option(n, v)
    assert s1["n_passes"] >= 3
    assert s1["n_items"] == w1["n_items"]
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
    bits = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(bits, w1["is_solid"][: bits.size])
    got["n_passes"] = max(got["n_passes"], 3)
    _check_sdbg(got, want)
