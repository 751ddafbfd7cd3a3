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
