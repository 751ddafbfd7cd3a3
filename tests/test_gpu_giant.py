"""GPU: giant buckets of stage 1's bucket streaming (S1Giant, s1.hip): a bucket of the plan's prefix with >= s1_giant_min records is
cut into slices that many workgroups reduce in parallel (k_s1_giant_reduce), skipped by the streaming launch and finished by a
second launch on the slices' partial entries — against the oracle (Read2SdbgS1::Lv2Postprocess, reference
src/sorting/read_to_sdbg_s1.cpp:368-464):
  * the threshold scaled down to fixture size, so that ordinary buckets take the path: buckets that reduce (few keys), buckets whose
    slices do not fit their region and go back to the streaming launch, both in one run;
  * >= 10^6 records of ONE key in each of three buckets (poly-A, poly-C, (AC)n reads) with the default threshold;
  * with sub-rounds (tables that overflow and split), position tags, prefix widths beyond 16 bits, variable-length libraries;
  * a profile line shows the path ran (s1_giant_groups)."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib
from test_gpu_count import load, make_reads
from test_gpu_round3_knobs import fixed_library
from test_gpu_sdbg import check_sdbg

pytestmark = pytest.mark.gpu

RESET = dict(s1_giant=1, s1_giant_min=262144, s1_stream_fill=7168, s1_pos_bits=0, s1_stream_bits=0, s1_stream_sub0=-1, s1_stream_direct=1)


def run(engine, reads, k, m, opts, expect_giants=None, expect_found=None):
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
        assert "stream" in engine.last_s1_plan(), engine.last_s1_plan()
        plan_text = engine.last_s1_plan()
        solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert r1.n_items == want1["n_items"]
        assert np.array_equal(solid, want1["is_solid"][: solid.size])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want1["hist"])
        check_sdbg(engine, engine.read2sdbg_s2(k, m), want2)
        if expect_giants is not None:
            assert ("s1_giant_groups" in stats) == expect_giants, sorted(stats)
        if expect_found is not None:
            assert ("giant buckets in slices" in plan_text) == expect_found, plan_text
    finally:
        engine.profile(False)
        for n, v in RESET.items():
            engine.set_option(n, v)


@pytest.mark.parametrize("opts", [dict(s1_giant_min=64), dict(s1_giant_min=1000), dict(s1_giant_min=64, s1_stream_fill=40), dict(s1_giant_min=64, s1_pos_bits=12),
                                  dict(s1_giant_min=64, s1_stream_bits=19), dict(s1_giant_min=300, s1_stream_sub0=2), dict(s1_giant=0, s1_giant_min=64)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()))
@pytest.mark.parametrize("kind,k", [("pe100", 21), ("repeats100", 21), ("lowcomplex", 21), ("var", 21), ("repeats100", 22), ("pe100", 17), ("lowcomplex", 16)])
def test_small_threshold(engine, kind, k, opts, monkeypatch):
    monkeypatch.setenv("MHX_S1_MARK", "nonsolid")  # (the giant path rides on the marks of the NON-solid occurrences taken from the table)
    reads = make_reads(kind, 7) if kind in ("var", "lowcomplex") else fixed_library(kind, seed=k)
    on = bool(opts.get("s1_giant", 1))
    run(engine, reads, k, 2, opts, expect_giants=on, expect_found=on if kind in ("repeats100", "lowcomplex") else (None if on else False))


def three_giants(n_each, seed):
    """n_each reads each of poly-A, poly-C and (AC)n, 150 bp (133 stage-1 records per read at k = 21, all of one key per kind), among
    random reads and reads of a small genome; a few of the low-complexity reads carry a substitution (singleton keys in the giants)"""
    rng = np.random.default_rng(seed)
    reads = []
    for base in (np.zeros(150, dtype=np.uint8), np.ones(150, dtype=np.uint8), np.tile(np.array([0, 1], dtype=np.uint8), 75)):
        for i in range(n_each):
            r = base.copy()
            if i % 97 == 0:
                r[int(rng.integers(0, 150))] = int(rng.integers(0, 4))
            reads.append(r)
    g = rng.integers(0, 4, size=20000, dtype=np.uint8)
    reads += [g[s:s + 150].copy() for s in rng.integers(0, 19850, size=6000)]
    reads += [rng.integers(0, 4, size=150, dtype=np.uint8) for _ in range(3000)]
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


@pytest.mark.parametrize("opts", [dict(), dict(s1_giant=0)], ids=["giant-path", "streamed-alone"])
def test_three_buckets_of_a_million_records_of_one_key(engine, opts):
    reads = three_giants(7600, 3)  # 7600 x 133 = 1 010 800 records per kind
    run(engine, reads, 21, 2, opts, expect_giants=not opts, expect_found=not opts)


def test_variable_length_library_with_giants(engine, monkeypatch):
    monkeypatch.setenv("MHX_S1_MARK", "nonsolid")
    rng = np.random.default_rng(5)
    reads = three_giants(2200, 4)
    reads = [r[: int(rng.integers(60, 151))] for r in reads]
    run(engine, reads, 21, 2, dict(s1_giant_min=100000), expect_giants=True, expect_found=True)
