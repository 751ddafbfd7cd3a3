"""GPU: the prefix sort + LDS segment finish (sort.hip: sort_whole_key / k_seg_finish) orders records exactly as the plain
LSD plan does — the job of kmlib::kmsort behind SelectSortingFunc (reference src/sorting/kmsort_selector.cpp:39-63,
src/kmlib/kmsort.h:45-122) — including its fallback when a segment outgrows the look-ahead, and the engines built on it
(stage 2, seq2sdbg, count's full-sort path, stage 1 with mercy) still reproduce the oracle / the reference's digests."""
import os

import numpy as np
import pytest

import golden_util as gu
import oracle_binding as ob
from megahit_amd import lib
from test_gpu_count import load, make_reads
from test_gpu_sdbg import check_sdbg

pytestmark = pytest.mark.gpu

HYB = {"sort_hybrid": 2, "sort_hybrid_min": 0}  # force the hybrid path also where it saves no pass, at any size


@pytest.fixture
def hyb(engine):
    def set_opts(**kw):
        for name, v in dict(HYB, **kw).items():
            engine.set_option(name, v)
    yield set_opts
    for name, v in (("sort_hybrid", 1), ("sort_hybrid_min", 1 << 16), ("sort_hybrid_bits", 0), ("sort_hybrid_avg", 4)):
        engine.set_option(name, v)


@pytest.mark.parametrize("kw,aux,n,bits", [(1, 1, 5000, 8), (2, 0, 300000, 0), (2, 1, 200000, 16), (2, 2, 100000, 0), (3, 1, 150000, 0),
                                           (5, 1, 60000, 16), (9, 1, 40000, 0), (10, 0, 30000, 8), (17, 1, 9000, 8), (20, 0, 7000, 0)])
def test_sort_records_hybrid(engine, hyb, kw, aux, n, bits):
    """random keys with many duplicates (stability) and a few heavy prefixes (long segments next to short ones)"""
    hyb(sort_hybrid_bits=bits)
    rng = np.random.default_rng(kw * 1000 + aux + n)
    items = rng.integers(0, 2 ** 32, size=(n, kw + aux), dtype=np.uint64).astype(np.uint32)
    dup = rng.integers(0, n, size=n // 3)
    items[: n // 3, :kw] = items[dup, :kw]           # exact duplicates of whole keys, aux differs
    heavy = rng.integers(0, n, size=n // 50)
    items[heavy, 0] = np.uint32(0x12345678)           # one prefix shared by 2 % of the records
    want = ob.sort_items(items, kw, kmsort=False)     # stable
    got = engine.sort_records(items.copy(), kw)
    assert np.array_equal(got, want)


def test_sort_records_hybrid_falls_back_on_a_huge_segment(engine, hyb):
    hyb(sort_hybrid_bits=16)
    rng = np.random.default_rng(5)
    n = 200000
    items = rng.integers(0, 2 ** 32, size=(n, 3), dtype=np.uint64).astype(np.uint32)
    items[n // 4: n // 2, 0] = np.uint32(0xABCD0000) | (items[n // 4: n // 2, 0] & np.uint32(0xFFFF))  # 50 000 records, one 16-bit prefix
    want = ob.sort_items(items, 2, kmsort=False)
    assert np.array_equal(engine.sort_records(items.copy(), 2), want)


@pytest.mark.parametrize("bits", [0, 8, 32])
@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("var", 27, 1), ("var", 27, 3), ("lowcomplex", 21, 2), ("var", 47, 2), ("fixed", 63, 2)])
def test_read2sdbg_with_hybrid_sort(engine, hyb, kind, k, m, bits):
    hyb(sort_hybrid_bits=bits)
    reads = make_reads(kind, 5)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    if m > 1:
        w1 = ob.s1(pkg, k, m, tie_stable=True)
        r1 = engine.read2sdbg_s1(k, m, want_mercy=True)  # full stage-1 sort (mercy): the first record of a group matters
        assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
        solid = w1["is_solid"]
        assert np.array_equal(engine.fetch(lib.BUF_IS_SOLID, np.uint64), solid[: engine.fetch(lib.BUF_IS_SOLID, np.uint64).size])
    else:
        solid = None
    r2 = engine.read2sdbg_s2(k, m)
    check_sdbg(engine, r2, ob.s2(pkg, k, m, solid))


@pytest.mark.parametrize("k", [21, 29, 61, 119])
def test_count_then_seq2sdbg_with_hybrid_sort(engine, hyb, k):
    hyb()
    reads = make_reads("var", 7)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    want = ob.count(pkg, k, 2)
    engine.set_option("count_seg", 0)  # the full-sort path of count
    try:
        rc = engine.count(k, 2)
    finally:
        engine.set_option("count_seg", 1)
    edges = engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, rc.words_per_edge)
    assert np.array_equal(edges, want["edges"])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want["hist"])


@pytest.mark.parametrize("ent", [e for e in gu.cases() if e["case"]["prog"] == "seq2sdbg"], ids=gu.case_id)
def test_golden_seq2sdbg_with_hybrid_sort(ent, tmp_path, monkeypatch):
    """the committed known answers of the reference's seq2sdbg (k = 21 ... 119, contigs + edges), hybrid sort forced"""
    monkeypatch.setenv("MHX_SORT_HYBRID", "2")
    monkeypatch.setenv("MHX_SORT_HYBRID_MIN", "0")
    got = gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    for key, want in ent.items():
        if key != "case":
            assert got.get(key) == want, key
