"""GPU: memory-bounded bucket-range passes (the reference's lv1 passes) give the single-pass result."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib, passes
from test_gpu_count import load, make_reads
from test_gpu_sdbg import edges_package

pytestmark = pytest.mark.gpu

BATCH = 8 << 20  # small staging so that every pass runs over many read batches


def _check_sdbg(got, want):
    assert got["n_passes"] >= 3
    assert got["wpt"] == want["wpt"]
    assert np.array_equal(got["bytes"], want["bytes"])
    assert np.array_equal(got["bucket_items"], want["bucket_items"])
    assert np.array_equal(got["bucket_tips"], want["bucket_tips"])
    assert np.array_equal(got["bucket_large"], want["bucket_large"])
    nz = want["bucket_items"] > 0
    assert np.array_equal(got["bucket_off"][nz], want["bucket_off"][nz])
    assert np.array_equal(got["w_count"][:9], want["w_count"]) and got["w_count"][9] == want["ones_in_last"]


@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("var", 31, 3), ("lowcomplex", 21, 2)])
def test_count_in_passes(engine, kind, k, m):
    reads = make_reads(kind, 31)
    pkg = ob.Package(reads, reverse=True)
    want = ob.count(pkg, k, m)
    load(engine, pkg)
    got = passes.count_in_passes(engine, k, m, max_items=-4, batch_bytes=BATCH)
    assert got["n_passes"] >= 3
    assert np.array_equal(got["edges"], want["edges"])
    assert np.array_equal(got["bucket_count"], want["bucket_count"])
    assert np.array_equal(got["hist"], want["hist"])
    assert np.array_equal(got["first_0_out"], want["first_0_out"])
    assert np.array_equal(got["last_0_in"], want["last_0_in"])
    # the filter is off again: a plain call gives the full result
    r = engine.count(k, m)
    assert r.n_items == want["n_items"] and r.n_edges == len(want["edges"])


@pytest.mark.parametrize("kind,k,m,mercy", [("fixed", 21, 2, 0), ("var", 21, 2, 1), ("var", 27, 3, 0), ("var", 27, 2, 2),
                                           ("lowcomplex", 21, 2, 0), ("var", 31, 1, 0)])
def test_read2sdbg_in_passes(engine, kind, k, m, mercy):
    reads = make_reads(kind, 32)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    if m > 1:
        w1 = ob.s1(pkg, k, m, tie_stable=mercy != 2)
        solid = w1["is_solid"]
        if mercy:
            n_want, solid = ob.s2_add_mercy(pkg, k, w1["is_solid"], w1["mercy"])
        want = ob.s2(pkg, k, m, solid)
    else:
        want = ob.s2(pkg, k, 1, None)
    s1, got = passes.read2sdbg_in_passes(engine, k, m, max_items_s1=-3, max_items_s2=-4, need_mercy=mercy, batch_bytes=BATCH)
    if m > 1:
        assert s1["n_passes"] >= 3 and s1["n_items"] == w1["n_items"]
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
        bits = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert np.array_equal(bits, solid[: bits.size])
        if mercy:
            assert s1["n_mercy"] == n_want
            assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
    _check_sdbg(got, want)


@pytest.mark.parametrize("k", [21, 39, 79])
def test_seq2sdbg_in_passes(engine, k):
    reads = make_reads("var", 33)
    cnt = ob.count(ob.Package(reads, reverse=True), k, 2)
    seqs, mult = edges_package(cnt["edges"], k)
    pkg = ob.Package(seqs, reverse=False)
    want = ob.seq2sdbg(pkg, mult, k)
    engine.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
    engine.load_multiplicity(mult)
    got = passes.seq2sdbg_in_passes(engine, k, max_items=-5, batch_bytes=BATCH)
    _check_sdbg(got, want)


@pytest.mark.parametrize("stage,k,m", [("count", 21, 2), ("s1", 21, 2), ("s1_mercy", 27, 2), ("s2", 21, 2), ("seq2sdbg", 29, 0)])
def test_bucket_histogram_equals_oracle_items(engine, stage, k, m):
    """mhx_bucket_histogram (the reference's Lv0CalcBucketSize: kmer_counter.cpp:114-156, read_to_sdbg_s1.cpp:145-206,
    read_to_sdbg_s2.cpp:271-345, seq_to_sdbg.cpp:530-577) = the histogram of the top 16 bits of the oracle's lv2 items"""
    import oracle_binding as ob
    from test_gpu_count import load, make_reads
    if stage == "seq2sdbg":
        from dist_inputs import seqs_with_mult as _seqs_with_mult
        seqs, mult = _seqs_with_mult(9)
        pkg = ob.Package(seqs, reverse=False)
        engine.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
        engine.load_multiplicity(mult)
        items = ob.seq2sdbg_items(pkg, mult, k)
        sid = 4
    else:
        pkg = ob.Package(make_reads("var", 13), reverse=True)
        load(engine, pkg)
        if stage == "count":
            items, sid = ob.count_items(pkg, k), 3
        elif stage in ("s1", "s1_mercy"):
            items, sid = ob.s1_items(pkg, k), (1 if stage == "s1" else 5)
        else:
            s1 = ob.s1(pkg, k, m)
            engine.set_is_solid(s1["is_solid"][: (pkg.start()[-1] + 63) // 64])  # per-occurrence stage 2 (no aggregates)
            items, sid = ob.s2_items(pkg, k, m, s1["is_solid"]), 2
    want = np.bincount(np.asarray(items)[:, 0] >> 16, minlength=65536).astype(np.uint64)
    got = engine.bucket_histogram(sid, k, m)
    assert np.array_equal(np.asarray(got, dtype=np.uint64), want)
