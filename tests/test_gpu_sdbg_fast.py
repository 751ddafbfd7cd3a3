"""GPU parity of the SdBG emission for 8-byte records (k_sdbg_fast: every run head on its own, s2.hip) against the oracle: the
three item layouts it serves (aggregated stage-2 items, per-occurrence items, seq2sdbg items), with the staged halo shrunk
so that groups and runs reach beyond the window and take the way through memory (galloping run enumeration, wavefront-wide
multiplicity sums), and the generic tile kernel (sdbg_fast = 0) beside it."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib
from test_gpu_count import load, make_reads
from test_gpu_sdbg import check_sdbg, edges_package, repetitive_reads

pytestmark = pytest.mark.gpu

VARIANTS = [dict(), dict(sdbg_fast=0), dict(sdbg_fast_halo=1), dict(sdbg_fast_halo=3), dict(sdbg_fast_halo=17), dict(sdbg_fast_keep=0), dict(sdbg_fast_keep=0, sdbg_fast_halo=2),
            dict(sdbg_fast_tile=1024), dict(sdbg_fast_tile=4096, sdbg_fast_keep=0)]
IDS = lambda o: ",".join("%s=%d" % kv for kv in o.items()) or "default"


def with_options(engine, opts, fn):
    try:
        for name, v in opts.items():
            engine.set_option(name, v)
        return fn()
    finally:
        engine.set_option("sdbg_fast", 1)
        engine.set_option("sdbg_fast_halo", 128)
        engine.set_option("sdbg_fast_keep", 1)
        engine.set_option("sdbg_fast_tile", 2048)
        engine.set_option("s2_agg_from_count", 1)


def many_dummies_reads(seed):
    """runs far longer than a tile and a cap: 70 000 reads that all start with the same 40 bases (their first solid edge yields the
    same '$' item 70 000 times: multiplicity capped at 65535), 3000 poly-A reads, a 300-base genome at high coverage, random reads"""
    rng = np.random.default_rng(seed)
    head = rng.integers(0, 4, size=40, dtype=np.uint8)
    reads = [np.concatenate([head, rng.integers(0, 4, size=20, dtype=np.uint8)]) for _ in range(70000)]
    reads += [np.zeros(80, dtype=np.uint8) for _ in range(3000)]
    return reads + repetitive_reads(seed)


@pytest.mark.parametrize("opts", VARIANTS, ids=IDS)
@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("lowcomplex", 21, 2), ("var", 15, 2), ("lowcomplex", 16, 2), ("rep", 21, 2), ("rep", 22, 3)])
def test_aggregated_items(engine, kind, k, m, opts):
    reads = repetitive_reads(3) if kind == "rep" else make_reads(kind, 8)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m)
    want = ob.s2(pkg, k, m, w1["is_solid"])
    load(engine, pkg)
    engine.read2sdbg_s1(k, m)
    r = with_options(engine, opts, lambda: engine.read2sdbg_s2(k, m))
    check_sdbg(engine, r, want)
    assert r.n_items < want["n_sort_items"] or want["n_sort_items"] == 0


@pytest.mark.parametrize("opts", VARIANTS, ids=IDS)
@pytest.mark.parametrize("kind,k,m", [("var", 21, 1), ("lowcomplex", 27, 1), ("rep", 30, 1), ("fixed", 13, 1), ("rep", 21, 2)])
def test_per_occurrence_items(engine, kind, k, m, opts):
    reads = repetitive_reads(4) if kind == "rep" else make_reads(kind, 9)
    pkg = ob.Package(reads, reverse=True)
    solid = None
    load(engine, pkg)
    if m > 1:  # the bitmap handed in: stage 2 extracts per occurrence
        solid = ob.s1(pkg, k, m)["is_solid"]
        engine.read2sdbg_s1(k, m)
        engine.set_is_solid(solid)
    opts = dict(opts)
    opts.setdefault("sdbg_fast", 2)  # (items per occurrence take the tile kernel by default: 2 = the run-head form anyway)
    opts.setdefault("s2_agg_from_count", 0)  # (round 6: min count 1 would take its solid items from a count of the (k+1)-mers)
    r = with_options(engine, opts, lambda: engine.read2sdbg_s2(k, m))
    check_sdbg(engine, r, ob.s2(pkg, k, m, solid), per_occurrence=True)


@pytest.mark.parametrize("opts", VARIANTS, ids=IDS)
@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("lowcomplex", 19, 2), ("rep", 22, 2)])
def test_seq2sdbg_items(engine, kind, k, m, opts):
    reads = repetitive_reads(5) if kind == "rep" else make_reads(kind, 3)
    pkg = ob.Package(reads, reverse=True)
    cnt = ob.count(pkg, k, m)
    seqs, mult = edges_package(cnt["edges"], k)
    epkg = ob.Package(seqs, reverse=False)
    engine.load_sequences(epkg.words(), epkg.n_seqs, k + 1, None)
    engine.load_multiplicity(mult)
    r = with_options(engine, opts, lambda: engine.seq2sdbg(k))
    check_sdbg(engine, r, ob.seq2sdbg(epkg, mult, k), per_occurrence=True)


@pytest.mark.parametrize("opts", [dict(), dict(sdbg_fast=0), dict(sdbg_fast_halo=2), dict(sdbg_fast_keep=0)], ids=IDS)
def test_runs_beyond_tile_and_cap(engine, opts):
    """one run of 70 000 identical '$' items (its multiplicity sum reaches the 65535 cap inside the wavefront-wide sum), poly-A
    groups spanning tiles: the far path on real shapes, also with the default halo"""
    k, m = 21, 2
    reads = many_dummies_reads(7)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m)
    want = ob.s2(pkg, k, m, w1["is_solid"])
    load(engine, pkg)
    engine.read2sdbg_s1(k, m)
    r = with_options(engine, opts, lambda: engine.read2sdbg_s2(k, m))
    check_sdbg(engine, r, want)
    assert int(want["bucket_large"].sum()) > 0
    # the same reads per occurrence (m = 1: every occurrence an item — runs of 10^5 records; the run-head form forced)
    opts = dict(opts)
    opts.setdefault("sdbg_fast", 2)
    r = with_options(engine, opts, lambda: engine.read2sdbg_s2(k, 1))
    check_sdbg(engine, r, ob.s2(pkg, k, 1, None), per_occurrence=True)
