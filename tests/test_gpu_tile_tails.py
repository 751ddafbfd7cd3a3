"""GPU: the tile kernel (k_tile_groups, tile_groups.h) on groups far longer than a tile — the tail of such a group is SEARCHED (64-ary, the
records are sorted) once it passes kTailWalk records instead of walked 64 records per memory round trip, its run heads are found one
search each, and the whole workgroup does the per-record phases of the tail (round 6: no input may take minutes).  Every engine that
goes through the tile kernel, on libraries with >= 10^5 records of one key and several runs inside one giant group, against the oracle
(reference src/sorting/kmer_counter.cpp:254-381, read_to_sdbg_s1.cpp:368-555, read_to_sdbg_s2.cpp:521-614, seq_to_sdbg.cpp:702-789)."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib
from test_gpu_count import load
from test_gpu_sdbg import check_sdbg, edges_package

pytestmark = pytest.mark.gpu


def giant_library(seed, n_poly=1500, L=100):
    """random reads + poly-A reads (one key in ~n_poly * (L - k) records) + poly-A reads with ONE other base near an end (the same (k-1)-mers
    with other head / tail chars: several runs inside the giant group, sorted behind each other) + a tandem repeat"""
    from megahit_amd import synth
    rng = np.random.default_rng(seed)
    reads = [x for x in synth.gen_pe_reads(600, 2500, read_len=L, frag=220, err=0.01, seed=seed)]
    reads += [np.zeros(L, dtype=np.uint8) for _ in range(n_poly)]
    for base in (1, 2, 3):
        for _ in range(n_poly // 6):
            r = np.zeros(L, dtype=np.uint8)
            r[int(rng.integers(0, 3))] = base
            r[L - 1 - int(rng.integers(0, 3))] = base
            reads.append(r)
    reads += [np.tile(np.array([0, 1], dtype=np.uint8), L // 2) for _ in range(n_poly // 3)]
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


@pytest.mark.parametrize("k,m,opts", [(21, 2, dict(count_stream=0)), (21, 2, dict(count_stream=0, count_seg=0)), (27, 2, dict()), (27, 3, dict(count_seg=0)), (31, 2, dict()), (47, 2, dict())])
def test_count_on_the_tile_path_with_giant_groups(engine, k, m, opts):
    pkg = ob.Package(giant_library(k), reverse=True)
    want = ob.count(pkg, k, m)
    load(engine, pkg)
    try:
        for n, v in opts.items():
            engine.set_option(n, v)
        r = engine.count(k, m)
    finally:
        for n in opts:
            engine.set_option(n, 1)
    assert r.n_items == want["n_items"]
    edges = engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge)
    assert np.array_equal(edges, want["edges"])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want["hist"])
    assert np.array_equal(engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), want["first_0_out"])
    assert np.array_equal(engine.fetch(lib.BUF_LAST_0_IN, np.uint32), want["last_0_in"])


@pytest.mark.parametrize("k,m,mercy,opts", [(21, 2, 1, dict()), (21, 2, 0, dict(s1_seg=0)), (31, 2, 0, dict()), (27, 2, 0, dict(s1_stream=0, s1_seg=0)), (27, 1, 0, dict()), (33, 2, 1, dict())])
def test_read2sdbg_on_the_tile_paths_with_giant_groups(engine, k, m, mercy, opts):
    pkg = ob.Package(giant_library(k + 1), reverse=True)
    load(engine, pkg)
    try:
        for n, v in opts.items():
            engine.set_option(n, v)
        if m > 1:
            w1 = ob.s1(pkg, k, m, tie_stable=True)
            solid = w1["is_solid"]
            r1 = engine.read2sdbg_s1(k, m, want_mercy=mercy)
            assert r1.n_items == w1["n_items"]
            bits = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
            assert np.array_equal(bits, w1["is_solid"][: bits.size])
            assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
            if mercy:
                assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
                n_want, solid = ob.s2_add_mercy(pkg, k, w1["is_solid"], w1["mercy"])
                assert engine.read2sdbg_add_mercy(k) == n_want
            want2 = ob.s2(pkg, k, m, solid)
        else:
            want2 = ob.s2(pkg, k, 1, None)
        check_sdbg(engine, engine.read2sdbg_s2(k, m), want2)
    finally:
        for n in opts:
            engine.set_option(n, 1)


@pytest.mark.parametrize("k", [21, 39])
def test_seq2sdbg_with_giant_groups(engine, k):
    """edges of a low-complexity library: thousands of items of one (k-1)-mer group in seq2sdbg's sort"""
    cnt = ob.count(ob.Package(giant_library(5, n_poly=600), reverse=True), k, 1)
    seqs, mult = edges_package(cnt["edges"], k)
    reps = 40  # the same edges many times over: giant groups of equal items
    seqs, mult = seqs * reps, np.tile(mult, reps)
    pkg = ob.Package(seqs, reverse=False)
    want = ob.seq2sdbg(pkg, mult, k)
    engine.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
    engine.load_multiplicity(mult)
    check_sdbg(engine, engine.seq2sdbg(k), want)
