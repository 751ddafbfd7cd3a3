"""GPU parity: mhx_count (HIP) vs the C oracle on the same seeded reads, through the C ABI."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import synth

pytestmark = pytest.mark.gpu


def make_reads(kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "fixed":
        r = synth.gen_pe_reads(3000, 8000, read_len=100, frag=250, err=0.01, seed=seed)
        return [x for x in r]
    if kind == "var":
        r = synth.gen_pe_reads(2000, 5000, read_len=120, frag=300, err=0.02, seed=seed)
        return [x[: rng.integers(0, 121)] for x in r]
    if kind == "lowcomplex":  # long runs: poly-A, tandem repeats, palindromes
        out = [np.zeros(150, dtype=np.uint8) for _ in range(200)]
        out += [np.tile(np.array([0, 3], dtype=np.uint8), 75) for _ in range(100)]
        out += [np.tile(np.array([0, 1, 2, 3], dtype=np.uint8), 40)[:150] for _ in range(50)]
        out += [rng.integers(0, 4, size=rng.integers(1, 60), dtype=np.uint8) for _ in range(300)]
        return out
    raise ValueError(kind)


def load(engine, pkg):
    start = pkg.start()
    engine.load_sequences(pkg.words(), pkg.n_seqs, 0, start)


@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("var", 21, 2), ("var", 27, 3), ("lowcomplex", 21, 2),
                                      ("fixed", 31, 1), ("var", 32, 2), ("var", 47, 2), ("fixed", 63, 2)])
def test_count_matches_oracle(engine, kind, k, m):
    from megahit_amd import lib
    reads = make_reads(kind, 11)
    pkg = ob.Package(reads, reverse=True)
    want = ob.count(pkg, k, m)
    load(engine, pkg)
    r = engine.count(k, m)
    assert r.n_items == want["n_items"]
    assert r.words_per_edge == want["wpe"]
    edges = engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge)
    assert edges.shape == want["edges"].shape
    assert np.array_equal(edges, want["edges"])
    assert np.array_equal(engine.fetch(lib.BUF_BUCKET_COUNT, np.uint64), want["bucket_count"])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want["hist"])
    assert np.array_equal(engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), want["first_0_out"])
    assert np.array_equal(engine.fetch(lib.BUF_LAST_0_IN, np.uint32), want["last_0_in"])


COUNT_SEG_DEFAULTS = dict(count_seg=1, count_seg_bits=0, count_seg_la=3, count_extract_fixed=1)


@pytest.mark.parametrize("opts", [dict(count_seg=0), dict(count_extract_fixed=0), dict(count_seg_bits=8, count_seg_la=0), dict(count_seg_bits=8), dict(count_seg_bits=16, count_seg_la=1),
                                  dict(count_seg_bits=32)], ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()))
@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("var", 21, 2), ("lowcomplex", 21, 2), ("var", 23, 3), ("fixed", 17, 1)])
def test_count_segment_groupby_variants(engine, kind, k, m, opts):
    """count at k <= 23 = partial sort + segment group-by (k_count_seg) + sort of the solid edges; every knob setting, the
    give-up -> classic fallback and the classic path itself must produce the oracle's outputs."""
    from megahit_amd import lib
    reads = make_reads(kind, 12)
    pkg = ob.Package(reads, reverse=True)
    want = ob.count(pkg, k, m)
    load(engine, pkg)
    try:
        for name, v in opts.items():
            engine.set_option(name, v)
        r = engine.count(k, m)
        assert r.n_items == want["n_items"] and r.n_edges == want["edges"].shape[0]
        edges = engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge)
        assert np.array_equal(edges, want["edges"])
        assert np.array_equal(engine.fetch(lib.BUF_BUCKET_COUNT, np.uint64), want["bucket_count"])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want["hist"])
        assert np.array_equal(engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), want["first_0_out"])
        assert np.array_equal(engine.fetch(lib.BUF_LAST_0_IN, np.uint32), want["last_0_in"])
    finally:
        for name, v in COUNT_SEG_DEFAULTS.items():
            engine.set_option(name, v)


def test_count_empty_and_short(engine):
    from megahit_amd import lib
    pkg = ob.Package([np.zeros(5, dtype=np.uint8), np.zeros(0, dtype=np.uint8), np.ones(21, dtype=np.uint8)], reverse=True)
    load(engine, pkg)
    r = engine.count(21, 2)
    assert r.n_items == 0 and r.n_edges == 0
    assert engine.fetch(lib.BUF_EDGES, np.uint32).size == 0


@pytest.mark.parametrize("kw,aux,n", [(1, 1, 5000), (2, 2, 100000), (2, 0, 77777), (3, 2, 30000), (5, 0, 20000), (9, 0, 5000), (17, 2, 3000)])
def test_sort_records(engine, kw, aux, n):
    rng = np.random.default_rng(kw * 100 + aux)
    items = rng.integers(0, 2 ** 32, size=(n, kw + aux), dtype=np.uint64).astype(np.uint32)
    # many duplicates in the key to exercise stability
    items[:, :kw] &= np.uint32(0x00030003)
    want = ob.sort_items(items, kw, kmsort=False)  # stable
    got = engine.sort_records(items.copy(), kw)
    assert np.array_equal(got, want)
