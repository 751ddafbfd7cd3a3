"""GPU parity for read2sdbg (S1, mercy, S2) and seq2sdbg (+ mercy edges) vs the C oracle, through the C ABI."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib
from test_gpu_count import load, make_reads

pytestmark = pytest.mark.gpu

CASES = [("fixed", 21, 2), ("var", 21, 2), ("var", 27, 3), ("lowcomplex", 21, 2), ("var", 32, 2), ("var", 47, 2), ("fixed", 63, 2)]


def check_sdbg(engine, r, want, per_occurrence=False):
    # stage 2 after stage 1 sorts pre-aggregated items (one per solid (k+1)-mer and strand, k <= 22), so its item
    # count only equals the reference's per-occurrence count on the legacy path
    if per_occurrence:
        assert r.n_items == want["n_sort_items"]
    assert r.words_per_tip_label == want["wpt"]
    got = engine.fetch(lib.BUF_SDBG_BYTES, np.uint8)
    assert got.size == want["bytes"].size == r.sdbg_bytes
    assert np.array_equal(got, want["bytes"])
    assert np.array_equal(engine.fetch(lib.BUF_BUCKET_COUNT, np.uint64), want["bucket_items"])
    assert np.array_equal(engine.fetch(lib.BUF_BUCKET_TIPS, np.uint64), want["bucket_tips"])
    assert np.array_equal(engine.fetch(lib.BUF_BUCKET_LARGE, np.uint64), want["bucket_large"])
    nz = want["bucket_items"] > 0
    assert np.array_equal(engine.fetch(lib.BUF_BUCKET_OFFSET, np.uint64)[nz], want["bucket_off"][nz])
    wc = engine.fetch(lib.BUF_W_COUNT, np.uint64)
    assert np.array_equal(wc[:9], want["w_count"]) and wc[9] == want["ones_in_last"]
    assert r.n_sdbg == int(want["bucket_items"].sum()) and r.n_tips == int(want["bucket_tips"].sum())


@pytest.mark.parametrize("kind,k,m", CASES)
def test_read2sdbg_matches_oracle(engine, kind, k, m):
    reads = make_reads(kind, 5)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=True)
    load(engine, pkg)
    r1 = engine.read2sdbg_s1(k, m, want_mercy=True)
    assert r1.n_items == w1["n_items"]
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(solid, w1["is_solid"][: solid.size])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
    assert r1.n_solid == int(sum(bin(int(x)).count("1") for x in w1["is_solid"]))
    mercy = engine.fetch(lib.BUF_MERCY_CAND, np.int64)
    assert np.array_equal(mercy, w1["mercy"])
    # S2 without mercy
    r2 = engine.read2sdbg_s2(k, m)
    check_sdbg(engine, r2, ob.s2(pkg, k, m, w1["is_solid"]))
    # mercy block, then S2 again
    n_want, solid_want = ob.s2_add_mercy(pkg, k, w1["is_solid"], w1["mercy"])
    n_got = engine.read2sdbg_add_mercy(k)
    assert n_got == n_want
    assert np.array_equal(engine.fetch(lib.BUF_IS_SOLID, np.uint64), solid_want[: solid.size])
    r3 = engine.read2sdbg_s2(k, m)
    check_sdbg(engine, r3, ob.s2(pkg, k, m, solid_want), per_occurrence=True)  # after mercy: per-occurrence path


@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("var", 21, 2), ("lowcomplex", 21, 2), ("var", 27, 3), ("var", 47, 2)])
def test_read2sdbg_reference_exact_tie_order(engine, kind, k, m):
    """want_mercy=2 replays kmlib::kmsort per bucket: mercy candidates equal the oracle's kmsort mode, which is
    pinned to the reference (H1).  'fixed' and 'lowcomplex' have buckets far larger than 64 records."""
    reads = make_reads(kind, 5)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=False)
    load(engine, pkg)
    r1 = engine.read2sdbg_s1(k, m, want_mercy=2)
    assert r1.n_items == w1["n_items"]
    assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(solid, w1["is_solid"][: solid.size])
    n_want, solid_want = ob.s2_add_mercy(pkg, k, w1["is_solid"], w1["mercy"])
    assert engine.read2sdbg_add_mercy(k) == n_want
    check_sdbg(engine, engine.read2sdbg_s2(k, m), ob.s2(pkg, k, m, solid_want))


def repetitive_reads(seed):
    """buckets of every size class of the kmsort replay: a 300-base genome at ~1000x (buckets of thousands of records, runs of
    equal keys far longer than 64), 2500 poly-A reads (one bucket of ~2 x 10^5 records), random reads with errors"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=300, dtype=np.uint8)
    reads = []
    for _ in range(3000):
        a = int(rng.integers(0, 200))
        r = g[a:a + 100].copy()
        e = rng.random(100) < 0.01
        r[e] = rng.integers(0, 4, size=int(e.sum()), dtype=np.uint8)
        reads.append(r if rng.random() < 0.5 else (3 - r[::-1]).astype(np.uint8))
    reads += [np.zeros(100, dtype=np.uint8) for _ in range(2500)]
    reads += [rng.integers(0, 4, size=int(rng.integers(30, 120)), dtype=np.uint8) for _ in range(1000)]
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


def size_class_reads(seed):
    """lv1 buckets of about 10^5 records (the replay's largest LDS class, 69 633..147 456 records) and of ~5 x 10^4: 1300
    poly-A reads and 2600 (AC)n reads, whose (k-1)-mers fall into one and two buckets, with sequencing errors sprinkled in so
    that the buckets also hold short runs and singletons, plus random reads"""
    rng = np.random.default_rng(seed)
    reads = []
    for _ in range(1300):
        r = np.zeros(100, dtype=np.uint8)
        e = rng.random(100) < 0.002
        r[e] = rng.integers(0, 4, size=int(e.sum()), dtype=np.uint8)
        reads.append(r)
    for i in range(2600):
        r = np.tile(np.array([0, 1], dtype=np.uint8), 50)[i % 2:][:98].copy()
        e = rng.random(r.size) < 0.002
        r[e] = rng.integers(0, 4, size=int(e.sum()), dtype=np.uint8)
        reads.append(r)
    reads += [rng.integers(0, 4, size=int(rng.integers(40, 120)), dtype=np.uint8) for _ in range(800)]
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


@pytest.mark.parametrize("k,m", [(21, 2), (27, 3)])
def test_kmsort_replay_largest_lds_class(engine, k, m):
    reads = size_class_reads(5)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=False)
    load(engine, pkg)
    sizes = np.sort(engine.bucket_histogram(lib.STAGE_S1_MERCY, k, m))[::-1]
    assert 69632 < sizes[0] <= 147456, sizes[:4]  # the class under test is really met
    r1 = engine.read2sdbg_s1(k, m, want_mercy=2)
    assert r1.n_items == w1["n_items"]
    assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(solid, w1["is_solid"][: solid.size])


@pytest.mark.parametrize("k,m,legacy", [(21, 2, 0), (21, 2, 1), (31, 3, 0), (31, 3, 1), (61, 2, 0)])
def test_kmsort_replay_on_repetitive_reads(engine, k, m, legacy):
    """the wave-per-bucket replay (tags in LDS / in global memory for the poly-A bucket) and the one-thread-per-bucket
    replay on whole records both leave the oracle's kmsort order"""
    reads = repetitive_reads(77)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=False)
    load(engine, pkg)
    engine.set_option("kmsort_emu_legacy", legacy)
    try:
        r1 = engine.read2sdbg_s1(k, m, want_mercy=2)
    finally:
        engine.set_option("kmsort_emu_legacy", 0)
    assert r1.n_items == w1["n_items"]
    assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(solid, w1["is_solid"][: solid.size])


@pytest.mark.parametrize("kind,k,m", CASES + [("var", 11, 2), ("fixed", 13, 2), ("fixed", 14, 3)])
def test_read2sdbg_s1_without_mercy_compact_records(engine, kind, k, m):
    """want_mercy=False takes the compact-record path (12-byte stage-1 items at k <= 29)."""
    reads = make_reads(kind, 6)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=True)
    load(engine, pkg)
    r1 = engine.read2sdbg_s1(k, m, want_mercy=False)
    assert r1.n_items == w1["n_items"] and r1.n_mercy_cand == 0
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(solid, w1["is_solid"][: solid.size])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
    assert r1.n_solid == int(sum(bin(int(x)).count("1") for x in w1["is_solid"]))
    check_sdbg(engine, engine.read2sdbg_s2(k, m), ob.s2(pkg, k, m, w1["is_solid"]))


SEG_DEFAULTS = dict(s1_seg=1, s1_seg_bits=0, s1_seg_la=3, s1_seg_per=8, s1_stream=1, s1_stream_max=40000, s1_stream_probes=1024, s1_stream_direct=1, s1_pack_fixed=1)
SEG_VARIANTS = [dict(s1_seg=0),                      # classic: full sort + tile kernel
                dict(s1_stream=0),                  # tile kernel (k_s1_seg) instead of bucket streaming (k_s1_stream)
                dict(s1_stream_direct=0),           # bucket streaming with the second read of the bucket instead of marks from the table
                dict(s1_stream_probes=0),           # every bucket 'overflows' its table: stream -> tile kernel fallback
                dict(s1_stream=0, s1_seg_la=0, s1_seg_bits=8),  # tile kernel gives up -> classic
                dict(s1_seg_bits=8, s1_seg_la=0),   # segments of ~1000 records with no look-ahead: tiles give up -> fallback
                dict(s1_seg_bits=8),                # long segments, several look-ahead chunks
                dict(s1_seg_bits=16, s1_seg_la=1),
                dict(s1_seg_bits=32),               # prefix = the whole first key word
                dict(s1_seg_per=4),
                dict(s1_seg_per=4, s1_seg_bits=8),
                dict(s1_pack_fixed=0)]               # reads of one length: the general byte-map -> bitmap kernel


@pytest.mark.parametrize("opts", SEG_VARIANTS, ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()))
@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("var", 21, 2), ("lowcomplex", 21, 2), ("var", 27, 3), ("fixed", 25, 2)])
def test_s1_segment_groupby_variants(engine, kind, k, m, opts):
    """The no-mercy stage 1 = partial sort + segment group-by (k_s1_seg); every knob setting, the give-up -> classic
    fallback and the classic path itself must produce the oracle's is_solid / histogram / SdBG."""
    reads = make_reads(kind, 8)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=True)
    load(engine, pkg)
    try:
        for name, v in opts.items():
            engine.set_option(name, v)
        r1 = engine.read2sdbg_s1(k, m, want_mercy=False)
        assert r1.n_items == w1["n_items"]
        solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert np.array_equal(solid, w1["is_solid"][: solid.size])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
        assert r1.n_solid == int(sum(bin(int(x)).count("1") for x in w1["is_solid"]))
        check_sdbg(engine, engine.read2sdbg_s2(k, m), ob.s2(pkg, k, m, w1["is_solid"]))
    finally:
        for name, v in SEG_DEFAULTS.items():
            engine.set_option(name, v)


@pytest.mark.parametrize("kind,k", [("var", 21), ("lowcomplex", 27), ("fixed", 31)])
def test_read2sdbg_min_count_1(engine, kind, k):
    reads = make_reads(kind, 9)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    want = ob.s2(pkg, k, 1, None)
    try:
        for from_count in (0, 1):  # (round 6: the solid items from a count of the (k+1)-mers, k <= 27 — fewer items than occurrences)
            engine.set_option("s2_agg_from_count", from_count)
            r = engine.read2sdbg_s2(k, 1)  # for_sure_solid: S1 is skipped (main_sdbg_build.cpp:142-146)
            check_sdbg(engine, r, want, per_occurrence=not from_count)
    finally:
        engine.set_option("s2_agg_from_count", 1)


def edges_package(edges, k):
    """count output -> (package of (k+1)-mers, multiplicities) as EdgeReader does."""
    seqs = []
    wpe = edges.shape[1]
    for e in edges:
        bases = np.empty(k + 1, dtype=np.uint8)
        for j in range(k + 1):
            bases[j] = (int(e[j >> 4]) >> (30 - 2 * (j & 15))) & 3
        seqs.append(bases)
    mult = (edges[:, wpe - 1] & 0xFFFF).astype(np.uint16) if len(edges) else np.zeros(0, dtype=np.uint16)
    return seqs, mult


@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("var", 27, 2), ("var", 47, 2)])
def test_seq2sdbg_from_edges_with_mercy(engine, kind, k, m):
    reads = make_reads(kind, 3)
    pkg = ob.Package(reads, reverse=True)
    cnt = ob.count(pkg, k, m)
    seqs, mult = edges_package(cnt["edges"], k)
    epkg = ob.Package(seqs, reverse=False)
    engine.load_sequences(epkg.words(), epkg.n_seqs, k + 1, None)
    engine.load_multiplicity(mult)
    r = engine.seq2sdbg(k)
    check_sdbg(engine, r, ob.seq2sdbg(epkg, mult, k), per_occurrence=True)
    # mercy: candidate reads as KmerCounter::Lv0Postprocess selects them (kmer_counter.cpp:390-403)
    first, last = cnt["first_0_out"], cnt["last_0_in"]
    sel = [i for i in range(len(reads)) if first[i] != 0xFFFFFFFF and last[i] != 0xFFFFFFFF and last[i] > first[i]]
    cand = ob.Package([reads[i][::-1] for i in sel], reverse=False)  # .cand holds the reversed reads
    n_want, mult2 = ob.gen_mercy_edges(epkg, mult, cand, k)
    engine.load_sequences(ob.Package(seqs, reverse=False).words(), len(seqs), k + 1, None)
    engine.load_multiplicity(mult)
    n_got = engine.gen_mercy_edges(k, cand.words(), cand.n_seqs, cand.start())
    assert n_got == n_want
    r2 = engine.seq2sdbg(k)
    check_sdbg(engine, r2, ob.seq2sdbg(epkg, mult2, k), per_occurrence=True)


@pytest.mark.parametrize("k", [21, 29, 39, 59, 79, 99, 119, 141, 255])
def test_seq2sdbg_wide_keys(engine, k):
    """contig-like inputs at the k-list of BASELINE config 4 (item widths 2..9 words) and kmax."""
    rng = np.random.default_rng(k)
    genome = rng.integers(0, 4, size=6000, dtype=np.uint8)
    seqs = []
    for _ in range(120):
        a = rng.integers(0, 5000)
        L = rng.integers(k - 3, 900)
        s = genome[a:a + L].copy()
        if rng.random() < 0.5:
            s = (3 - s)[::-1]
        seqs.append(s)
    seqs.append(np.zeros(k + 1, dtype=np.uint8))
    seqs.append(np.zeros(3, dtype=np.uint8))
    mult = rng.integers(0, 70000, size=len(seqs)).clip(0, 65535).astype(np.uint16)
    mult[:5] = [0, 1, 254, 255, 65535]
    pkg = ob.Package(seqs, reverse=True)
    engine.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
    engine.load_multiplicity(mult)
    r = engine.seq2sdbg(k)
    check_sdbg(engine, r, ob.seq2sdbg(pkg, mult, k), per_occurrence=True)


@pytest.mark.parametrize("kind,k,m", [("fixed", 21, 2), ("var", 21, 3), ("lowcomplex", 21, 2), ("var", 15, 2), ("fixed", 22, 2), ("lowcomplex", 16, 2)])
def test_s2_aggregated_equals_per_occurrence(engine, kind, k, m, monkeypatch):
    """k <= 22: stage 2 fed by stage 1's aggregated items must give the reference's records (incl. tips labels,
    multiplicities > 254 in the poly-A reads) and must sort far fewer items."""
    reads = make_reads(kind, 8)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m)
    want = ob.s2(pkg, k, m, w1["is_solid"])
    load(engine, pkg)
    engine.read2sdbg_s1(k, m)
    r = engine.read2sdbg_s2(k, m)
    check_sdbg(engine, r, want)
    assert r.n_items < want["n_sort_items"] or want["n_sort_items"] == 0
    # replacing the bitmap invalidates the aggregated items -> per-occurrence path, same answer
    engine.set_is_solid(w1["is_solid"])
    r = engine.read2sdbg_s2(k, m)
    check_sdbg(engine, r, want, per_occurrence=True)


@pytest.mark.parametrize("polarity", ["solid", "nonsolid", None])
@pytest.mark.parametrize("kind,k,m,mercy", [("fixed", 21, 2, 0), ("var", 21, 2, 1), ("lowcomplex", 21, 2, 0), ("var", 27, 3, 0),
                                           ("var", 47, 2, 1), ("short", 21, 2, 0)])
def test_s1_marking_polarity(engine, kind, k, m, mercy, polarity, monkeypatch):
    """is_solid is built from a byte map that marks either the solid occurrences or (when a sample says most are
    solid) the non-solid ones; both polarities and the sampled choice give the reference's bitmap."""
    if polarity is None:
        monkeypatch.delenv("MHX_S1_MARK", raising=False)
    else:
        monkeypatch.setenv("MHX_S1_MARK", polarity)
    if kind == "short":  # reads shorter than k+1 and of exactly k+1 among longer ones: no edge may be marked there
        rng = np.random.default_rng(11)
        g = rng.integers(0, 4, size=3000, dtype=np.uint8)
        reads = []
        for _ in range(4000):
            L = int(rng.choice([5, k - 1, k, k + 1, k + 2, 60, 100]))
            o = int(rng.integers(0, g.size - L))
            reads.append(g[o:o + L].copy())
    else:
        reads = make_reads(kind, 9)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=True)
    load(engine, pkg)
    r1 = engine.read2sdbg_s1(k, m, want_mercy=mercy)
    assert r1.n_items == w1["n_items"]
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(solid, w1["is_solid"][: solid.size])
    assert r1.n_solid == int(sum(bin(int(x)).count("1") for x in w1["is_solid"]))
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
    if mercy:
        assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
    check_sdbg(engine, engine.read2sdbg_s2(k, m), ob.s2(pkg, k, m, w1["is_solid"]))


@pytest.mark.parametrize("k,m,mercy", [(21, 2, 0), (21, 2, 1), (27, 2, 0), (32, 3, 0), (47, 2, 1)])
def test_read2sdbg_fixed_length_reads(engine, k, m, mercy):
    """Reads of one length loaded without a start array (fixed_len): the flattened extraction path."""
    reads = make_reads("fixed", 21)
    assert len({len(r) for r in reads}) == 1
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=True)
    engine.load_sequences(pkg.words(), pkg.n_seqs, len(reads[0]), None)
    r1 = engine.read2sdbg_s1(k, m, want_mercy=mercy)
    assert r1.n_items == w1["n_items"]
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(solid, w1["is_solid"][: solid.size])
    if mercy:
        assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
    check_sdbg(engine, engine.read2sdbg_s2(k, m), ob.s2(pkg, k, m, w1["is_solid"]))
    want = ob.count(pkg, k, m)
    r = engine.count(k, m)
    assert np.array_equal(engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge), want["edges"])
    assert np.array_equal(engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), want["first_0_out"])
    assert np.array_equal(engine.fetch(lib.BUF_LAST_0_IN, np.uint32), want["last_0_in"])


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["shared-cursor", "regions", "regions-then-fallback"])
@pytest.mark.parametrize("kind,k,m,exact", [("fixed", 21, 2, False), ("var", 27, 3, False), ("lowcomplex", 21, 2, True)])
def test_mercy_candidates_from_per_workgroup_regions(engine, kind, k, m, exact, mode):
    """stage 1 with mercy candidates: the candidates leave the tile kernel through per-workgroup regions (default), through
    the shared cursor (0), or through the regions followed by the forced way back (2) — the same sorted list every time"""
    reads = make_reads(kind, 11)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, k, m, tie_stable=not exact)
    load(engine, pkg)
    engine.set_option("s1_mercy_regions", mode)
    try:
        engine.read2sdbg_s1(k, m, want_mercy=2 if exact else 1)
    finally:
        engine.set_option("s1_mercy_regions", 1)
    assert np.array_equal(engine.fetch(lib.BUF_MERCY_CAND, np.int64), w1["mercy"])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(solid, w1["is_solid"][: solid.size])
