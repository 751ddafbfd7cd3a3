"""GPU: read2sdbg stage 1 on super-k-mer records (round 6, megahit_amd/csrc/s1_skm.hip): the windows of a read that share a minimizer
leave as one 16-byte record, two sort passes order the records by minimizer bin, one workgroup groups a bin in an LDS table with 64-bit
keys.  Against the oracle's Read2SdbgS1 / Read2SdbgS2 (reference src/sorting/read_to_sdbg_s1.cpp:208-464, read_to_sdbg_s2.cpp:521-614):
is_solid, the multiplicity histogram, the item count the reference sorts, and the SdBG stage 2 builds from the aggregated items — every k
the path serves, min count 1 and 2, tables that overflow and split by a second hash, probe limits, libraries whose reads are shorter than
two blocks, low-complexity reads (the path hands the job to the prefix plan and says so), and the shapes it declines."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib, synth
from test_gpu_count import load, make_reads
from test_gpu_round3_knobs import fixed_library
from test_gpu_sdbg import check_sdbg

pytestmark = pytest.mark.gpu

RESET = dict(s1_skm=1, s1_stream_fill=7168, s1_stream_probes=1024, s1_skm_max_bin=65536, s1_skm_min_windows=1 << 22, s1_skm_bin_bits=0, s1_skm_tags=0, s1_skm_cap_pct=36,
             s1_var_min_fill=50, s1_skm_passes=0, s1_skm_deal=1, s1_skm_hp=1)


def run(engine, reads, k, m, opts, want_plan="super-k-mers", want_kernels=("s1_skm_make", "s1_skm_groups"), absent=("s1_groups",), why=None):
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
        plan = engine.last_s1_plan()
        assert plan.startswith(want_plan), plan
        if why:
            assert why in plan, plan
        for kn in want_kernels:
            assert kn in stats, sorted(stats)
        for kn in absent:
            assert kn not in stats, sorted(stats)
        solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert r1.n_items == want1["n_items"]
        assert np.array_equal(solid, want1["is_solid"][: solid.size])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want1["hist"])
        assert r1.n_solid == int(sum(bin(int(x)).count("1") for x in want1["is_solid"]))
        check_sdbg(engine, engine.read2sdbg_s2(k, m), want2)
    finally:
        engine.profile(False)
        for n, v in RESET.items():
            engine.set_option(n, v)


@pytest.mark.parametrize("opts", [dict(), dict(s1_stream_fill=40), dict(s1_stream_fill=3), dict(s1_stream_probes=2), dict(s1_skm_deal=0), dict(s1_skm_deal=0, s1_stream_fill=40)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()) or "default")
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 21, 2), ("short30", 21, 2), ("tiny60", 21, 2), ("pe100", 19, 2), ("repeats100", 22, 2),
                                      ("pe100", 20, 1), ("repeats100", 21, 1)])
def test_stage1_on_super_kmer_records(engine, kind, k, m, opts):
    run(engine, fixed_library(kind, seed=k * 7 + m), k, m, dict(opts, s1_skm=2, s1_skm_max_bin=1 << 30))


def test_a_larger_library_takes_the_path_by_itself(engine):
    """above s1_skm_min_windows the path is the default: 60 000 reads of 100 bases at k = 21"""
    reads = [x for x in synth.gen_pe_reads(20000, 60000, read_len=100, frag=250, err=0.01, seed=3)]
    run(engine, reads, 21, 2, dict(s1_skm_min_windows=1 << 20))


def repeat_reads(n, unit, length=100):
    return [np.tile(np.array(unit, dtype=np.uint8), length // len(unit) + 1)[:length] for _ in range(n)]


def test_low_complexity_reads_go_to_the_prefix_plan(engine):
    """(AC)n reads put every window behind one minimizer: the bin is found too large for one workgroup — in the digit histograms of the
    make kernel already when it is far over the limit, else behind the sort — and the prefix plan (with its giant path) does the stage;
    the plan line says why"""
    reads = fixed_library("pe100", seed=11) + repeat_reads(3000, [0, 1])
    run(engine, reads, 21, 2, dict(s1_skm=2, s1_skm_max_bin=1024), want_plan="stream", want_kernels=("s1_skm_make", "s1_groups"), absent=("s1_skm_groups",),
        why="super-k-mer records given up: a bin of")
    run(engine, reads, 21, 2, dict(s1_skm=2, s1_skm_max_bin=25000), want_plan="stream", want_kernels=("s1_skm_make", "radix_scatter_16B", "s1_groups"),
        absent=("s1_skm_groups",), why="super-k-mer records given up: a bin of")  # (less than 1.5 x the limit: found behind the sort)


def test_the_same_reads_with_the_limit_lifted(engine):
    """... and with the limit lifted one workgroup streams the (AC)n bin: slow, but the same answer"""
    reads = fixed_library("pe100", seed=11) + repeat_reads(3000, [0, 1])
    run(engine, reads, 21, 2, dict(s1_skm=2, s1_skm_max_bin=1 << 30))


@pytest.mark.parametrize("case", ["polyA+polyG", "one window", "m1", "var", "passes", "two windows"])
def test_homopolymer_windows_are_counted_beside_the_records(engine, case):
    """(k+1)-mers of one base — poly-A tails, the poly-G of dark cycles — are ONE key by the million: they never enter a record
    (k_skm_make counts them, k_skm_hp_publish makes the one or two keys they are), so such reads do not cost the path its bins.  A key seen
    once gets its mark, a solid one its aggregated items; with s1_skm_hp = 0 the same library is given up"""
    k, m, opts = 21, 2, dict(s1_skm=2, s1_skm_max_bin=1024)
    reads = fixed_library("pe100", seed=13)
    if case in ("polyA+polyG", "m1", "passes"):
        reads = reads + repeat_reads(3000, [0]) + repeat_reads(700, [2]) + repeat_reads(200, [3]) + repeat_reads(90, [1])
    if case == "m1":
        m = 1
    if case == "passes":
        opts["s1_skm_passes"] = 3
    if case in ("one window", "two windows"):  # exactly k + 1 (k + 2) bases of C inside a read: one (two) windows of one key in the whole job
        rng = np.random.default_rng(4)
        r = rng.integers(0, 4, size=100, dtype=np.uint8)
        n_c = k + 1 if case == "one window" else k + 2
        r[30:30 + n_c] = 1
        r[29], r[30 + n_c] = 0, 3
        reads = reads + [r]
    if case == "var":
        reads = make_reads("var", 21) + repeat_reads(500, [3], length=77) + repeat_reads(300, [1], length=33)
        opts["s1_var_min_fill"] = 5
    run(engine, reads, k, m, opts)
    if case == "polyA+polyG":
        run(engine, reads, k, m, dict(opts, s1_skm_hp=0), want_plan="stream", want_kernels=("s1_skm_make", "s1_groups"), absent=("s1_skm_groups",),
            why="super-k-mer records given up: a bin of")


@pytest.mark.parametrize("opts", [dict(s1_skm_bin_bits=20), dict(s1_skm_bin_bits=18, s1_stream_fill=40), dict(s1_skm_bin_bits=11), dict(s1_skm_tags=1), dict(s1_skm_tags=1, s1_skm_deal=0)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()))
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 22, 2), ("short30", 19, 1)])
def test_three_sort_passes_and_position_tags(engine, kind, k, m, opts):
    """2^17..2^20 bins take a third sort pass (jobs beyond 14 M reads); s1_skm_tags: the kernel of read sets beyond 2^32 bases (tag 0 here;
    tests/test_gpu_fullsize_100M.py has 1.5 x 10^10 bases)"""
    run(engine, fixed_library(kind, seed=k + m), k, m, dict(opts, s1_skm=2, s1_skm_max_bin=1 << 30))


@pytest.mark.parametrize("kind,k,m", [("var", 21, 2), ("var", 19, 1), ("lowcomplex", 22, 2), ("fixed", 21, 2)])
def test_reads_of_several_lengths(engine, kind, k, m):
    """every read takes the blocks of the longest; its place in the store comes from start[] (reads shorter than k + 1 make nothing)"""
    run(engine, make_reads(kind, 11), k, m, dict(s1_skm=2, s1_skm_max_bin=1 << 30, s1_var_min_fill=5))


@pytest.mark.parametrize("kind,k,m,n_passes", [("pe100", 21, 2, 3), ("repeats100", 22, 2, 5), ("var", 21, 2, 2), ("short30", 20, 1, 4)])
def test_passes_over_ranges_of_bins(engine, kind, k, m, n_passes):
    """a job whose record arrays would not fit runs in passes over ranges of the minimizer bins (s1_skm_pass_gb; here forced): every pass
    makes the records of its bins from the reads again; marks, histogram and aggregated items add up"""
    reads = make_reads(kind, 5) if kind == "var" else fixed_library(kind, seed=k)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    engine.set_option("s1_skm", 2)
    engine.set_option("s1_var_min_fill", 5)
    try:
        assert engine.s1_self_planned(k, m) and not engine.s1_self_planned(k, m, want_mercy=True) and not engine.s1_self_planned(k, 3)
    finally:
        engine.set_option("s1_skm", 1)
        engine.set_option("s1_var_min_fill", 50)
    run(engine, reads, k, m, dict(s1_skm=2, s1_skm_max_bin=1 << 30, s1_skm_passes=n_passes, s1_var_min_fill=5), why="%d passes over ranges of bins" % n_passes)


def test_a_caller_that_left_the_plan_to_the_path_hears_when_it_gives_up(engine):
    """s1_skm = 3 (what host/mhx_core.cpp sets after mhx_s1_self_planned said 1): no quiet change to the prefix plan on the whole job"""
    reads = fixed_library("pe100", seed=11) + repeat_reads(3000, [0, 1])
    load(engine, ob.Package(reads, reverse=True))
    try:
        engine.set_option("s1_skm", 3)
        engine.set_option("s1_skm_max_bin", 1024)
        with pytest.raises(lib.MhxError, match="super-k-mer records given up"):
            engine.read2sdbg_s1(21, 2)
    finally:
        for n, v in RESET.items():
            engine.set_option(n, v)
    run(engine, reads, 21, 2, dict(s1_skm=0), want_plan="stream", want_kernels=("s1_groups",), absent=("s1_skm_make",))


def test_the_record_array_is_too_small(engine):
    """s1_skm_cap_pct: the array holds that many records per 100 windows (0.284 per window in random sequence); one that overflows hands
    the job to the prefix plan"""
    run(engine, fixed_library("pe100", seed=8) * 12, 21, 2, dict(s1_skm=2, s1_skm_cap_pct=3), want_plan="stream", want_kernels=("s1_skm_make", "s1_groups"),
        absent=("s1_skm_groups",), why="more records than the array holds")


@pytest.mark.parametrize("k,m,how", [(17, 2, "k"), (23, 2, "k"), (21, 3, "m"), (21, 2, "sparse"), (21, 2, "off")])
def test_shapes_the_path_declines(engine, k, m, how):
    """k outside 19..22, min count beyond 2, a library whose padded blocks would be mostly empty, the knob"""
    reads = make_reads("var", 3) + [np.zeros(2000, dtype=np.uint8)] if how == "sparse" else fixed_library("pe100", seed=k)
    run(engine, reads, k, m, dict(s1_skm=0 if how == "off" else 2), want_plan="", want_kernels=(), absent=("s1_skm_make", "s1_skm_groups"))


def test_mercy_takes_the_sorted_records(engine):
    reads = fixed_library("pe100", seed=2)
    pkg = ob.Package(reads, reverse=True)
    w1 = ob.s1(pkg, 21, 2, tie_stable=True)
    load(engine, pkg)
    try:
        engine.set_option("s1_skm", 2)
        r1 = engine.read2sdbg_s1(21, 2, want_mercy=True)
        assert not engine.last_s1_plan().startswith("super-k-mers")
        assert r1.n_items == w1["n_items"]
        solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert np.array_equal(solid, w1["is_solid"][: solid.size])
    finally:
        engine.set_option("s1_skm", 1)


@pytest.mark.parametrize("world,k,opts", [(2, 21, {}), (3, 22, {"s1_stream_fill": 40}), (3, 21, {"s1_skm_bin_bits": 18}), (2, 19, {"s1_skm_deal": 0}),
                                          (2, 21, {"s1_skm_tags": 1}), (3, 20, {"s1_stream_fill": 3, "s1_skm_bin_bits": 9})])
def test_several_ranks_exchange_records_by_bin(world, k, opts):
    """comm.hip dist_s1_skm: every rank makes the records of its reads (global positions), orders them by bin, sends each owner of a
    range of bins its slice; the owner's group-by reads a bin as one sub-range per sender; marks back to the read owners as lists, the
    aggregated items on to stage 2's exchange.  Ranks as threads on one device; against the oracle on the union of the reads."""
    from test_gpu_comm import run_ranks, load_reads, all_reads, sdbg_of, check_sdbg as check_ranks

    def body(r, e, cm):
        cm.setup(0, k, 2)
        cm.read2sdbg(k, 2)
        r1, r2, _ = cm.read2sdbg(k, 2)  # buffers are reused: same answer the second time
        return sdbg_of(e) + (e.fetch(lib.BUF_MUL_HIST, np.int64), int(r1.n_solid), e.last_s1_plan(), int(r1.n_items), cm.bytes_sent())

    outs = run_ranks(world, load_reads, body, dict(opts, s1_skm=2, s1_skm_max_bin=1 << 30, s1_var_min_fill=5))
    pkg = all_reads(world)
    s1 = ob.s1(pkg, k, 2)
    for o in outs:
        assert o[6].startswith("super-k-mers") and "exchanged by bin" in o[6], o[6]
    assert np.array_equal(sum(o[4] for o in outs), s1["hist"])
    assert sum(o[5] for o in outs) == int(sum(bin(int(x)).count("1") for x in s1["is_solid"]))
    assert sum(o[7] for o in outs) == s1["n_items"]
    # what crossed between the ranks in the two runs — the records of stage 1, the marks back, the items of stage 2 — stays under 8 bytes
    # per stage-1 item of the job (the pre-sorted exchange of the prefix plan moves 12-byte records alone: 12 (N - 1) / N per item)
    assert sum(o[8] for o in outs) <= 2 * 8 * s1["n_items"], (sum(o[8] for o in outs), s1["n_items"])
    check_ranks(outs, ob.s2(pkg, k, 2, s1["is_solid"]))


def test_several_ranks_give_the_path_up_together(engine):
    """one rank holds low-complexity reads: its largest bin is over the limit, all ranks hear of it and take the pre-sorted exchange of
    the prefix plan"""
    from test_gpu_comm import run_ranks, load_fixed_reads, sdbg_of, check_sdbg as check_ranks
    world, k = 2, 21
    reads = [None] * world

    def load_r(r, e):
        reads[r] = load_fixed_reads(r, e)

    def body(r, e, cm):
        cm.setup(0, k, 2)
        cm.read2sdbg(k, 2)
        return sdbg_of(e) + (e.fetch(lib.BUF_MUL_HIST, np.int64), e.last_s1_plan())

    outs = run_ranks(world, load_r, body, dict(s1_skm=2, s1_skm_max_bin=256))
    pkg = ob.Package(reads[0] + reads[1], reverse=True)
    s1 = ob.s1(pkg, k, 2)
    for o in outs:
        assert o[5].startswith("stream") and "pre-sorted exchange" in o[5], o[5]
    assert np.array_equal(sum(o[4] for o in outs), s1["hist"])
    check_ranks(outs, ob.s2(pkg, k, 2, s1["is_solid"]))


# ---- `count` on super-k-mer records (k_skm_make<.., COUNT>, k_count_skm): against the oracle's KmerCounter (kmer_counter.cpp:158-414) ----
def run_count(engine, reads, k, m, opts, want_plan="count: super-k-mers", want_kernels=("count_skm_make", "count_skm_groups"), absent=("count_groups",)):
    pkg = ob.Package(reads, reverse=True)
    want = ob.count(pkg, k, m)
    load(engine, pkg)
    try:
        for n, v in opts.items():
            engine.set_option(n, v)
        engine.profile(True)
        engine.profile_reset()
        r = engine.count(k, m)
        stats = engine.profile_get()
        engine.profile(False)
        plan = engine.last_s1_plan()
        assert plan.startswith(want_plan), plan
        for kn in want_kernels:
            assert kn in stats, sorted(stats)
        for kn in absent:
            assert kn not in stats, sorted(stats)
        assert r.n_items == want["n_items"] and r.words_per_edge == want["wpe"]
        edges = engine.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge)
        assert edges.shape == want["edges"].shape and np.array_equal(edges, want["edges"])
        assert np.array_equal(engine.fetch(lib.BUF_BUCKET_COUNT, np.uint64), want["bucket_count"])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want["hist"])
        assert np.array_equal(engine.fetch(lib.BUF_FIRST_0_OUT, np.uint32), want["first_0_out"])
        assert np.array_equal(engine.fetch(lib.BUF_LAST_0_IN, np.uint32), want["last_0_in"])
    finally:
        engine.profile(False)
        for n, v in dict(RESET, count_skm=1, count_skm_group=2).items():
            engine.set_option(n, v)


@pytest.mark.parametrize("opts", [dict(), dict(s1_stream_fill=40), dict(s1_stream_fill=3), dict(s1_stream_probes=2), dict(s1_skm_bin_bits=18), dict(s1_skm_bin_bits=10),
                                  dict(s1_skm_tags=1), dict(count_skm_group=4)], ids=lambda o: ",".join("%s=%d" % kv for kv in o.items()) or "default")
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 21, 2), ("short30", 21, 2), ("tiny60", 20, 2), ("pe100", 19, 1), ("repeats100", 21, 1)])
def test_count_on_super_kmer_records(engine, kind, k, m, opts):
    run_count(engine, fixed_library(kind, seed=k * 11 + m), k, m, dict(opts, s1_skm=2, s1_skm_max_bin=1 << 30))


@pytest.mark.parametrize("kind,k,m", [("var", 21, 2), ("var", 19, 1), ("lowcomplex", 21, 2), ("fixed", 20, 2)])
def test_count_of_reads_of_several_lengths(engine, kind, k, m):
    run_count(engine, make_reads(kind, 17), k, m, dict(s1_skm=2, s1_skm_max_bin=1 << 30, s1_var_min_fill=5, s1_skm_cap_pct=300))


@pytest.mark.parametrize("case", ["polyA+polyG", "one window", "m1", "var", "passes", "two windows", "read ends"])
def test_count_homopolymer_windows_beside_the_records(engine, case):
    """count: the windows of one base are tallied per class with the bases in front of and behind them (k_skm_make<.., COUNT>), and
    k_count_hp_publish makes the one or two keys they are: histogram entry, packed edge of a solid one (its in- and out-edges are its own
    base).  With s1_skm_hp = 0 the same library gives the path up"""
    k, m, opts = 21, 2, dict(s1_skm=2, s1_skm_max_bin=1024)
    reads = fixed_library("pe100", seed=17)
    if case in ("polyA+polyG", "m1", "passes"):
        reads = reads + repeat_reads(3000, [0]) + repeat_reads(700, [2]) + repeat_reads(200, [3]) + repeat_reads(90, [1])
    if case == "m1":
        m = 1
    if case == "passes":
        opts["s1_skm_passes"] = 3
    if case in ("one window", "two windows"):
        rng = np.random.default_rng(4)
        r = rng.integers(0, 4, size=100, dtype=np.uint8)
        n_c = k + 1 if case == "one window" else k + 2
        r[30:30 + n_c] = 1
        r[29], r[30 + n_c] = 0, 3
        reads = reads + [r]
    if case == "read ends":  # reads that ARE one base from end to end, and one that ends in k + 3 of them: '$' in front of / behind some windows
        rng = np.random.default_rng(6)
        r = rng.integers(0, 4, size=100, dtype=np.uint8)
        r[100 - (k + 3):] = 2
        r[100 - (k + 4)] = 0
        reads = reads + repeat_reads(40, [2]) + [r]
    if case == "var":
        reads = make_reads("var", 21) + repeat_reads(500, [3], length=77) + repeat_reads(300, [1], length=33)
        opts["s1_var_min_fill"] = 5
        opts["s1_skm_cap_pct"] = 300
    if case == "two windows":  # solid, but neither base in front is seen twice: such a key would move first_0_out / last_0_in — the prefix plan does that
        run_count(engine, reads, k, m, opts, want_plan="count: stream", want_kernels=("count_skm_make", "count_skm_groups", "count_groups"), absent=())
        return
    run_count(engine, reads, k, m, opts)
    if case == "polyA+polyG":
        run_count(engine, reads, k, m, dict(opts, s1_skm_hp=0), want_plan="count: stream", want_kernels=("count_skm_make", "count_groups"), absent=("count_skm_groups",))


@pytest.mark.parametrize("how", ["k22", "m3", "off", "repeats"])
def test_count_shapes_outside_the_path(engine, how):
    """k = 22 (no room for the two flanking bases), min count 3, the knob, and reads of a two-base repeat take the prefix plan"""
    k, m, opts = 21, 2, dict(s1_skm=2)
    reads = fixed_library("pe100", seed=5)
    if how == "k22":
        k = 22
    if how == "m3":
        m = 3
    if how == "off":
        opts["count_skm"] = 0
    if how == "repeats":
        reads, opts = reads + repeat_reads(3000, [0, 1]), dict(opts, s1_skm_max_bin=1024)
    low = how == "repeats"
    run_count(engine, reads, k, m, opts, want_plan="count: stream", want_kernels=("count_groups",) + (("count_skm_make",) if low else ()), absent=("count_skm_groups",))


@pytest.mark.parametrize("kind,k,m,n_passes", [("pe100", 21, 2, 3), ("repeats100", 20, 2, 5), ("var", 21, 2, 2), ("short30", 19, 1, 4)])
def test_count_in_passes_over_ranges_of_bins(engine, kind, k, m, n_passes):
    """a job beyond s1_skm_pass_gb of records: every pass makes the records of its bins again, the passes' edges are packed behind each
    other and ordered once; mhx_count_self_planned tells a caller that plans lv1 ranges to leave this one alone"""
    reads = make_reads(kind, 5) if kind == "var" else fixed_library(kind, seed=k)
    load(engine, ob.Package(reads, reverse=True))
    engine.set_option("s1_skm", 2)
    engine.set_option("s1_var_min_fill", 5)
    try:
        assert engine.count_self_planned(k, m) and not engine.count_self_planned(22, m) and not engine.count_self_planned(k, 3)
    finally:
        engine.set_option("s1_skm", 1)
        engine.set_option("s1_var_min_fill", 50)
    run_count(engine, reads, k, m, dict(s1_skm=2, s1_skm_max_bin=1 << 30, s1_skm_passes=n_passes, s1_var_min_fill=5, s1_skm_cap_pct=300))
    assert "%d passes over ranges of bins" % n_passes in engine.last_s1_plan()


def test_count_fails_for_a_caller_that_left_the_plan_to_it(engine):
    reads = fixed_library("pe100", seed=11) + repeat_reads(3000, [0, 1])
    load(engine, ob.Package(reads, reverse=True))
    try:
        engine.set_option("s1_skm", 2)
        engine.set_option("count_skm", 3)
        engine.set_option("s1_skm_max_bin", 1024)
        with pytest.raises(lib.MhxError, match="super-k-mer records given up"):
            engine.count(21, 2)
    finally:
        for n, v in dict(RESET, count_skm=1).items():
            engine.set_option(n, v)
    run_count(engine, reads, 21, 2, dict(count_skm=0), want_plan="count: stream", want_kernels=("count_groups",), absent=("count_skm_make",))


@pytest.mark.parametrize("what", ["stage 1", "count"])
def test_a_library_of_homopolymer_reads_only(engine, what):
    """every window is counted beside the records: not one record is made, ordered or grouped"""
    reads = repeat_reads(300, [0]) + repeat_reads(120, [3]) + repeat_reads(50, [2], length=64)
    if what == "stage 1":
        run(engine, reads, 21, 2, dict(s1_skm=2, s1_var_min_fill=5))
    else:
        run_count(engine, reads, 21, 2, dict(s1_skm=2, s1_var_min_fill=5))
