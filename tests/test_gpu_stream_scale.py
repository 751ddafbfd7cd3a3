"""GPU: stage 1's bucket streaming at any job size (Read2SdbgS1::Lv2Postprocess without mercy, reference
src/sorting/read_to_sdbg_s1.cpp:368-464; the lv1 passes of src/sorting/base_engine.cpp:254-281) against the oracle:

  * every prefix width of the plan (16..24 bits: two or three LSD passes) and every number of sub-rounds per bucket,
    alone and together with the prefetching kernel, on fixed-length libraries (the generating first pass) and on
    variable-length ones (loaded passes);
  * buckets that overflow their LDS table split themselves inside the kernel (s1_stream_fill makes every bucket do
    it); only a probe limit of 0 still sends the stage to the tile kernel;
  * the plan the library makes for a given density (records per lv1 bucket): s1_stream_max scales the thresholds
    down to fixture size;
  * position tags: a position word of a few bits makes every record carry a tag, on the stream, tile and classic paths;
  * the bucket filter of a memory plan applied inside the generating pass (and the batch path next to it);
  * the lv1 histogram taken straight from the packed reads."""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib, passes
from test_gpu_count import load, make_reads
from test_gpu_passes import _check_sdbg
from test_gpu_round3_knobs import fixed_library
from test_gpu_sdbg import check_sdbg

pytestmark = pytest.mark.gpu

DEFAULTS = dict(s1_stream_bits=0, s1_stream_sub0=-1, s1_stream_prefetch=0, s1_stream_fill=7168, s1_stream_probes=1024, s1_stream_max=40000,
                s1_stream_sub_max=1, s1_stream_direct=1, s1_pos_bits=0, s1_stream=1, s1_seg=1, s1_filter_in_gen=1,
                s1_stream_unroll=4, s1_gen_blocked=1)


class knobs:
    def __init__(self, engine, **kw):
        self.e, self.kw = engine, kw

    def __enter__(self):
        for n, v in self.kw.items():
            self.e.set_option(n, v)

    def __exit__(self, *a):
        for n in self.kw:
            self.e.set_option(n, DEFAULTS[n])


def library(kind, seed):
    if kind in ("fixed", "var", "lowcomplex"):
        return make_reads(kind, seed)
    return fixed_library(kind, seed)


def check_read2sdbg(engine, pkg, k, m, plan_has=None):
    want1 = ob.s1(pkg, k, m, tie_stable=True)
    want2 = ob.s2(pkg, k, m, want1["is_solid"])
    r1 = engine.read2sdbg_s1(k, m)
    if plan_has:
        for part in plan_has:
            assert part in engine.last_s1_plan(), engine.last_s1_plan()
    solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert r1.n_items == want1["n_items"]
    assert np.array_equal(solid, want1["is_solid"][: solid.size])
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want1["hist"])
    check_sdbg(engine, engine.read2sdbg_s2(k, m), want2)


@pytest.mark.parametrize("prefetch", [0, 1])
@pytest.mark.parametrize("bits,sub0", [(0, -1), (17, -1), (20, 0), (24, 0), (16, 1), (16, 3), (19, 2)])
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 21, 2), ("tiny60", 21, 2), ("short30", 21, 2), ("var", 21, 2), ("pe100", 17, 3),
                                      ("repeats100", 22, 2), ("pe100", 15, 2), ("lowcomplex", 16, 2)])
def test_every_prefix_width_and_sub_round_count(engine, kind, k, m, bits, sub0, prefetch):
    pkg = ob.Package(library(kind, k * 10 + m), reverse=True)
    load(engine, pkg)
    eff = min(bits or 16, 2 * (k - 1))
    with knobs(engine, s1_stream_bits=bits, s1_stream_sub0=sub0, s1_stream_prefetch=prefetch):
        check_read2sdbg(engine, pkg, k, m, plan_has=["stream p%d " % eff, "%d passes" % ((eff + 7) // 8)])


@pytest.mark.parametrize("opts", [dict(s1_stream_fill=1), dict(s1_stream_fill=3, s1_stream_prefetch=1), dict(s1_stream_fill=16),
                                  dict(s1_stream_fill=5, s1_stream_direct=0), dict(s1_stream_fill=2, s1_stream_sub0=2, s1_stream_bits=18),
                                  dict(s1_stream_probes=0)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in sorted(o.items())))
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 21, 2), ("var", 21, 3), ("lowcomplex", 21, 2)])
def test_overflowing_buckets_split_themselves(engine, kind, k, m, opts):
    """s1_stream_fill = n: a round gives up once it has claimed n slots, so every bucket with more than n distinct keys is taken
    in sub-rounds of halved key ranges until each round holds at most n (down to one key per round); probes = 0: nothing can be
    inserted at all, the one case left in which the stage falls back to the tile kernel"""
    pkg = ob.Package(library(kind, 77), reverse=True)
    load(engine, pkg)
    with knobs(engine, **opts):
        check_read2sdbg(engine, pkg, k, m, plan_has=["stream"])


@pytest.mark.parametrize("cap", [40000, 6, 3, 1])
def test_the_plan_follows_the_density(engine, cap):
    """7.6 records per lv1 bucket in this library (498 000 items): with s1_stream_max = 6 the buckets are twice too large (one
    level of sub-rounds, still two passes), with 3 four times and with 1 eight times (three passes; their width aims at half
    the density: 19 prefix bits both): the plan is a pure function of the density and the knobs"""
    pkg = ob.Package(library("pe100", 5), reverse=True)
    load(engine, pkg)
    with knobs(engine, s1_stream_max=cap):
        check_read2sdbg(engine, pkg, 21, 2)
        per_bucket = 6000 * (100 - 21 + 4) / 65536.0
        need = 0
        while per_bucket > cap * (1 << need):
            need += 1
        plan = engine.last_s1_plan()
        if need == 0:
            assert plan.startswith("stream p16 sub0 2 passes"), plan
        elif need == 1:
            assert plan.startswith("stream p16 sub1 2 passes"), plan
        else:  # three passes whatever the width: the width aims at buckets half as full (s1_stream_max3 = s1_stream_max / 2)
            cap3, need3 = max(1, cap // 2), 0
            while per_bucket > cap3 * (1 << need3):
                need3 += 1
            assert plan.startswith("stream p%d sub0 3 passes" % (16 + max(need3, 1))), plan


@pytest.mark.parametrize("opts", [dict(), dict(s1_stream_direct=0), dict(s1_stream=0), dict(s1_seg=0), dict(s1_stream_bits=20, s1_stream_prefetch=1), dict(s1_gen_blocked=0)],
                         ids=lambda o: ",".join("%s=%d" % kv for kv in sorted(o.items())) or "defaults")
@pytest.mark.parametrize("kind,k,m,pos_bits", [("pe100", 21, 2, 13), ("repeats100", 21, 2, 12), ("var", 21, 2, 12), ("short30", 22, 2, 11), ("pe100", 17, 3, 14)])
def test_position_tags(engine, kind, k, m, pos_bits, opts):
    """the third record word holds only the low `pos_bits` bits of a position, the rest rides as a tag in the key words:
    what happens to every record of a read set beyond 2^32 bases (100 M reads on one GPU), at fixture size"""
    reads = library(kind, 31)
    pkg = ob.Package(reads, reverse=True)
    assert (int(pkg.start()[-1]) >> pos_bits) >= 2  # the tags are really used ...
    assert (int(pkg.start()[-1]) >> pos_bits) < 256  # ... and fit their 8 bits
    load(engine, pkg)
    with knobs(engine, s1_pos_bits=pos_bits, **opts):
        check_read2sdbg(engine, pkg, k, m)


@pytest.mark.parametrize("in_gen", [1, 0])
@pytest.mark.parametrize("kind,k,m,opts", [("fixed", 21, 2, {}), ("pe100", 21, 2, dict(s1_stream_bits=19)), ("repeats100", 22, 2, dict(s1_pos_bits=12)),
                                           ("short30", 21, 2, dict(s1_gen_blocked=0)), ("tiny60", 21, 2, {}), ("pe100", 17, 3, dict(s1_stream_fill=4))])
def test_bucket_filter_inside_the_generating_pass(engine, kind, k, m, in_gen, opts):
    """stage 1 in bucket-range passes (passes.read2sdbg_in_passes = what mhx_core's memory plan does): with s1_filter_in_gen the
    first sort pass of each range makes only the records of the kept buckets; without it the extraction batches + split run"""
    pkg = ob.Package(library(kind, 32), reverse=True)
    load(engine, pkg)
    w1 = ob.s1(pkg, k, m, tie_stable=True)
    want = ob.s2(pkg, k, m, w1["is_solid"])
    with knobs(engine, s1_filter_in_gen=in_gen, **opts):
        s1, got = passes.read2sdbg_in_passes(engine, k, m, max_items_s1=-3, max_items_s2=-4, need_mercy=0, batch_bytes=8 << 20)
    assert s1["n_passes"] >= 3 or kind == "tiny60"
    assert s1["n_items"] == w1["n_items"]
    assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"])
    bits = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
    assert np.array_equal(bits, w1["is_solid"][: bits.size])
    got["n_passes"] = max(got["n_passes"], 3)
    _check_sdbg(got, want)


@pytest.mark.parametrize("kind,k", [("pe100", 21), ("repeats100", 22), ("short30", 21), ("tiny60", 21), ("pe100", 12)])
def test_fast_bucket_histogram(engine, kind, k):
    """mhx_bucket_histogram of stage 1 on reads of one length comes straight from the packed reads (two launches, one half of
    the bucket space each) = the histogram of the oracle's items (Lv0CalcBucketSize, read_to_sdbg_s1.cpp:145-206)"""
    pkg = ob.Package(library(kind, 9), reverse=True)
    load(engine, pkg)
    items = np.asarray(ob.s1_items(pkg, k))
    want = np.bincount(items[:, 0] >> 16, minlength=65536).astype(np.uint64)
    for fast in (1, 0):
        engine.set_option("s1_bucket_hist_fast", fast)
        got = np.asarray(engine.bucket_histogram(1, k, 2), dtype=np.uint64)
        assert np.array_equal(got, want), fast
    engine.set_option("s1_bucket_hist_fast", 1)


def test_one_huge_low_complexity_bucket(engine):
    """a planted poly-A / dinucleotide load: hundreds of thousands of records with ONE key in one bucket (one lane inserts for a
    wavefront whose records all carry the same key) next to ordinary reads; the result is the oracle's"""
    rng = np.random.default_rng(3)
    reads = [x for x in fixed_library("pe100", 8)]
    reads += [np.zeros(100, dtype=np.uint8) for _ in range(6000)]
    reads += [np.tile(np.array([0, 3], dtype=np.uint8), 50) for _ in range(3000)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    for pf in (0, 1):
        with knobs(engine, s1_stream_prefetch=pf):
            check_read2sdbg(engine, pkg, 21, 2, plan_has=["stream"])
