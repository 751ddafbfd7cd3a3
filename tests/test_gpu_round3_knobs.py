"""GPU: the code paths the round-3 knobs select (include/mhx.h: mhx_get_option; megahit_amd/mhx_tuning.conf may switch any
of them on) reproduce the oracle — Read2SdbgS1 / S2 of the reference (src/sorting/read_to_sdbg_s1.cpp:208-464,
read_to_sdbg_s2.cpp:521-614) — on fixed-length libraries, where the generating first sort pass runs:
  s1_gen_blocked         consecutive items per thread in the generating pass, window words of a whole unit requested up front
  s1_digit_hist_preload  the same in the digit-histogram pre-pass
  s1_stream_read_first   a plain LDS read in front of the compare-and-swap
  sort_rank_uniform      one LDS atomic per record where all records of a wavefront instruction agree on the bits sorted so far
  s1_stream_prefetch     the bucket streaming requests the records of its next trip before it inserts those of the current one"""
import numpy as np
import pytest

import oracle_binding as ob
from megahit_amd import lib, synth
from test_gpu_count import load
from test_gpu_sdbg import check_sdbg

pytestmark = pytest.mark.gpu

KNOBS = ["s1_gen_blocked", "s1_digit_hist_preload", "s1_stream_read_first", "sort_rank_uniform", "s1_stream_prefetch"]
SETTINGS = [{}] + [{k: 1} for k in KNOBS] + [{k: 1 for k in KNOBS}]


def fixed_library(kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "pe100":
        return [x for x in synth.gen_pe_reads(3000, 8000, read_len=100, frag=250, err=0.01, seed=seed)]
    if kind == "repeats100":  # fixed length with poly-A, tandem repeats and exact duplicates: long equal-key runs, hot table slots
        r = [x for x in synth.gen_pe_reads(1500, 3000, read_len=100, frag=220, err=0.005, seed=seed)]
        r += [np.zeros(100, dtype=np.uint8) for _ in range(150)]
        r += [np.tile(np.array([0, 3], dtype=np.uint8), 50) for _ in range(80)]
        r += [np.tile(np.array([0, 1, 2, 3], dtype=np.uint8), 25) for _ in range(40)]
        r += [r[i].copy() for i in rng.integers(0, 1500, size=400)]
        return r
    if kind == "tiny60":  # fewer items than one tile: nearly every thread of the unit is beyond the last item
        return [rng.integers(0, 4, size=60, dtype=np.uint8) for _ in range(24)]
    if kind == "short30":  # 13 slots per read at k = 21: read boundaries inside most threads' runs of eight items
        g = rng.integers(0, 4, size=4000, dtype=np.uint8)
        return [g[s:s + 30].copy() for s in rng.integers(0, 3970, size=5000)]
    raise ValueError(kind)


@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s_: "+".join(sorted(s_)) or "defaults")
@pytest.mark.parametrize("kind,k,m", [("pe100", 21, 2), ("repeats100", 21, 2), ("tiny60", 21, 2), ("short30", 21, 2), ("pe100", 17, 3), ("repeats100", 22, 2)])
def test_read2sdbg_under_every_knob(engine, kind, k, m, setting):
    reads = fixed_library(kind, seed=k * 10 + m)
    pkg = ob.Package(reads, reverse=True)
    load(engine, pkg)
    want1 = ob.s1(pkg, k, m, tie_stable=True)
    want2 = ob.s2(pkg, k, m, want1["is_solid"])
    try:
        for name in KNOBS:
            engine.set_option(name, setting.get(name, 0))
        r1 = engine.read2sdbg_s1(k, m)
        solid = engine.fetch(lib.BUF_IS_SOLID, np.uint64)
        assert r1.n_items == want1["n_items"]
        assert np.array_equal(solid, want1["is_solid"][: solid.size])
        assert np.array_equal(engine.fetch(lib.BUF_MUL_HIST, np.int64), want1["hist"])
        r2 = engine.read2sdbg_s2(k, m)
        check_sdbg(engine, r2, want2)
    finally:
        for name in KNOBS:
            engine.set_option(name, engine_default(engine, name))


def engine_default(engine, name):
    """what a fresh handle would use for `name`: the tuned default of the installation, else the built-in default"""
    from bench import tuned_defaults
    return tuned_defaults().get(name, 1 if name == "sort_rank_uniform" else 0)
