"""GPU: tuned defaults (mhx_tuning.conf beside libmhx.so, include/mhx.h: mhx_get_option) — a handle takes a knob from its
explicit options, then the environment, then the file, then the built-in default; the file only ever selects between code
paths with identical results (checked for the knobs it may carry in tests/test_gpu_sort_unit_runs.py)."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


def test_precedence_of_option_environment_file_default(tmp_path, monkeypatch):
    from megahit_amd import lib
    conf = tmp_path / "mhx_tuning.conf"
    conf.write_text("# comment\ns1_stream_prefetch = 1\ns1_stream_unroll 2   # trailing comment\nbroken line here\n")
    monkeypatch.setenv("MHX_TUNING_FILE", str(conf))
    monkeypatch.delenv("MHX_S1_STREAM_PREFETCH", raising=False)
    monkeypatch.delenv("MHX_NO_TUNING", raising=False)
    e = lib.Engine(0)
    try:
        assert e.get_option("s1_stream_prefetch", 0) == 1
        assert e.get_option("s1_stream_unroll", 4) == 2
        assert e.get_option("no_such_knob", 7) == 7
        monkeypatch.setenv("MHX_S1_STREAM_UNROLL", "8")
        assert e.get_option("s1_stream_unroll", 4) == 8
        e.set_option("s1_stream_unroll", 1)
        assert e.get_option("s1_stream_unroll", 4) == 1
    finally:
        e.close()
    monkeypatch.setenv("MHX_NO_TUNING", "1")
    e = lib.Engine(0)
    try:
        assert e.get_option("s1_stream_prefetch", 0) == 0
    finally:
        e.close()


@pytest.mark.parametrize("kw,aux,n", [(2, 0, 700001), (2, 1, 900001), (3, 1, 500001), (5, 1, 200001)])
def test_general_sorts_are_stable_by_construction(engine, kw, aux, n):
    """every pass of a general sort ranks with the match-any ballots (sort_rank_uniform only touches plans that declare the
    bits sorted before a pass, which mhx_sort_records never does) — many equal keys, aux words differ: stability shows"""
    rng = np.random.default_rng(n)
    items = rng.integers(0, 2 ** 32, size=(n, kw + aux), dtype=np.uint64).astype(np.uint32)
    items[:, :kw] &= np.uint32(0x00FF00FF)  # few distinct digit values: long same-address queues inside every wavefront
    want = ob.sort_items(items, kw, kmsort=False)
    engine.set_option("sort_hybrid", 0)
    try:
        for v in (1, 0):
            engine.set_option("sort_rank_uniform", v)
            assert np.array_equal(engine.sort_records(items.copy(), kw), want)
    finally:
        engine.set_option("sort_rank_uniform", 1)
        engine.set_option("sort_hybrid", 1)
