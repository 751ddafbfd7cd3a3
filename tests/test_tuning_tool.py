"""CPU: tools/ab_options.py picks the tuned defaults of an installation (megahit_amd/mhx_tuning.conf, read by libmhx at
mhx_create: include/mhx.h mhx_get_option) — the greedy walk over the knobs keeps a knob only when the step gets faster AND the
outputs still match the reference's digest; bench.py reports the file it ran under.  The engine is a stand-in here (the
measured runs are GPU work: tests/test_gpu_tuning.py, tests/test_gpu_round3_knobs.py)."""
import os
import runpy
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeEngine:
    """a step costs 10 ms; knob `fast` saves 4 ms, `slow` costs 3, `wrong` saves 6 but breaks the output"""
    def __init__(self, device):
        self.o = {}

    def load_sequences(self, *a):
        pass

    def set_option(self, name, v):
        self.o[name] = v

    def read2sdbg_s1(self, k, m):
        time.sleep(0.001 * (10 - 4 * self.o.get("fast", 0) + 3 * self.o.get("slow", 0) - 6 * self.o.get("wrong", 0)))
        return types.SimpleNamespace(n_items=1)

    def read2sdbg_s2(self, k, m):
        return types.SimpleNamespace(n_sdbg=1)

    def synchronize(self):
        pass

    def profile(self, on):
        pass

    def profile_reset(self):
        pass

    def profile_get(self):
        return {"kernel": {"ms": 1.0, "launches": 1, "bytes": 1}}

    def close(self):
        pass


def run_tool(monkeypatch, argv):
    import bench
    from megahit_amd import lib
    monkeypatch.setattr(lib, "Engine", FakeEngine)
    monkeypatch.setattr(bench, "make_reads", lambda n, r, w: np.zeros(4, dtype=np.uint32))
    monkeypatch.setattr(bench, "output_parity", lambda eng, e, n, w, res: {"checked": eng.o.get("wrong", 0) == 0})
    monkeypatch.setattr(sys, "argv", ["ab_options.py"] + argv)
    runpy.run_path(os.path.join(ROOT, "tools", "ab_options.py"), run_name="__main__")


def read_conf(path):
    import bench
    os.environ["MHX_TUNING_FILE"] = str(path)
    try:
        return bench.tuned_defaults()
    finally:
        del os.environ["MHX_TUNING_FILE"]


def test_greedy_keeps_what_is_faster_and_still_right(tmp_path, monkeypatch, capsys):
    conf = tmp_path / "mhx_tuning.conf"
    run_tool(monkeypatch, ["base=1", "--greedy", "wrong fast slow", "--write-tuning", str(conf), "--steps", "3", "--warmup", "1", "--min-gain-ms", "1.0"])
    assert read_conf(conf) == {"base": 1, "wrong": 0, "fast": 1, "slow": 0}
    text = conf.read_text()
    assert "REJECTED (outputs differ)" in text and text.count("ms/step") == 5  # start, three trials, the result measured again


def test_list_mode_keeps_the_first_configuration_unless_another_one_is_clearly_faster(tmp_path, monkeypatch, capsys):
    conf = tmp_path / "mhx_tuning.conf"
    run_tool(monkeypatch, ["fast=0 slow=0", "fast=0 slow=1", "fast=1 slow=0", "--rounds", "2", "--write-tuning", str(conf), "--steps", "3", "--warmup", "1",
                           "--min-gain-ms", "1.0"])
    assert read_conf(conf) == {"fast": 1, "slow": 0}
    run_tool(monkeypatch, ["fast=0 slow=0", "fast=0 slow=1", "--write-tuning", str(conf), "--steps", "3", "--warmup", "1", "--min-gain-ms", "1.0"])
    assert read_conf(conf) == {"fast": 0, "slow": 0}


def test_nothing_is_written_when_no_configuration_reproduces_the_reference(tmp_path, monkeypatch, capsys):
    conf = tmp_path / "mhx_tuning.conf"
    run_tool(monkeypatch, ["wrong=1", "--write-tuning", str(conf), "--steps", "2", "--warmup", "0"])
    assert not conf.exists()


def test_the_committed_tuning_file_parses_and_names_known_knobs():
    import bench
    known = {"sort_unit_runs", "sort_rank_uniform", "s1_gen_blocked", "s1_digit_hist_preload", "s1_stream_read_first", "s1_stream_used_list",
             "s1_stream_unroll", "s1_stream_prefetch", "s1_stream_next_bucket"}
    got = bench.tuned_defaults()
    assert set(got) <= known, set(got) - known
    with open(os.path.join(ROOT, "include", "mhx.h")) as f:
        header = f.read()
    for name in got:
        assert name in header, "knob %s of mhx_tuning.conf is not described in include/mhx.h" % name
