"""CPU: the front process of mhx_core.  The work runs in a forked child; the process the caller started exits with the
child's status as soon as the child reports "outputs complete" (so the release of the GPU mappings is not on the caller's
clock), or, when the child ends without that report, with the child's exit status.  Exercised here through
`buildlib` with the sequential host parser (MHX_BUILDLIB_HOST=1: no GPU involved)."""
import os
import subprocess

import pytest

import golden_util as gu
from megahit_amd import canon

pytestmark = pytest.mark.skipif(not os.path.exists(gu.MHX_CORE), reason="needs mhx_core")


def write_inputs(d, n=400):
    fa = os.path.join(d, "r.fa")
    with open(fa, "w") as f:
        for i in range(n):
            f.write(">r%d\n%s\n" % (i, "ACGT" * 10 + "ACGTTGCA"[i % 8:] + "GGA"))
    lib = os.path.join(d, "in.lib")
    with open(lib, "w") as f:
        f.write("%s\nse %s\n" % (fa, fa))
    return lib


def run(args, env):
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([gu.MHX_CORE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=60)


@pytest.mark.parametrize("fork", [True, False], ids=["front-process", "MHX_NO_FORK"])
def test_outputs_are_complete_when_the_started_process_returns(tmp_path, fork):
    d = str(tmp_path)
    lib = write_inputs(d)
    env = {"MHX_BUILDLIB_HOST": "1"}
    if not fork:
        env["MHX_NO_FORK"] = "1"
    outs = []
    for i in range(3):  # the child of run i may still be on its way out when run i+1 starts
        p = run(["buildlib", lib, os.path.join(d, "out%d" % i)], env)
        assert p.returncode == 0, p.stderr
        # complete the moment the started process is gone: sizes consistent with the header, equal across runs
        info = open(os.path.join(d, "out%d.lib_info" % i)).read().split("\n")
        total_bases, n_reads = (int(x) for x in info[0].split())
        assert n_reads == 400
        outs.append((canon.digest_file(os.path.join(d, "out%d.bin" % i)), total_bases))
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("fork", [True, False], ids=["front-process", "MHX_NO_FORK"])
def test_failure_status_and_message_reach_the_caller(tmp_path, fork):
    env = {"MHX_BUILDLIB_HOST": "1"}
    if not fork:
        env["MHX_NO_FORK"] = "1"
    p = run(["buildlib", os.path.join(str(tmp_path), "missing.lib"), os.path.join(str(tmp_path), "out")], env)
    assert p.returncode == 1
    assert p.stderr.strip() != ""


def test_both_ways_of_running_write_the_same_files(tmp_path):
    d = str(tmp_path)
    lib = write_inputs(d)
    a = run(["buildlib", lib, os.path.join(d, "a")], {"MHX_BUILDLIB_HOST": "1"})
    b = run(["buildlib", lib, os.path.join(d, "b")], {"MHX_BUILDLIB_HOST": "1", "MHX_NO_FORK": "1"})
    assert a.returncode == 0 and b.returncode == 0
    assert canon.digest_file(os.path.join(d, "a.bin")) == canon.digest_file(os.path.join(d, "b.bin"))
    assert open(os.path.join(d, "a.lib_info")).read() == open(os.path.join(d, "b.lib_info")).read()


def test_a_chained_scan_timeout_is_retried_with_the_classic_sort(tmp_path):
    """the worker reports "chained scan timed out" (here: a test hook pretends it once) -> the front process starts the
    sub-program again with MHX_SORT=classic instead of failing the caller; without a front process the error stays fatal"""
    d = str(tmp_path)
    lib = write_inputs(d)
    p = run(["buildlib", lib, os.path.join(d, "out")], {"MHX_BUILDLIB_HOST": "1", "MHX_TEST_SCAN_TIMEOUT_ONCE": "1"})
    assert p.returncode == 0, p.stderr
    assert "running the sub-program again with MHX_SORT=classic" in p.stderr
    ref = run(["buildlib", lib, os.path.join(d, "ref")], {"MHX_BUILDLIB_HOST": "1"})
    assert ref.returncode == 0
    assert canon.digest_file(os.path.join(d, "out.bin")) == canon.digest_file(os.path.join(d, "ref.bin"))
    q = run(["buildlib", lib, os.path.join(d, "out2")], {"MHX_BUILDLIB_HOST": "1", "MHX_TEST_SCAN_TIMEOUT_ONCE": "1", "MHX_NO_FORK": "1"})
    assert q.returncode == 1 and "chained scan timed out" in q.stderr
    # a caller who chose the sort himself gets no retry loop (the hook only fires without MHX_SORT)
    r = run(["buildlib", lib, os.path.join(d, "out3")], {"MHX_BUILDLIB_HOST": "1", "MHX_TEST_SCAN_TIMEOUT_ONCE": "1", "MHX_SORT": "classic"})
    assert r.returncode == 0
