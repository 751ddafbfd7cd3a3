"""CPU: `mhx_core iterate` rejects the option combinations the reference's iterate rejects (src/main_iterate.cpp:71-93) with
the same message and exit status, before any device is opened."""
import os
import subprocess

import pytest

import consume_util as cu
import golden_util as gu

pytestmark = pytest.mark.skipif(not os.path.exists(gu.MHX_CORE), reason="needs mhx_core")

BASE = {"-c": "c.fa", "-b": "b.fa", "-r": "r.bin", "-k": "21", "-s": "8", "-o": "out", "-t": "2"}


def first_line(exe, opts):
    args = [exe, "iterate"]
    for key, v in opts.items():
        if v is not None:
            args += [key, v]
    p = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return p.returncode, p.stderr.splitlines()[0] if p.stderr else ""


CASES = [
    ("no contig file", {"-c": None}, "No contig file!"),
    ("no bubble file", {"-b": None}, "No bubble file!"),
    ("no reads", {"-r": None}, "No reads file!"),
    ("no output", {"-o": None}, "No output prefix!"),
    ("k = 0", {"-k": "0"}, "Invalid kmer size!"),
    ("odd step", {"-s": "7"}, "Invalid step size!"),
    ("step > 28", {"-s": "30"}, "Invalid step size!"),
    ("step = 0", {"-s": "0"}, "Invalid step size!"),
    ("k + step too large", {"-k": "241", "-s": "20"}, "kmer_k + step must less than 256"),
]


@pytest.mark.parametrize("name,change,message", CASES, ids=[c[0] for c in CASES])
def test_iterate_argument_errors(name, change, message):
    opts = dict(BASE)
    opts.update(change)
    rc, line = first_line(gu.MHX_CORE, opts)
    assert rc == 1 and line == message
    if os.path.exists(cu.REF_FULL):
        assert first_line(cu.REF_FULL, opts) == (rc, line)
