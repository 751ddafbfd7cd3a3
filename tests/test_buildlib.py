"""CPU: buildlib host logic.  `mhx_core buildlib` with MHX_BUILDLIB_HOST=1 (the sequential kseq-compatible parser, no GPU
involved) writes the same .bin / .lib_info as the reference's buildlib (oracle/_ref/ref_core = reference sources:
src/sequence/io/sequence_lib.cpp:8-91, fastx_reader.cpp, kseq.h) on every text shape kseq accepts."""
import os
import subprocess

import pytest

import buildlib_util as bu
import golden_util as gu
from megahit_amd import canon

pytestmark = pytest.mark.skipif(not os.path.exists(gu.REF_CORE) or not os.path.exists(gu.MHX_CORE), reason="needs oracle/_ref/ref_core and mhx_core")


def run_both(lib_text, d, name, env):
    libf = os.path.join(d, name + ".lib")
    with open(libf, "w") as f:
        f.write(lib_text)
    out = {}
    for tag, exe, e in (("ref", gu.REF_CORE, None), ("mhx", gu.MHX_CORE, env)):
        prefix = os.path.join(d, name + "_" + tag)
        ee = dict(os.environ)
        ee.update(e or {})
        p = subprocess.run([exe, "buildlib", libf, prefix], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=ee)
        assert p.returncode == 0, p.stderr[-1500:]
        out[tag] = (canon.digest_file(prefix + ".bin"), open(prefix + ".lib_info").read())
    return out


def test_sequential_parser_matches_reference_buildlib(tmp_path):
    cases = bu.make_cases(str(tmp_path))
    for name, (lib_text, _seq_only) in cases.items():
        got = run_both(lib_text, str(tmp_path), name, {"MHX_BUILDLIB_HOST": "1"})
        assert got["mhx"] == got["ref"], name
