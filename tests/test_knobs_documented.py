"""Every knob the sources read (mhx_ctx::opt("name", default): mhx_set_option / MHX_<NAME> / mhx_tuning.conf) is described in
include/mhx.h or INTEGRATION.md — a caller of the C ABI meets no switch that only the source knows."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_knob_is_documented():
    names = set()
    for f in glob.glob(os.path.join(ROOT, "megahit_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "megahit_amd", "csrc", "*.h")) + \
            glob.glob(os.path.join(ROOT, "megahit_amd", "csrc", "host", "*.cpp")):
        with open(f) as fh:
            names.update(re.findall(r'opt\("([a-z0-9_]+)"', fh.read()))
    with open(os.path.join(ROOT, "include", "mhx.h")) as fh:
        doc = fh.read()
    with open(os.path.join(ROOT, "INTEGRATION.md")) as fh:
        doc += fh.read()
    assert len(names) > 50
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
