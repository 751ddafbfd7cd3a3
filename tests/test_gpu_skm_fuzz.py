"""GPU: twenty seconds of tools/fuzz_skm.py — random libraries (ragged lengths, planted poly-X and short-period repeats, duplicated reads),
k = 19..22, min count 1..2 and random knobs of the super-k-mer path (bins, passes, table fill, probe limits, tags, dealing, homopolymer
side path, give-up limits) through read2sdbg stage 1 + 2 and count, every output against the oracle (profiles/r06_fuzz_skm.txt: 1 839
rounds in five minutes on the final build of round 6)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_twenty_seconds_of_random_libraries():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_skm.py"), "20", "77000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "all equal to the oracle" in p.stdout, p.stdout[-1500:]
