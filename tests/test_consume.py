"""CPU: the staged reference harness itself reproduces the reference's integration known answer (SURVEY.md §8c:
`./megahit --test -t 4` -> "2 contigs, total 1788 bp ..."), so that tests/test_gpu_consume.py compares against a
checked baseline.  Skipped where oracle/_ref/harness was not staged (needs /root/reference at build time)."""
import os

import pytest

import consume_util as cu

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(cu.HARNESS, "bin", "megahit")) or not os.path.exists(cu.REF_FULL),
                                reason="oracle/_ref/harness not staged (make -C oracle ref)")


@pytest.mark.parametrize("extra", [(), ("--kmin-1pass",)], ids=["2pass", "kmin-1pass"])
def test_reference_pipeline_known_answer(tmp_path, extra):
    summary, contigs = cu.run_orchestrator("bin", str(tmp_path / "out"), extra)
    assert summary.startswith("2 contigs, total 1788 bp, min 559 bp, max 1229 bp")
    assert [c[0] for c in contigs[0]] == [559, 1229] and len(contigs[1]) > 1500
