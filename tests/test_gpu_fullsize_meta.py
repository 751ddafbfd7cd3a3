"""GPU, BASELINE configs[4] at single-GPU-shard size: `mhx_core read2sdbg -k 27 -m 1` (meta-large preset: stage 1 skipped,
every (k+1)-mer occurrence a stage-2 item, reference src/main_sdbg_build.cpp:139-147, src/sorting/read_to_sdbg_s2.cpp:295,381)
on 40 M synthetic metagenome reads (160 genomes, log-normal abundances) reproduces the digest of the reference's own run
(tests/golden/fullsize_meta.json, tools/make_fullsize_golden.py --preset meta).  ~10 G items of 8 bytes: two sort buffers
plus the filter copy exceed the HBM, so the memory plan (plan_ranges, the reference's lv1 passes: base_engine.cpp:54-141)
has to fire by itself — asserted from the log."""
import json
import os
import re
import subprocess
import sys

import pytest

import golden_util as gu
from megahit_amd import canon

pytestmark = pytest.mark.gpu

PATH = os.path.join(gu.GOLD, "fullsize_meta.json")
if os.path.exists(PATH):
    with open(PATH) as f:
        FULL = json.load(f)
else:
    FULL = None
sys.path.insert(0, os.path.join(gu.ROOT, "tools"))


@pytest.mark.skipif(FULL is None, reason="tests/golden/fullsize_meta.json not generated")
def test_meta_shard_read2sdbg_k27_m1(tmp_path):
    """two routes, one answer: (1) round 6's default — stage 2's solid items from a COUNT of the (k+1)-mers (s2.hip s2_agg_from_count: 4.9 G
    12-byte records, 2.2 G aggregated items; everything fits, no memory plan); (2) the per-occurrence stage 2 (MHX_S2_AGG_FROM_COUNT=0:
    10 G items, where the memory plan has to fire by itself)"""
    import make_fullsize_golden as mfg
    d = str(tmp_path)
    mfg.gen_meta_library(os.path.join(d, "reads"), FULL["reads"])
    assert canon.digest_file(os.path.join(d, "reads.bin")) == FULL["lib_bin_md5"], "the generator is not deterministic across boxes"
    want = FULL["cases"]["read2sdbg"]
    for label, env in (("count", {}), ("occurrences", {"MHX_S2_AGG_FROM_COUNT": "0"})):
        out = os.path.join(d, "r2s_" + label)
        p = subprocess.run([gu.MHX_CORE, "read2sdbg", "-k", str(FULL["k"]), "-m", str(FULL["m"]), "--host_mem", "64e9", "--num_cpu_threads", "8",
                            "--read_lib_file", os.path.join(d, "reads"), "--output_prefix", out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, MHX_PROFILE="1", **env))
        assert p.returncode == 0, p.stderr[-2000:]
        m = re.search(r"Memory plan: (\d+) passes", p.stderr)
        if label == "count":
            assert "s2_edges_to_items" in p.stderr and "s2_count" not in p.stderr, p.stderr[-2500:]
        else:
            assert m and int(m.group(1)) >= 2, "the memory plan did not fire:\n" + p.stderr[-1500:]
            assert "s2_edges_to_items" not in p.stderr
        _hdr, rows = canon.read_sdbg_info(out)
        live = [r for r in rows if r[0] != canon.NULL_ID]
        assert (sum(r[3] for r in live), sum(r[4] for r in live), sum(r[5] for r in live)) == (want["n_sdbg"], want["n_tips"], want["n_large"]), label
        assert canon.digest_sdbg(out) == want["digest"], label
        for fn in os.listdir(d):
            if fn.startswith("r2s_"):
                os.remove(os.path.join(d, fn))
