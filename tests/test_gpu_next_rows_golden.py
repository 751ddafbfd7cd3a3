"""GPU: the SURVEY.md section 8f rows against COMMITTED known answers (tests/golden/next_rows.json, produced from the
reference's own code by tools/make_next_rows_golden.py) — nothing of the reference is needed at run time.
   N4  mhx_sdbg_remove_tips on the graph mhx_core builds for a golden case: tips removed, invalid bit vector before / after
   N2  mhx_core iterate on tests/golden/iterate_*: header and sorted edge records"""
import hashlib
import json
import os

import numpy as np
import pytest

import golden_util as gu
from megahit_amd import canon, lib

pytestmark = pytest.mark.gpu

with open(os.path.join(gu.GOLD, "next_rows.json")) as f:
    GOLDEN = json.load(f)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_files_into(engine, prefix):
    hdr, buckets = canon.canonical_sdbg(prefix)
    off = np.zeros(65536, dtype=np.uint64)
    items, tips, large = off.copy(), off.copy(), off.copy()
    parts, pos = [], 0
    for bid, ni, nt, nl, b in buckets:
        off[bid], items[bid], tips[bid], large[bid] = pos, ni, nt, nl
        parts.append(b)
        pos += len(b)
    data = np.frombuffer(b"".join(parts), dtype=np.uint8) if parts else np.zeros(0, dtype=np.uint8)
    engine.sdbg_load_bytes(data, off, items, tips, large)
    return hdr["k"]


@pytest.mark.parametrize("g", GOLDEN["tips"], ids=lambda g: gu.case_id({"case": g["case"]}))
def test_tip_trimming_matches_the_committed_answer(engine, g, tmp_path):
    gu.run_case(gu.MHX_CORE, {"case": g["case"]}, str(tmp_path))
    k = load_files_into(engine, os.path.join(str(tmp_path), "out"))
    info = engine.sdbg_build_index(k)
    assert info.n_items == g["n_items"]
    assert sha(engine.fetch(lib.BUF_SDBG_INVALID, np.uint64)) == g["invalid_before"]
    assert engine.sdbg_remove_tips(info, g["max_tip_len"]) == g["tips_removed"]
    assert sha(engine.fetch(lib.BUF_SDBG_INVALID, np.uint64)) == g["invalid_after"]


@pytest.mark.parametrize("g", GOLDEN["iterate"], ids=lambda g: g["dir"])
def test_iterate_matches_the_committed_answer(g, tmp_path):
    import subprocess
    d = os.path.join(gu.GOLD, g["dir"])
    out = os.path.join(str(tmp_path), "it")
    p = subprocess.run([gu.MHX_CORE, "iterate", "-c", os.path.join(d, "c.fa"), "-b", os.path.join(d, "b.fa"), "-t", "2", "-k", str(g["k"]),
                        "-s", str(g["step"]), "-o", out, "-r", os.path.join(d, "reads.bin")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-1500:]
    hdr, edges, _ = canon.canonical_edges(out)
    assert hdr == g["header"]
    edges = np.ascontiguousarray(edges)
    edges = edges[np.lexsort(edges.T[::-1])]
    assert edges.shape[0] == g["n_edges"] and sha(edges) == g["sorted_edges"]
