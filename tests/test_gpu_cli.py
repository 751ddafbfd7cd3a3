"""GPU, file level (boundary B1): the drop-in CLI mhx_core reproduces the reference's known answers on
the golden inputs — bucket-ordered canonical streams of .edges/.sdbg, byte-equal .cand/.counting."""
import os

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ent", [e for e in gu.cases() if e["case"]["prog"] == "read2sdbg" and e["case"].get("mercy")][:3], ids=gu.case_id)
def test_cli_stable_tie_mode_matches_oracle_stable(ent, tmp_path, monkeypatch):
    """MHX_STABLE_TIES=1 (the fast mode): equality with the oracle in stable-tie mode."""
    monkeypatch.setenv("MHX_STABLE_TIES", "1")
    got = gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    gu.ensure_oracle()
    os.makedirs(str(tmp_path / "o"), exist_ok=True)
    want = gu.run_case(gu.ORACLE_CORE, ent, str(tmp_path / "o"), extra=["--tie", "stable"])
    assert got["sdbg"] == want["sdbg"] and got["n_sdbg"] == want["n_sdbg"]


@pytest.mark.parametrize("ent", gu.cases(), ids=gu.case_id)
def test_cli_reproduces_reference(ent, tmp_path):
    assert os.path.exists(gu.MHX_CORE), "mhx_core not built"
    c = ent["case"]
    got = gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    # read2sdbg --need_mercy included: the CLI replays kmsort's tie order (H1), so it matches the reference too
    for key, want in ent.items():
        if key in ("case", "mercy_cand_kmsort"):
            continue
        assert got.get(key) == want, key


@pytest.mark.parametrize("ent", gu.cases(), ids=gu.case_id)
def test_cli_memory_bounded_passes(ent, tmp_path, monkeypatch):
    """MHX_MAX_ITEMS caps the items per pass: every stage then runs over several lv1 bucket ranges (the reference's
    lv1 passes) and must write the same files."""
    monkeypatch.setenv("MHX_MAX_ITEMS", "20000")
    got = gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    for key, want in ent.items():
        if key in ("case", "mercy_cand_kmsort"):
            continue
        assert got.get(key) == want, key


def _pick_multi_file_cases():
    seen, out = set(), []
    for e in gu.cases():
        c = e["case"]
        key = (c["prog"], c.get("input"))
        if key not in seen:
            seen.add(key)
            out.append(e)
    return out


@pytest.mark.parametrize("ent", _pick_multi_file_cases(), ids=gu.case_id)
def test_cli_multi_file_outputs(ent, tmp_path, monkeypatch):
    """MHX_NUM_OUT_FILES=3: .edges.<i> / .sdbg.<i> split at bucket boundaries as the reference does per thread
    (--num_cpu_threads 3 in these runs); the canonical streams stay the same, and seq2sdbg reads the split edges back."""
    monkeypatch.setenv("MHX_NUM_OUT_FILES", "3")
    got = gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    for key, want in ent.items():
        if key in ("case", "mercy_cand_kmsort"):
            continue
        assert got.get(key) == want, key
    produced = [f for f in os.listdir(str(tmp_path)) if ".sdbg." in f or ".edges." in f]
    assert any(f.endswith(".2") for f in produced), produced


@pytest.mark.parametrize("ent", [e for e in gu.cases() if e["case"]["prog"] in ("count", "read2sdbg")][:4], ids=gu.case_id)
def test_cli_automatic_memory_plan(ent, tmp_path, monkeypatch):
    """the planner itself (plan_ranges: free HBM minus the stage's fixed state, / 3 item buffers) decides on several lv1
    passes when the free memory is small — here a faked 6 MB (MHX_FREE_BYTES) — and the outputs stay the reference's"""
    import subprocess
    with open(os.path.join(gu.GOLD, ent["case"]["lib"] + ".lib_info")) as f:
        total_bases = int(f.read().split()[0])
    # room for ~0.7 (read2sdbg) / ~0.45 (count: 0.79 items per base on these libraries) 12-byte items per base in two sort buffers: two or three
    # passes.  (Round 6: count runs on the 12-byte records of the stage-1 design also in passes — 24.5 bytes per kept item instead of 49.)
    monkeypatch.setenv("MHX_FREE_BYTES", str((14 if ent["case"]["prog"] == "count" else 25) * total_bases))
    got = gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    for key, want in ent.items():
        if key in ("case", "mercy_cand_kmsort"):
            continue
        assert got.get(key) == want, key
    c = ent["case"]
    p = subprocess.run([gu.MHX_CORE, c["prog"], "-k", str(c["k"]), "-m", str(c["m"]), "--read_lib_file", os.path.join(gu.GOLD, c["lib"]),
                        "--output_prefix", str(tmp_path / "again"), "--host_mem", "2e9", "--num_cpu_threads", "3"], stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "Memory plan:" in p.stderr, p.stderr[-800:]


@pytest.mark.parametrize("ent", [e for e in gu.cases() if e["case"]["k"] == 21 and e["case"].get("lib") == "hc" and not e["case"].get("input")][:3], ids=gu.case_id)
def test_cli_retries_with_the_classic_sort_after_a_scan_timeout(ent, tmp_path, monkeypatch):
    """the front process restarts a sub-program whose chained-scan sort gave up (here: a test hook pretends it in the first
    worker) with MHX_SORT=classic: exit 0 and the reference's outputs, from the histogram + scan + scatter passes"""
    monkeypatch.setenv("MHX_TEST_SCAN_TIMEOUT_ONCE", "1")
    got = gu.run_case(gu.MHX_CORE, ent, str(tmp_path))
    for key, want in ent.items():
        if key in ("case", "mercy_cand_kmsort"):
            continue
        assert got.get(key) == want, key


@pytest.mark.parametrize("prog", ["count", "read2sdbg"])
def test_memory_plan_by_time(prog, tmp_path):
    """round 6: where hipMalloc is slow (it costs per byte mapped on this driver) the memory plan takes several bucket-range passes of a
    small working set instead of one large one — forced here with MHX_ALLOC_S_PER_GB; the outputs stay the reference's
    (base_engine.cpp:54-141 plans its lv1 passes by space only)"""
    import subprocess
    ent = [e for e in gu.cases() if e["case"]["prog"] == prog and e["case"]["k"] == 21 and not e["case"].get("mercy")][0]
    c = ent["case"]
    out = str(tmp_path / "out")
    env = dict(os.environ, MHX_ALLOC_S_PER_GB="1e6")
    p = subprocess.run([gu.MHX_CORE, prog, "-k", str(c["k"]), "-m", str(c["m"]), "--read_lib_file", os.path.join(gu.GOLD, c["lib"]), "--output_prefix", out,
                        "--host_mem", "2e9", "--num_cpu_threads", "3"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    if "Memory plan by time" in p.stderr:  # (the lean form of the stage applies to this library: the plan fired)
        assert "Memory plan:" in p.stderr and "passes over lv1 bucket ranges" in p.stderr, p.stderr[-1500:]
    from megahit_amd import canon
    if prog == "count":
        assert canon.digest_edges(out) == ent["edges"] and canon.digest_file(out + ".cand") == ent["cand"] and canon.digest_file(out + ".counting") == ent["counting"]
    else:
        assert canon.digest_sdbg(out) == ent["sdbg"]


SKM_CASES = [e for e in gu.cases() if e["case"]["prog"] in ("count", "read2sdbg") and e["case"]["k"] == 21 and e["case"]["m"] == 2 and not e["case"].get("mercy")]


@pytest.mark.parametrize("mode", ["in path", "given up"])
@pytest.mark.parametrize("ent", SKM_CASES, ids=gu.case_id)
def test_cli_leaves_the_plan_to_the_super_kmer_path(ent, mode, tmp_path, monkeypatch):
    """round 6: `mhx_core read2sdbg` / `count` ask mhx_s1_self_planned / mhx_count_self_planned before they plan lv1 bucket ranges, run the
    stage once without a filter with s1_skm / count_skm = 3, and — "given up": every bin of more than a few records counts as giant —
    plan lv1 ranges after the call failed.  Same files as the reference's either way (the golden libraries: r3 fixed length, hc with
    low-complexity reads, rag with reads of every length)."""
    import subprocess
    from megahit_amd import canon
    c = ent["case"]
    for name, v in (("MHX_S1_SKM", "2"), ("MHX_S1_VAR_MIN_FILL", "5"), ("MHX_S1_SKM_CAP_PCT", "400"), ("MHX_S1_SKM_MAX_BIN", str(1 << 30) if mode == "in path" else "1")):
        monkeypatch.setenv(name, v)
    out = str(tmp_path / "out")
    args = [c["prog"], "-k", str(c["k"]), "-m", str(c["m"]), "--read_lib_file", os.path.join(gu.GOLD, c["lib"]), "--output_prefix", out, "--host_mem", "2e9",
            "--num_cpu_threads", "3"]
    p = subprocess.run([gu.MHX_CORE] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    if mode == "in path":
        assert "super-k-mers" in p.stderr and "planning lv1 bucket ranges instead" not in p.stderr, p.stderr[-1500:]
    else:
        assert "planning lv1 bucket ranges instead" in p.stderr, p.stderr[-1500:]
    got = {}
    if c["prog"] == "count":
        got = {"edges": canon.digest_edges(out), "cand": canon.digest_file(out + ".cand"), "counting": canon.digest_file(out + ".counting"),
               "n_edges": int(canon.canonical_edges(out)[1].shape[0])}
    else:
        got = {"sdbg": canon.digest_sdbg(out), "n_sdbg": int(sum(b[1] for b in canon.canonical_sdbg(out)[1])), "counting": canon.digest_file(out + ".counting")}
    for key, want in ent.items():
        if key in ("case", "mercy_cand_kmsort"):
            continue
        assert got.get(key) == want, key
