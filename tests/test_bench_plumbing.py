"""CPU: the evidence plumbing of bench.py (no GPU work): PMC traffic lookup and the committed profile files."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_traffic_lookup_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC measurement only if that measurement was taken on the very kernel
    sources that are running (megahit_amd/buildid.py)"""
    import bench
    from megahit_amd.buildid import build_id, lib_id
    doc = {"build_id": build_id(), "lib_id": lib_id(),
           "kernels": {"k_radix_onesweep<3, 8, 3, SrcArray<3> >": {"hbm_bytes": 40000000000},
                       "k_radix_onesweep<3, 8, 3, S1Gen>": {"hbm_bytes": 25000000000},
                       "k_radix_onesweep<2, 8, 2, SrcArray<2> >": {"hbm_bytes": 2000000000},
                       "k_s1_stream<false, 4, 1024, 13>": {"hbm_bytes": 250000000},
                       "k_s1_stream<true, 4, 1024, 13>": {"hbm_bytes": 21000000000}}}
    p = str(tmp_path / "pmc.json")
    with open(p, "w") as f:
        json.dump(doc, f)
    traffic, src = bench.pmc_traffic("radix_scatter_12B", p)
    assert traffic == 40000000000 and "k_radix_onesweep<3" in src
    assert bench.pmc_traffic("radix_scatter_12B_gen", p)[0] == 25000000000
    assert bench.pmc_traffic("radix_scatter_8B", p)[0] == 2000000000
    assert bench.pmc_traffic("s1_groups", p)[0] == 21000000000
    doc["lib_id"] = "1" * 16  # another binary: accepted only if it was built from the current sources (not older than any of them)
    with open(p, "w") as f:
        json.dump(doc, f)
    monkeypatch.setattr(bench, "lib_built_from_current_sources", lambda: True)
    assert bench.pmc_traffic("radix_scatter_12B", p)[0] == 40000000000
    monkeypatch.setattr(bench, "lib_built_from_current_sources", lambda: False)
    traffic, why = bench.pmc_traffic("radix_scatter_12B", p)
    assert traffic is None and "another build of libmhx.so" in why
    doc["build_id"] = "0" * 16
    with open(p, "w") as f:
        json.dump(doc, f)
    traffic, why = bench.pmc_traffic("radix_scatter_12B", p)
    assert traffic is None and "other kernel sources" in why


def test_committed_bench_line_has_the_contract_fields():
    """the round's committed evidence is self-consistent: the bench line quotes the counter measurement of the build it ran on"""
    import glob
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench.json")))
    assert lines
    with open(lines[-1]) as f:
        d = json.loads(f.read())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity_checked", "e2e"):
        assert key in d, key
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["peak"] == 8000.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"] and d["vs_baseline"] is None
    assert d["parity_checked"] is True and d["parity"]["digest"] == d["parity"]["reference_digest"]
    # the counter measurement the line quotes: the file of the same round, taken on the build the line ran on
    tag = os.path.basename(lines[-1])[:3]
    with open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")) as f:
        pmc = json.load(f)
    if d["roofline"]["traffic"] is not None:
        assert d["roofline"]["traffic"] in [v["hbm_bytes"] for v in pmc["kernels"].values()]
        assert d.get("build_id") in (None, pmc["build_id"])


def test_bench_with_several_gpus_launches_its_own_ranks():
    """`python bench.py --gpus N` the way `--gpus 1` is called, without a launcher (no WORLD_SIZE): the script becomes the launcher —
    torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 — instead of exiting with "launch with torch.distributed.run"
    (VERDICT r4 weak #8).  No GPU here: every one of the N ranks gets as far as the check for one."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd="/tmp", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)  # (the first `import torch` of a fresh container takes minutes)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert p.returncode == 0 and p.stdout.strip().startswith("{")
        return
    assert p.returncode != 0
    assert "launch with torch.distributed.run" not in p.stderr
    started = p.stderr.count("bench.py needs a GPU") + p.stderr.count("this node shows")
    assert started == 2, p.stderr[-1500:]
