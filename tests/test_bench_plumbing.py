"""CPU: the evidence plumbing of bench.py (no GPU work): PMC traffic lookup and the committed profile files."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_traffic_lookup_matches_committed_profile():
    import bench
    traffic, src = bench.pmc_traffic("radix_scatter_12B")
    assert traffic and traffic > 3.0e10 and "k_radix_onesweep<3" in src
    t8, _ = bench.pmc_traffic("radix_scatter_8B")
    assert t8 and t8 < traffic
    assert bench.pmc_traffic("s1_groups") == (None, None)


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r01_bench_v5.json")) as f:
        d = json.loads(f.read())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["peak"] == 8000.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"] and d["vs_baseline"] is None
