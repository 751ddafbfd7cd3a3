"""The reference's CPU path at FULL size (BASELINE configs[1]: 10 M reads, read2sdbg k=21 m=2) on this host, at several
OpenMP thread counts -> JSON (profiles/r03_cpu_fullsize.json).  Run once per round on the GPU box's host by
tools/gpu_evidence.sh; bench.py quotes it beside its bounded in-run sample.

    python tools/cpu_fullsize.py [--threads 8,32] > profiles/r03_cpu_fullsize.json"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_fullsize_golden as mfg  # noqa: E402
from megahit_amd import canon  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="8,32")
    ap.add_argument("--reads", type=float, default=1e7)
    args = ap.parse_args()
    n = int(args.reads) // 2 * 2
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_core")
    E = n * (150 - 21)
    out = {"workload": "read2sdbg k=21 m=2, %d synthetic 150 bp PE reads (the bench.py library)" % n, "edges": E, "host_cores": os.cpu_count(),
           "binary": "oracle/_ref/ref_core (reference sources, -O3 -fopenmp)", "runs": {}}
    with tempfile.TemporaryDirectory(prefix="mhx_cpufull_") as d:
        mfg.gen_library(os.path.join(d, "reads"), n)
        for t in [int(x) for x in args.threads.split(",")]:
            cmd = [ref, "read2sdbg", "-k", "21", "-m", "2", "--host_mem", "64e9", "--num_cpu_threads", str(t), "--read_lib_file",
                   os.path.join(d, "reads"), "--output_prefix", os.path.join(d, "out%d" % t)]
            t0 = time.perf_counter()
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            out["runs"][str(t)] = {"wall_s": round(dt, 1), "M_edges_per_s": round(E / dt / 1e6, 2),
                                   "digest": canon.digest_sdbg(os.path.join(d, "out%d" % t))}
    best = max(out["runs"], key=lambda k: out["runs"][k]["M_edges_per_s"])
    out["best_threads"] = int(best)
    out["best_M_edges_per_s"] = out["runs"][best]["M_edges_per_s"]
    out["best_wall_s"] = out["runs"][best]["wall_s"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
