"""End-to-end (files in -> files out) run of the drop-in CLI on the BASELINE configs[1] input, next to the
reference's own binary on the same files; compares the canonical streams of the outputs at full size.

    python tools/e2e_cli.py [--reads 1e7] [--skip-ref] [--threads N]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megahit_amd import canon, synth  # noqa: E402


def run(cmd):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-2000:])
        raise SystemExit("command failed: " + " ".join(cmd))
    return dt, p.stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=float, default=1e7)
    ap.add_argument("--skip-ref", action="store_true")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--prog", default="read2sdbg", choices=["read2sdbg", "count"])
    args = ap.parse_args()
    n = int(args.reads) // 2 * 2
    k, m = 21, 2
    mhx = os.path.join(ROOT, "megahit_amd", "mhx_core")
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_core")
    out = {"reads": n, "k": k, "m": m, "prog": args.prog, "edges": n * (150 - k)}
    with tempfile.TemporaryDirectory(prefix="mhx_e2e_") as d:
        t0 = time.time()
        G = int(n * 2.5)
        import numpy as np
        genome = np.random.default_rng(1).integers(0, 4, size=G, dtype=np.uint8)
        blocks = []
        for i, lo in enumerate(range(0, n // 2, 1000000)):
            c = min(1000000, n // 2 - lo)
            blocks.append(synth.gen_pe_reads(c, G, read_len=150, frag=400, err=0.005, seed=1001 + i, genome=genome))
        synth.write_read_lib(os.path.join(d, "reads"), blocks)
        out["gen_s"] = round(time.time() - t0, 1)
        common = ["-k", str(k), "-m", str(m), "--host_mem", "64e9", "--read_lib_file", os.path.join(d, "reads")]
        dt, log = run([mhx, args.prog] + common + ["--num_cpu_threads", "8", "--output_prefix", os.path.join(d, "gpu")])
        out["mhx_core_wall_s"] = round(dt, 2)
        out["mhx_core_M_edges_per_s"] = round(out["edges"] / dt / 1e6, 1)
        out["mhx_core_log"] = [l for l in log.splitlines() if "Time elapsed" in l]
        dig = canon.digest_sdbg if args.prog == "read2sdbg" else canon.digest_edges
        out["mhx_digest"] = dig(os.path.join(d, "gpu"))
        if not args.skip_ref and os.path.exists(ref):
            dt, log = run([ref, args.prog] + common + ["--num_cpu_threads", str(args.threads), "--output_prefix", os.path.join(d, "cpu")])
            out["ref_core_wall_s"] = round(dt, 2)
            out["ref_core_threads"] = args.threads
            out["ref_core_M_edges_per_s"] = round(out["edges"] / dt / 1e6, 2)
            out["ref_digest"] = dig(os.path.join(d, "cpu"))
            out["bit_identical_canonical_stream"] = out["ref_digest"] == out["mhx_digest"]
            out["speedup_end_to_end"] = round(out["ref_core_wall_s"] / out["mhx_core_wall_s"], 1)
            if args.prog == "read2sdbg":
                out["counting_equal"] = canon.digest_file(os.path.join(d, "gpu.counting")) == canon.digest_file(os.path.join(d, "cpu.counting"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
