"""buildlib on a FASTA pair of N synthetic 150 bp reads: the reference (single-threaded parse + pack) vs mhx_core
(text in HBM, parsed and packed by kernels).   python tools/buildlib_bench.py [--reads 4e6]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from megahit_amd import canon, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=float, default=4e6)
    args = ap.parse_args()
    n = int(args.reads) // 2 * 2
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = {"reads": n}
    with tempfile.TemporaryDirectory(prefix="mhx_bl_") as d:
        G = int(n * 2.5)
        reads = synth.gen_pe_reads(n // 2, G, read_len=150, frag=400, err=0.005, seed=3)
        for mate in (0, 1):
            rows = lut[reads[mate::2]]
            hdr = np.array([(">r%d/%d\n" % (i, mate + 1)).encode() for i in range(rows.shape[0])], dtype=object)
            with open(os.path.join(d, "r%d.fa" % (mate + 1)), "wb") as f:
                for h, row in zip(hdr, rows):
                    f.write(h)
                    f.write(row.tobytes())
                    f.write(b"\n")
        out["text_MB"] = round(sum(os.path.getsize(os.path.join(d, "r%d.fa" % m)) for m in (1, 2)) / 1e6, 1)
        with open(os.path.join(d, "lib"), "w") as f:
            f.write("bench\npe %s %s\n" % (os.path.join(d, "r1.fa"), os.path.join(d, "r2.fa")))
        for tag, exe in (("reference", os.path.join(ROOT, "oracle", "_ref", "ref_core")), ("mhx_core", os.path.join(ROOT, "megahit_amd", "mhx_core"))):
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                subprocess.run([exe, "buildlib", os.path.join(d, "lib"), os.path.join(d, tag)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[tag + "_s"] = round(best, 3)
            out[tag + "_bin_md5"] = canon.digest_file(os.path.join(d, tag + ".bin"))
        # where mhx_core's time goes: its own phase clocks and per-kernel HIP-event times (MHX_PROFILE)
        import re
        prof = os.path.join(d, "prof.json")
        env = dict(os.environ, MHX_PROFILE="1", MHX_PROFILE_JSON=prof)
        p = subprocess.run([os.path.join(ROOT, "megahit_amd", "mhx_core"), "buildlib", os.path.join(d, "lib"), os.path.join(d, "p")], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
        with open(prof) as f:
            ks = json.load(f)["kernels"]
        out["kernel_ms_total"] = round(sum(v["ms"] for v in ks.values()), 3)
        out["kernels"] = {k: {"ms": round(v["ms"], 3), "algo_bytes": v["bytes"], "GBs": round(v["bytes"] / v["ms"] / 1e6, 1) if v["ms"] else None}
                          for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"])[:8]}
        m = re.search(r"Device memory: .*", p.stderr)
        out["device_memory"] = m.group(0) if m else None
        out["host_share_s"] = round(out["mhx_core_s"] - out["kernel_ms_total"] / 1e3, 3)
        out["identical"] = out["reference_bin_md5"] == out["mhx_core_bin_md5"]
        out["speedup"] = round(out["reference_s"] / out["mhx_core_s"], 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
