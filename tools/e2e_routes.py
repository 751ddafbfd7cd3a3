"""Files in -> files out for the three ways MEGAHIT builds its first graph, on the BASELINE configs[1] library
(10 M synthetic 150 bp PE reads, k=21, m=2), through mhx_core; digests against tests/golden/fullsize.json.
   1-pass            read2sdbg                      (--kmin-1pass without mercy)
   1-pass + mercy    read2sdbg --need_mercy         (reference-exact tie order)
   2-pass            count, then seq2sdbg --need_mercy   (the orchestrator's default, src/megahit:939-966)
   python tools/e2e_routes.py > profiles/r03_e2e_routes.json"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_fullsize_golden as mfg  # noqa: E402
from megahit_amd import canon  # noqa: E402

MHX = os.path.join(ROOT, "megahit_amd", "mhx_core")


def run(args):
    best = None
    for _ in range(2):  # back to back, no pauses: mhx_core gives its device memory back before it returns
        t0 = time.perf_counter()
        p = subprocess.run([MHX] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr[-2000:]
        if best is None or dt < best[0]:
            best = (dt, p.stderr)
    phases = {m.group(1).strip()[:44]: float(m.group(2)) for m in re.finditer(r"INFO\s+(.*?)\.? Time elapsed: ([0-9.]+)", best[1])}
    return round(best[0], 3), phases


def main():
    with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as f:
        full = json.load(f)
    n, k, m = full["reads"], full["k"], full["m"]
    out = {"reads": n, "k": k, "m": m, "routes": {}}
    with tempfile.TemporaryDirectory(prefix="mhx_routes_") as d:
        mfg.gen_library(os.path.join(d, "reads"), n)
        common = ["-k", str(k), "-m", str(m), "--host_mem", "64e9", "--num_cpu_threads", "8", "--read_lib_file", os.path.join(d, "reads")]
        w, ph = run(["read2sdbg"] + common + ["--output_prefix", os.path.join(d, "a")])
        out["routes"]["read2sdbg"] = {"wall_s": w, "phases_s": ph, "bit_identical": canon.digest_sdbg(os.path.join(d, "a")) == full["cases"]["read2sdbg"]["digest"]}
        w, ph = run(["read2sdbg"] + common + ["--need_mercy", "--output_prefix", os.path.join(d, "b")])
        out["routes"]["read2sdbg_need_mercy"] = {"wall_s": w, "phases_s": ph,
                                                 "bit_identical": canon.digest_sdbg(os.path.join(d, "b")) == full["cases"]["read2sdbg_need_mercy"]["digest"]}
        w1, ph1 = run(["count"] + common + ["--output_prefix", os.path.join(d, "c")])
        w2, ph2 = run(["seq2sdbg", "-k", str(k), "--kmer_from", "0", "--host_mem", "64e9", "--num_cpu_threads", "8", "--input_prefix", os.path.join(d, "c"),
                       "--need_mercy", "--output_prefix", os.path.join(d, "e")])
        out["routes"]["count_then_seq2sdbg_need_mercy"] = {
            "wall_s": round(w1 + w2, 3), "count_wall_s": w1, "seq2sdbg_wall_s": w2, "count_phases_s": ph1, "seq2sdbg_phases_s": ph2,
            "bit_identical": canon.digest_edges(os.path.join(d, "c")) == full["cases"]["count"]["digest"] and
            canon.digest_sdbg(os.path.join(d, "e")) == full["cases"]["seq2sdbg_need_mercy"]["digest"]}
    for name, ref in (("read2sdbg", "read2sdbg"), ("read2sdbg_need_mercy", "read2sdbg_need_mercy")):
        out["routes"][name]["reference_wall_s_8_threads_build_container"] = full["cases"][ref]["wall_s"]
    out["routes"]["count_then_seq2sdbg_need_mercy"]["reference_wall_s_8_threads_build_container"] = round(
        full["cases"]["count"]["wall_s"] + full["cases"]["seq2sdbg_need_mercy"]["wall_s"], 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
