"""GPU probe: what a GPU process pays when it starts right behind another one (device init, hipMalloc, hipFree), with the
two ways mhx_core can leave (default: device memory released before it returns; MHX_EARLY_EXIT=1: the front process
returns while the worker still dies).  Sequences of mhx_core processes on the 10 M-read library, no pauses.
    python tools/churn_probe.py > profiles/r03_process_churn.json"""
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

MHX = os.path.join(ROOT, "megahit_amd", "mhx_core")


def call(args, env):
    e = dict(os.environ)
    e.update(env)
    t0 = time.perf_counter()
    p = subprocess.run([MHX] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=e)
    dt = time.perf_counter() - t0
    assert p.returncode == 0, p.stderr[-1500:]
    ph = {m.group(1).strip()[:28]: float(m.group(2)) for m in re.finditer(r"INFO\s+(.*?)\.? Time elapsed: ([0-9.]+)", p.stderr)}
    m = re.search(r"Device memory: (\d+) allocations, ([0-9.]+) GB, ([0-9.]+) s in hipMalloc, ([0-9.]+) s in hipFree", p.stderr)
    mem = {"allocs": int(m.group(1)), "GB": float(m.group(2)), "hipMalloc_s": float(m.group(3)), "hipFree_s": float(m.group(4))} if m else None
    return {"wall_s": round(dt, 3), "device_ready_s": ph.get("Device ready"), "mem": mem, "phases": ph}


def main():
    out = {}
    with tempfile.TemporaryDirectory(prefix="mhx_churn_") as d:
        mfg.gen_library(os.path.join(d, "reads"), 10000000)
        common = ["-k", "21", "-m", "2", "--host_mem", "64e9", "--num_cpu_threads", "8", "--read_lib_file", os.path.join(d, "reads")]
        r2s = ["read2sdbg"] + common + ["--output_prefix", os.path.join(d, "a")]
        cnt = ["count"] + common + ["--output_prefix", os.path.join(d, "c")]
        s2s = ["seq2sdbg", "-k", "21", "--kmer_from", "0", "--host_mem", "64e9", "--num_cpu_threads", "8", "--input_prefix", os.path.join(d, "c"),
               "--need_mercy", "--output_prefix", os.path.join(d, "e")]
        for label, env in (("release_before_return", {}), ("early_exit", {"MHX_EARLY_EXIT": "1"}), ("no_fork_clean", {"MHX_NO_FORK": "1"})):
            time.sleep(4.0)  # let the previous sequence's memory come back: each sequence starts on a quiet device
            seq = []
            t0 = time.perf_counter()
            for name, args in (("read2sdbg", r2s), ("read2sdbg", r2s), ("read2sdbg", r2s), ("count", cnt), ("seq2sdbg_need_mercy", s2s), ("count", cnt), ("seq2sdbg_need_mercy", s2s)):
                r = call(args, env)
                r["sub_program"] = name
                seq.append(r)
            out[label] = {"total_s": round(time.perf_counter() - t0, 3), "processes": seq}
            sys.stderr.write("%s %.2f s: %s\n" % (label, out[label]["total_s"], " ".join("%s=%.2f(m%.2f,f%.2f,d%.2f)" % (
                r["sub_program"][:5], r["wall_s"], r["mem"]["hipMalloc_s"] if r["mem"] else -1, r["mem"]["hipFree_s"] if r["mem"] else -1, r["device_ready_s"] or -1) for r in seq)))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
