"""Per-kernel times of read2sdbg stage 1 with the reference-exact tie order (want_mercy=2) on the BASELINE configs[1]
library (bench.py's reads).  python tools/mercy_prof.py [reads]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from megahit_amd import lib  # noqa: E402


def main():
    n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
    n_reads = n_reads // 16 * 16
    packed = bench.make_reads(n_reads, 0, 1)
    eng = lib.Engine(0)
    eng.load_sequences(packed, n_reads, bench.READ_LEN, None)
    for key in sys.argv[2:]:
        name, v = key.split("=")
        eng.set_option(name, int(v))
    t0 = time.perf_counter()
    eng.read2sdbg_s1(bench.K, bench.MIN_COUNT, want_mercy=2)
    eng.synchronize()
    first = time.perf_counter() - t0  # includes the first-use allocations of every workspace
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    r1 = eng.read2sdbg_s1(bench.K, bench.MIN_COUNT, want_mercy=2)
    eng.synchronize()
    dt = time.perf_counter() - t0
    stats = eng.profile_get()
    print(json.dumps({"reads": n_reads, "stage1_s": round(dt, 4), "stage1_first_call_s": round(first, 4), "n_items": r1.n_items, "n_mercy_cand": r1.n_mercy_cand,
                      "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])},
                      "launches": {k: v["launches"] for k, v in stats.items()}}, indent=1))


if __name__ == "__main__":
    main()
