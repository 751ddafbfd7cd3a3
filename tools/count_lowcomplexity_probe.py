"""GPU probe: `count` (the stream form, count_stream = 1) on the bench library with 1 % of its reads replaced by poly-A / poly-G — giant
buckets are NOT cut into slices in `count`: a workgroup streams such a bucket alone and looks at it a second time when it
holds a solid key without an in- or out-edge.  Run under a short `timeout`: the tile path (count_stream = 0) on such a library takes
minutes (one wavefront walks the tail of a 10^7-record group) and is not measured here.

    timeout 150 python tools/count_lowcomplexity_probe.py > profiles/r05_count_lowcomplexity.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from megahit_amd import lib  # noqa: E402


def main():
    n_reads = 10000000
    packed = bench.make_reads(n_reads, 0, 1)
    eng = lib.Engine(0)
    out = {"reads": n_reads, "runs": []}

    def measure(words, label):
        eng.load_sequences(words, n_reads, bench.READ_LEN, None)
        eng.count(bench.K, bench.MIN_COUNT)
        eng.synchronize()
        eng.profile(True)
        eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(2):
            r = eng.count(bench.K, bench.MIN_COUNT)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / 2
        st = eng.profile_get()
        eng.profile(False)
        out["runs"].append({"label": label, "ms_per_step": round(dt * 1e3, 2), "solid_edges": int(r.n_edges),
                            "kernel_ms_per_step": {k: round(v["ms"] / 2, 2) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])[:5]}})
        sys.stderr.write(json.dumps(out["runs"][-1]) + "\n")
        sys.stderr.flush()

    measure(packed, "the bench library")
    for frac, word, name in ((0.01, 0, "1 % poly-A"), (0.01, 0xAAAAAAAA, "1 % poly-G")):
        n_plant = int(n_reads * frac) // 16 * 16
        w = packed.copy()
        w[: n_plant // 16 * 150] = word
        measure(w, name)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
