"""GPU probe: what low-complexity reads cost the stage-1 fast path.  The 10 M-read library of bench.py, then the same library
with a fraction of its reads replaced by (a) poly-A reads — one lv1 bucket (AAAAAAAA) then holds ~133 records of ONE key per
planted read, and (b) a dinucleotide repeat (ACACAC…: two keys in two buckets) — timed the same way (warm-up + steps of stage 1 +
stage 2, per-kernel clocks).  VERDICT r3 item 1(iv): "a read set with one planted poly-A / low-complexity bucket costs < +2 ms".

    python tools/lowcomplexity_probe.py [reads] [planted_fraction] > profiles/r04_lowcomplexity.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from megahit_amd import lib  # noqa: E402


def main():
    n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
    fracs = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.001, 0.01]
    n_reads = n_reads // 16 * 16
    packed = bench.make_reads(n_reads, 0, 1)
    words_per_read16 = bench.READ_LEN * 16 // 16  # 16 reads of 150 bases = 150 words
    eng = lib.Engine(0)

    def measure(words, label):
        eng.load_sequences(words, n_reads, bench.READ_LEN, None)
        for _ in range(2):
            eng.read2sdbg_s1(bench.K, bench.MIN_COUNT)
            eng.read2sdbg_s2(bench.K, bench.MIN_COUNT)
        eng.synchronize()
        eng.profile(True)
        eng.profile_reset()
        steps = 5
        t0 = time.perf_counter()
        for _ in range(steps):
            r1 = eng.read2sdbg_s1(bench.K, bench.MIN_COUNT)
            r2 = eng.read2sdbg_s2(bench.K, bench.MIN_COUNT)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = eng.profile_get()
        eng.profile(False)
        return {"label": label, "ms_per_step": round(dt * 1e3, 3), "s1_plan": eng.last_s1_plan(), "s1_items": int(r1.n_items), "sdbg_records": int(r2.n_sdbg),
                "kernel_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] / steps > 0.3}}

    out = {"reads": n_reads, "planted_fractions": fracs, "runs": []}
    out["runs"].append(measure(packed, "the bench library"))
    for frac in fracs:
        n_plant = int(n_reads * frac) // 16 * 16
        polya = packed.copy()
        polya[: n_plant // 16 * words_per_read16] = 0  # 2-bit A = 0: n_plant reads of 150 A's (133 records of one key each, all in lv1 bucket 0)
        out["runs"].append(measure(polya, "%d reads replaced by poly-A" % n_plant))
        del polya
        acac = packed.copy()
        acac[: n_plant // 16 * words_per_read16] = 0x11111111  # ACACAC... (A = 0, C = 1, MSB first)
        out["runs"].append(measure(acac, "%d reads replaced by (AC)n" % n_plant))
        del acac
    base = out["runs"][0]["ms_per_step"]
    for r in out["runs"][1:]:
        r["extra_ms"] = round(r["ms_per_step"] - base, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
