"""GPU probe: what low-complexity reads cost the stage-1 fast path.  The 10 M-read library of bench.py, then the same library
with a fraction of its reads replaced by (a) poly-A reads — one lv1 bucket (AAAAAAAA) then holds ~133 records of ONE key per
planted read, and (b) a dinucleotide repeat (ACACAC…: two keys in two buckets) — timed the same way (warm-up + steps of stage 1 +
stage 2, per-kernel clocks).  VERDICT r3 item 1(iv): "a read set with one planted poly-A / low-complexity bucket costs < +2 ms".

Round 5: every planted library is also run on the round-4 paths (s1_giant = 0, sdbg_fast = 0) and the outputs compared.

    python tools/lowcomplexity_probe.py [reads] [planted_fractions] > profiles/r05_lowcomplexity.json"""
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
    fracs = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.001, 0.01, 0.05]
    n_reads = n_reads // 16 * 16
    packed = bench.make_reads(n_reads, 0, 1)
    words_per_read16 = bench.READ_LEN * 16 // 16  # 16 reads of 150 bases = 150 words
    eng = lib.Engine(0)

    def digest():
        import hashlib
        h = hashlib.md5()
        for b, t in ((lib.BUF_SDBG_BYTES, np.uint8), (lib.BUF_BUCKET_COUNT, np.uint64), (lib.BUF_BUCKET_TIPS, np.uint64), (lib.BUF_IS_SOLID, np.uint64),
                     (lib.BUF_MUL_HIST, np.int64)):
            h.update(eng.fetch(b, t).tobytes())
        return h.hexdigest()

    def timed(steps):
        eng.synchronize()
        eng.profile(True)
        eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            r1 = eng.read2sdbg_s1(bench.K, bench.MIN_COUNT)
            r2 = eng.read2sdbg_s2(bench.K, bench.MIN_COUNT)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = eng.profile_get()
        eng.profile(False)
        return dt, st, r1, r2

    def measure(words, label):
        eng.load_sequences(words, n_reads, bench.READ_LEN, None)
        for _ in range(2):
            eng.read2sdbg_s1(bench.K, bench.MIN_COUNT)
            eng.read2sdbg_s2(bench.K, bench.MIN_COUNT)
        steps = 5
        dt, st, r1, r2 = timed(steps)
        plan = eng.last_s1_plan()
        d_new = digest()
        # the same library on the paths of round 4 (a workgroup streams a giant bucket alone, the generic tile kernel emits the SdBG):
        # equal outputs (is_solid, histogram, SdBG bytes, per-bucket tables), and what the two changes are worth here
        eng.set_option("s1_giant", 0)
        eng.set_option("sdbg_fast", 0)
        try:
            eng.read2sdbg_s1(bench.K, bench.MIN_COUNT)
            eng.read2sdbg_s2(bench.K, bench.MIN_COUNT)
            dt_old, st_old, _, _ = timed(2)
            d_old = digest()
        finally:
            eng.set_option("s1_giant", 1)
            eng.set_option("sdbg_fast", 1)
        return {"label": label, "ms_per_step": round(dt * 1e3, 3), "s1_plan": plan, "s1_items": int(r1.n_items), "sdbg_records": int(r2.n_sdbg),
                "kernel_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] / steps > 0.3},
                "outputs_equal_round4_paths": d_new == d_old, "digest": d_new,
                "ms_per_step_round4_paths": round(dt_old * 1e3, 3),
                "kernel_ms_per_step_round4_paths": {k: round(v["ms"] / 2, 3) for k, v in sorted(st_old.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] / 2 > 0.3}}

    out = {"reads": n_reads, "planted_fractions": fracs, "runs": []}
    out["runs"].append(measure(packed, "the bench library"))
    for frac in fracs:
        n_plant = int(n_reads * frac) // 16 * 16
        # 2-bit codes, MSB first: poly-A = 0 (133 records of ONE key per planted read, all in lv1 bucket 0), (AC)n = 0x1111...,
        # poly-G = 0xAAAA... (what two-colour instruments emit; canonical form poly-C)
        for name, word in (("poly-A", 0), ("(AC)n", 0x11111111), ("poly-G", 0xAAAAAAAA)):
            if name == "poly-G" and frac < 0.05 and len(fracs) > 1:
                continue
            planted = packed.copy()
            planted[: n_plant // 16 * words_per_read16] = word
            out["runs"].append(measure(planted, "%d reads (%g %%) replaced by %s" % (n_plant, frac * 100, name)))
            del planted
    base = out["runs"][0]["ms_per_step"]
    for r in out["runs"][1:]:
        r["extra_ms"] = round(r["ms_per_step"] - base, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
