"""GPU probe (round 6): no input may take minutes — what low-complexity reads cost on EVERY path a sub-program can take.
The 10 M-read library of bench.py, clean and with 1 % poly-A, 1 % (AC)n, 5 % poly-G reads planted, through
    count      k = 21 / 27, min count 2 / 3     (streaming design where it applies; the tile path beside it: count_stream = 0)
    read2sdbg  k = 27, min count 2              (bucket streaming with 64-bit table keys + stage 2 from a count of the (k+1)-mers; beside it
                                                 k_s1_seg + stage 2 per occurrence: s1_stream_wide = 0, s2_agg_from_count = 0)
    read2sdbg  k = 27, min count 1              (stage 1 skipped, stage 2 from a count; beside it stage 2 per occurrence)
each timed (warm-up + steps, per-kernel clocks); where two paths exist their outputs are compared (digest of every result buffer).

    python tools/lowcomplexity_paths_probe.py [reads] > profiles/r06_lowcomplexity_paths.json"""
import hashlib
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
    n_reads = (int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000) // 16 * 16
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    packed = bench.make_reads(n_reads, 0, 1)
    eng = lib.Engine(0)
    libs = [("clean", None, 0.0), ("1 % poly-A", 0, 0.01), ("1 % (AC)n", 0x11111111, 0.01), ("5 % poly-G", 0xAAAAAAAA, 0.05)]

    def dig(bufs):
        h = hashlib.md5()
        for b, t in bufs:
            h.update(eng.fetch(b, t).tobytes())
        return h.hexdigest()

    COUNT_BUFS = ((lib.BUF_EDGES, np.uint32), (lib.BUF_BUCKET_COUNT, np.uint64), (lib.BUF_MUL_HIST, np.int64), (lib.BUF_FIRST_0_OUT, np.uint32), (lib.BUF_LAST_0_IN, np.uint32))
    SDBG_BUFS = ((lib.BUF_SDBG_BYTES, np.uint8), (lib.BUF_BUCKET_COUNT, np.uint64), (lib.BUF_BUCKET_TIPS, np.uint64))

    def step_count(k, m):
        return lambda: eng.count(k, m)

    def step_r2s(k, m):
        def f():
            if m > 1:
                eng.read2sdbg_s1(k, m)
            return eng.read2sdbg_s2(k, m)
        return f

    cases = [("count k=21 m=2", step_count(21, 2), COUNT_BUFS, {"count_stream": 0}), ("count k=21 m=3", step_count(21, 3), COUNT_BUFS, {"count_stream": 0}),
             ("count k=27 m=2", step_count(27, 2), COUNT_BUFS, {"count_stream": 0}), ("count k=27 m=3", step_count(27, 3), COUNT_BUFS, {"count_stream": 0}),
             ("read2sdbg k=27 m=2", step_r2s(27, 2), SDBG_BUFS + ((lib.BUF_IS_SOLID, np.uint64), (lib.BUF_MUL_HIST, np.int64)), {"s1_stream_wide": 0, "s2_agg_from_count": 0}),
             ("read2sdbg k=27 m=1", step_r2s(27, 1), SDBG_BUFS, {"s2_agg_from_count": 0})]
    if only:
        cases = [c for c in cases if any(o in c[0] for o in only)]

    def timed(fn, steps):
        fn()
        eng.synchronize()
        eng.profile(True)
        eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = eng.profile_get()
        eng.profile(False)
        return dt, {k: round(v["ms"] / steps, 2) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])[:6]}

    out = {"reads": n_reads, "cases": {}}
    for name, fn, bufs, alt in cases:
        ent = {"libraries": {}}
        for lname, word, frac in libs:
            w = packed
            if word is not None:
                w = packed.copy()
                w[: int(n_reads * frac) // 16 * 150] = word
            eng.load_sequences(w, n_reads, bench.READ_LEN, None)
            dt, top = timed(fn, 2)
            r = {"ms_per_step": round(dt * 1e3, 2), "plan": eng.last_s1_plan(), "kernel_ms_top": top}
            if alt:
                d0 = dig(bufs)
                for o, v in alt.items():
                    eng.set_option(o, v)
                try:
                    dt2, top2 = timed(fn, 1)
                    r["other_path"] = {"options": alt, "ms_per_step": round(dt2 * 1e3, 2), "kernel_ms_top": top2, "outputs_equal": dig(bufs) == d0}
                finally:
                    for o in alt:
                        eng.set_option(o, 1)
            ent["libraries"][lname] = r
            sys.stderr.write("%s | %s | %s\n" % (name, lname, json.dumps(r)[:400]))
            sys.stderr.flush()
            if word is not None:
                del w
        base = ent["libraries"]["clean"]["ms_per_step"]
        ent["worst_over_clean"] = round(max(v["ms_per_step"] for v in ent["libraries"].values()) / base, 3)
        out["cases"][name] = ent
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
