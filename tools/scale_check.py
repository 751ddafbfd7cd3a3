"""GPU scale check (size-independent property, no oracle needed): read2sdbg on C copies of a read set with
min count C*m must give the same solid positions (repeated C times), the same multiplicity histogram (counts
scaled by C) and the same SdBG structure as one copy with min count m.  With 10 M reads and C = 4 this drives
5.3 G 16-byte stage-1 items (> 2^32 items, 6 G base positions > 2^32) through every kernel.

    python tools/scale_check.py [reads] [copies]      -> one JSON line
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from megahit_amd import lib

def main():
    n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_reads = n_reads // 16 * 16
    K, M = bench.K, bench.MIN_COUNT
    packed = bench.make_reads(n_reads, 0, 1)
    e = lib.Engine(0)


    def note(msg):
        print("[scale_check %.1f s] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


    T0 = time.perf_counter()


    plans = []

    def run(words, n, m):
        note("load %d reads" % n)
        e.load_sequences(words, n, bench.READ_LEN, None)
        t0 = time.perf_counter()
        e.profile(True)
        e.profile_reset()
        r1 = e.read2sdbg_s1(K, m)
        plans.append(e.last_s1_plan())
        note("S1 done: %d items of %d words; kernels ms: %s" % (r1.n_items, r1.item_words,
             {k2: round(v["ms"], 1) for k2, v in e.profile_get().items() if v["ms"] > 1}))
        e.profile_reset()
        r2 = e.read2sdbg_s2(K, m)
        note("S2 done: %d items; kernels ms: %s" % (r2.n_items, {k2: round(v["ms"], 1) for k2, v in e.profile_get().items() if v["ms"] > 1}))
        e.profile(False)
        dt = time.perf_counter() - t0
        return dict(r1=r1, r2=r2, dt=dt, solid=e.fetch(lib.BUF_IS_SOLID, np.uint64), hist=e.fetch(lib.BUF_MUL_HIST, np.int64),
                    items=e.fetch(lib.BUF_BUCKET_COUNT, np.uint64), tips=e.fetch(lib.BUF_BUCKET_TIPS, np.uint64),
                    wc=e.fetch(lib.BUF_W_COUNT, np.uint64))


    one = run(packed, n_reads, M)
    big = run(np.tile(packed, C), n_reads * C, M * C)
    assert (n_reads * bench.READ_LEN) % 64 == 0
    ok = {
        "solid_bitmap_repeats": bool(np.array_equal(big["solid"], np.tile(one["solid"], C))),
        "n_solid_scales": int(big["r1"].n_solid) == C * int(one["r1"].n_solid),
        "hist_scales": bool(np.array_equal(big["hist"][::C][: 65536 // C], one["hist"][: 65536 // C]) and big["hist"].sum() == one["hist"].sum()),
        "sdbg_records_equal": int(big["r2"].n_sdbg) == int(one["r2"].n_sdbg) and int(big["r2"].n_tips) == int(one["r2"].n_tips),
        "bucket_items_equal": bool(np.array_equal(big["items"], one["items"])),
        "bucket_tips_equal": bool(np.array_equal(big["tips"], one["tips"])),
        "w_counts_equal": bool(np.array_equal(big["wc"], one["wc"])),
    }
    print(json.dumps({"reads": n_reads, "copies": C, "k": K, "min_count": [M, M * C], "s1_items": [int(one["r1"].n_items), int(big["r1"].n_items)],
                      "s1_item_words": [int(one["r1"].item_words), int(big["r1"].item_words)], "seconds": [round(one["dt"], 3), round(big["dt"], 3)],
                      "s1_plans": plans, "checks": ok, "all_ok": all(ok.values())}))
    sys.exit(0 if all(ok.values()) else 1)


if __name__ == "__main__":
    main()
