"""Randomised libraries through stage 1 / count on super-k-mer records against the oracle (GPU).

    python tools/fuzz_skm.py [seconds] [seed]

Every round draws a library (fixed or ragged read lengths, planted poly-X / short-period repeats / duplicated reads), k in 19..22,
min count 1..2 and a set of knobs (bins, passes, table fill, probe limit, tags, dealing, cap), runs read2sdbg stage 1 + stage 2 and
count through the C ABI and compares every output with oracle/ (is_solid, histogram, item count, SdBG bytes and tables; edges, per-bucket
counts, first_0_out / last_0_in).  Prints a line per round; exits 1 at the first difference with the round's seed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from megahit_amd import lib, synth  # noqa: E402

RESET = dict(s1_skm=1, s1_stream_fill=7168, s1_stream_probes=1024, s1_skm_max_bin=65536, s1_skm_bin_bits=0, s1_skm_tags=0, s1_skm_cap_pct=36, s1_var_min_fill=50,
             s1_skm_passes=0, s1_skm_deal=1, s1_skm_hp=1, count_skm=1, count_skm_group=2)


def library(rng):
    L = int(rng.choice([30, 44, 60, 100, 150, 251]))
    n = int(rng.integers(200, 6000))
    g = int(rng.integers(max(600, 2 * L + 50), 20000))
    reads = [x for x in synth.gen_pe_reads(max(1, n // 2), g, read_len=L, frag=min(g - 10, max(L + 10, 2 * L)), err=float(rng.choice([0.0, 0.005, 0.02])), seed=int(rng.integers(1 << 30)))]
    if rng.random() < 0.5:  # ragged lengths
        reads = [r[: int(rng.integers(0, L + 1))] if rng.random() < 0.4 else r for r in reads]
    for unit in ([0], [3], [1], [2], [0, 1], [0, 3], [0, 1, 2], [1, 1, 2]):
        if rng.random() < 0.35:
            reads += [np.tile(np.array(unit, dtype=np.uint8), L)[: int(rng.integers(max(1, L // 2), L + 1))] for _ in range(int(rng.integers(1, 400)))]
    if rng.random() < 0.3:
        reads += [reads[i].copy() for i in rng.integers(0, len(reads), size=int(rng.integers(1, 300)))]
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


def knobs(rng):
    o = dict(s1_skm=2, s1_var_min_fill=1, s1_skm_max_bin=1 << 30, s1_skm_cap_pct=400)
    if rng.random() < 0.5:
        o["s1_skm_bin_bits"] = int(rng.choice([8, 9, 12, 16, 17, 20]))
    if rng.random() < 0.4:
        o["s1_skm_passes"] = int(rng.integers(2, 7))
    if rng.random() < 0.5:
        o["s1_stream_fill"] = int(rng.choice([2, 3, 17, 200, 4000]))
    if rng.random() < 0.2:
        o["s1_stream_probes"] = int(rng.choice([1, 2, 5]))
    if rng.random() < 0.3:
        o["s1_skm_tags"] = 1
    if rng.random() < 0.3:
        o["s1_skm_deal"] = 0
    if rng.random() < 0.2:
        o["s1_skm_hp"] = 0
    if rng.random() < 0.2:
        o["count_skm_group"] = 4
    if rng.random() < 0.15:
        o["s1_skm_max_bin"] = int(rng.choice([8, 200, 5000]))
    if rng.random() < 0.1:
        o["s1_skm_cap_pct"] = int(rng.choice([5, 20, 36]))
    return o


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    e = lib.Engine(0)
    t0, rounds, on_path = time.time(), 0, 0
    while time.time() - t0 < seconds:
        seed = seed0 + rounds
        rng = np.random.default_rng(seed)
        reads = library(rng)
        k, m = int(rng.integers(19, 23)), int(rng.integers(1, 3))
        opts = knobs(rng)
        pkg = ob.Package(reads, reverse=True)
        e.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
        for n_, v in opts.items():
            e.set_option(n_, v)
        bad = None
        try:
            w1 = ob.s1(pkg, k, m, tie_stable=True)
            r1 = e.read2sdbg_s1(k, m)
            plan1 = e.last_s1_plan()
            solid = e.fetch(lib.BUF_IS_SOLID, np.uint64)
            if r1.n_items != w1["n_items"] or not np.array_equal(solid, w1["is_solid"][: solid.size]) or not np.array_equal(e.fetch(lib.BUF_MUL_HIST, np.int64), w1["hist"]):
                bad = "stage 1"
            w2 = ob.s2(pkg, k, m, w1["is_solid"])
            e.read2sdbg_s2(k, m)
            if bad is None and (not np.array_equal(e.fetch(lib.BUF_SDBG_BYTES, np.uint8), w2["bytes"]) or
                                not np.array_equal(e.fetch(lib.BUF_BUCKET_COUNT, np.uint64), w2["bucket_items"]) or
                                not np.array_equal(e.fetch(lib.BUF_BUCKET_TIPS, np.uint64), w2["bucket_tips"])):
                bad = "stage 2"
            wc = ob.count(pkg, k, m)
            rc = e.count(k, m)
            planc = e.last_s1_plan()
            ed = e.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, rc.words_per_edge)
            if bad is None and (rc.n_items != wc["n_items"] or ed.shape != wc["edges"].shape or not np.array_equal(ed, wc["edges"]) or
                                not np.array_equal(e.fetch(lib.BUF_BUCKET_COUNT, np.uint64), wc["bucket_count"]) or
                                not np.array_equal(e.fetch(lib.BUF_MUL_HIST, np.int64), wc["hist"]) or
                                not np.array_equal(e.fetch(lib.BUF_FIRST_0_OUT, np.uint32), wc["first_0_out"]) or
                                not np.array_equal(e.fetch(lib.BUF_LAST_0_IN, np.uint32), wc["last_0_in"])):
                bad = "count"
        finally:
            for n_, v in RESET.items():
                e.set_option(n_, v)
        on_path += plan1.startswith("super-k-mers") + planc.startswith("count: super-k-mers")
        print("seed %d: %d reads, k %d, m %d, %s | %s | %s%s" % (seed, len(reads), k, m, " ".join("%s=%d" % kv for kv in sorted(opts.items()) if kv[0] not in ("s1_skm", "s1_var_min_fill")),
                                                                  plan1[:28], planc[:34], "  *** " + bad if bad else ""), flush=True)
        if bad:
            sys.exit(1)
        rounds += 1
    print("%d rounds, %d of %d stages on super-k-mer records, all equal to the oracle" % (rounds, on_path, 2 * rounds))


if __name__ == "__main__":
    main()
