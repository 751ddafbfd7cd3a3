"""Randomised libraries on 2-3 thread ranks through mhx_dist_read2sdbg with the super-k-mer exchange, against the oracle on the union (GPU).

    python tools/fuzz_skm_dist.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_binding as ob  # noqa: E402
from megahit_amd import lib  # noqa: E402
from fuzz_skm import library  # noqa: E402
from test_gpu_comm import run_ranks, sdbg_of, check_sdbg  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, rounds, on_path = time.time(), 0, 0
    while time.time() - t0 < seconds:
        seed = seed0 + rounds
        rng = np.random.default_rng(seed)
        world = int(rng.integers(2, 4))
        k = int(rng.integers(19, 23))
        shards = [library(np.random.default_rng(seed * 10 + r)) for r in range(world)]
        opts = dict(s1_skm=2, s1_var_min_fill=1, s1_skm_max_bin=1 << 30, s1_skm_cap_pct=400)
        if rng.random() < 0.5:
            opts["s1_skm_bin_bits"] = int(rng.choice([8, 9, 12, 16, 18, 20]))
        if rng.random() < 0.5:
            opts["s1_stream_fill"] = int(rng.choice([3, 17, 200, 4000]))
        if rng.random() < 0.3:
            opts["s1_skm_tags"] = 1
        if rng.random() < 0.3:
            opts["s1_skm_deal"] = 0
        if rng.random() < 0.15:
            opts["s1_skm_max_bin"] = int(rng.choice([8, 500]))

        def load(r, e):
            pkg = ob.Package(shards[r], reverse=True)
            e.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())

        def body(r, e, cm):
            cm.setup(int(seed % 2), k, 2)
            r1, r2, _ = cm.read2sdbg(k, 2)
            return sdbg_of(e) + (e.fetch(lib.BUF_MUL_HIST, np.int64), int(r1.n_solid), e.last_s1_plan(), int(r1.n_items))

        outs = run_ranks(world, load, body, opts)
        pkg = ob.Package(sum(shards, []), reverse=True)
        s1 = ob.s1(pkg, k, 2)
        try:
            assert np.array_equal(sum(o[4] for o in outs), s1["hist"]), "histogram"
            assert sum(o[5] for o in outs) == int(sum(bin(int(x)).count("1") for x in s1["is_solid"])), "n_solid"
            check_sdbg(outs, ob.s2(pkg, k, 2, s1["is_solid"]))
        except AssertionError as ex:
            print("seed %d: world %d k %d %s *** %s | %s" % (seed, world, k, opts, ex, outs[0][6]), flush=True)
            sys.exit(1)
        on_path += outs[0][6].startswith("super-k-mers")
        print("seed %d: world %d, %d reads, k %d, %s | %s" % (seed, world, sum(len(s) for s in shards), k,
                                                            " ".join("%s=%d" % kv for kv in sorted(opts.items()) if kv[0] not in ("s1_skm", "s1_var_min_fill", "s1_skm_cap_pct")), outs[0][6][:40]), flush=True)
        rounds += 1
    print("%d rounds, %d on the super-k-mer exchange, all equal to the oracle" % (rounds, on_path))


if __name__ == "__main__":
    main()
