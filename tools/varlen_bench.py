"""GPU: read2sdbg on libraries whose reads are NOT of one length, at BASELINE configs[1] size — the bench library (10 M x 150 bp)
with its reads trimmed (megahit_amd/synth.py VARLEN_RULES: "u100_150" every read cut to U[100, 150]; "trim2pct" 2 % of the reads
cut, the way N-trimming leaves a real library), loaded with a start[] array.  Timed like bench.py (warm-up + steps of stage 1 +
stage 2, the library's per-kernel clocks), on the generating pass with padded item slots (S1GenVarT, s1_var_fast = 1) and on the
extraction kernel + loaded passes (s1_var_fast = 0: what every such library took before round 5); ns per stage-1 record beside the
fixed-length library's; SdBG digest against the reference's (tests/golden/fullsize_varlen.json, tools/make_fullsize_golden.py
--preset varlen) and between the two paths.

    python tools/varlen_bench.py [reads] > profiles/r05_varlen.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from megahit_amd import canon, lib, synth  # noqa: E402


def main():
    n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
    n_reads = n_reads // 16 * 16
    eng = lib.Engine(0)
    try:
        with open(os.path.join(ROOT, "tests", "golden", "fullsize_varlen.json")) as f:
            golden = json.load(f)
    except Exception:
        golden = None
    steps = 5

    def timed(var_fast):
        eng.set_option("s1_var_fast", var_fast)
        for _ in range(2):
            eng.read2sdbg_s1(bench.K, bench.MIN_COUNT)
            eng.read2sdbg_s2(bench.K, bench.MIN_COUNT)
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
        digest = canon.digest_sdbg_buffers(bench.K, eng.fetch(lib.BUF_SDBG_BYTES, np.uint8), eng.fetch(lib.BUF_BUCKET_COUNT, np.uint64),
                                           eng.fetch(lib.BUF_BUCKET_TIPS, np.uint64), eng.fetch(lib.BUF_BUCKET_LARGE, np.uint64),
                                           eng.fetch(lib.BUF_BUCKET_OFFSET, np.uint64))
        s1_ms = sum(v["ms"] for k_, v in st.items() if k_.startswith("s1_") or k_.startswith("radix_scatter_12B") or k_ in ("item_counts", "bucket_bounds")) / steps
        return {"ms_per_step": round(dt * 1e3, 3), "s1_plan": eng.last_s1_plan(), "s1_records": int(r1.n_items), "sdbg_records": int(r2.n_sdbg),
                "stage1_kernel_ms": round(s1_ms, 3), "stage1_ns_per_record": round(s1_ms * 1e6 / max(1, int(r1.n_items)), 4), "digest": digest,
                "kernel_ms_per_step": {k_: round(v["ms"] / steps, 3) for k_, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] / steps > 0.2}}

    out = {"reads": n_reads, "k": bench.K, "m": bench.MIN_COUNT, "libraries": {}}
    packed = bench.make_reads(n_reads, 0, 1)
    eng.load_sequences(packed, n_reads, bench.READ_LEN, None)
    out["libraries"]["fixed 150 bp (the bench library)"] = timed(1)
    del packed
    for rule in synth.VARLEN_RULES:
        t0 = time.time()
        allb, alll = [], []  # one store for the whole library: the blocks' reversed trimmed reads concatenated base by base
        for reads, lens in synth.varlen_blocks(n_reads, rule):
            allb.append(reads)
            alll.append(lens)
        reads = np.concatenate(allb)
        lens = np.concatenate(alll)
        del allb, alll
        w, n_bases = synth.pack_var_reversed(reads, lens)
        start = np.concatenate([np.zeros(1, dtype=np.uint64), np.cumsum(lens, dtype=np.uint64)])
        del reads
        sys.stderr.write("%s: library made in %.1f s (%d bases)\n" % (rule, time.time() - t0, n_bases))
        eng.load_sequences(w, n_reads, 0, start)
        fast = timed(1)
        slow = timed(0)
        want = golden["rules"].get(rule) if golden and golden.get("reads") == n_reads else None
        out["libraries"][rule] = {"bases": n_bases, "mean_length": round(n_bases / n_reads, 2),
                                  "generating_pass_with_padded_slots": fast, "extraction_kernel_and_loaded_passes": slow,
                                  "digests_equal_between_paths": fast["digest"] == slow["digest"],
                                  "reference_digest": want["digest"] if want else None,
                                  "bit_identical_to_reference": (fast["digest"] == want["digest"] and fast["sdbg_records"] == want["n_sdbg"]) if want else None}
        del w
    fx = out["libraries"]["fixed 150 bp (the bench library)"]["stage1_ns_per_record"]
    for rule in synth.VARLEN_RULES:
        e = out["libraries"][rule]
        e["stage1_ns_per_record_vs_fixed"] = round(e["generating_pass_with_padded_slots"]["stage1_ns_per_record"] / fx, 3)
        e["step_speedup_over_extraction_path"] = round(e["extraction_kernel_and_loaded_passes"]["ms_per_step"] / e["generating_pass_with_padded_slots"]["ms_per_step"], 3)
    eng.set_option("s1_var_fast", 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
