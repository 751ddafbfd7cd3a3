"""A/B of tuning knobs on ONE box, in ONE process: the 10 M-read library of bench.py is generated and uploaded once, then
every configuration (space-separated name=value lists, mhx_set_option knobs) runs warm-up + timed steps of read2sdbg with
the library's own per-kernel clocks, and the SdBG its last step left in HBM is digested against the reference's
(tests/golden/fullsize.json) — box-to-box variation is larger than most effects, and generating the reads costs more
than the steps.

    python tools/ab_options.py "sort_unit_runs=0" "sort_unit_runs=1" [--steps 6] [--engine read2sdbg|count]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=float, default=10e6)
    ap.add_argument("--engine", default="read2sdbg")
    ap.add_argument("--rounds", type=int, default=1, help="repeat the whole list (drift of the box shows)")
    ap.add_argument("--write-tuning", default=None, metavar="FILE",
                    help="write the knobs of the fastest configuration whose outputs matched the reference in every round to FILE "
                         "(megahit_amd/mhx_tuning.conf: the tuned defaults libmhx reads at mhx_create); the first configuration "
                         "is kept unless another one is faster by --min-gain-ms")
    ap.add_argument("--min-gain-ms", type=float, default=0.15)
    args = ap.parse_args()
    from megahit_amd import lib
    n_reads = int(args.reads) // 16 * 16
    t0 = time.time()
    packed = bench.make_reads(n_reads, 0, 1)
    print("generated %d reads in %.1f s" % (n_reads, time.time() - t0), file=sys.stderr, flush=True)
    eng = lib.Engine(0)
    eng.load_sequences(packed, n_reads, bench.READ_LEN, None)
    E = n_reads * (bench.READ_LEN - bench.K)
    out = []
    for rnd in range(args.rounds):
        for cfg in args.configs:
            opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in cfg.split() if kv)
            s1_only = bool(opts.pop("stage1_only", 0))  # (pseudo-knob of this tool: time stage 1 alone, no digest)
            for name, v in opts.items():  # (a knob keeps its value until a later configuration sets it again: list it in every one)
                eng.set_option(name, v)

            def step():
                if args.engine == "count":
                    return eng.count(bench.K, bench.MIN_COUNT), None
                r1 = eng.read2sdbg_s1(bench.K, bench.MIN_COUNT)
                return r1, (None if s1_only else eng.read2sdbg_s2(bench.K, bench.MIN_COUNT))
            for _ in range(args.warmup):
                res = step()
            eng.synchronize()
            eng.profile(True)
            eng.profile_reset()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                res = step()
            eng.synchronize()
            dt = time.perf_counter() - t0
            stats = eng.profile_get()
            eng.profile(False)
            par = None if s1_only else bench.output_parity(eng, args.engine, n_reads, 1, res)
            line = {"config": cfg, "round": rnd, "ms_per_step": round(dt / args.steps * 1e3, 3), "M_edges_per_s": round(E * args.steps / dt / 1e6, 1),
                    "parity_checked": bool(par["checked"]) if par else None,
                    "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] / args.steps >= 0.05}}
            out.append(line)
            print(json.dumps(line), flush=True)
    eng.close()
    if args.write_tuning:
        best = {}
        for line in out:
            b = best.setdefault(line["config"], {"ms": 1e30, "ok": True})
            b["ms"] = min(b["ms"], line["ms_per_step"])
            b["ok"] = b["ok"] and line["parity_checked"] is True
        base = args.configs[0]
        pick = base if best[base]["ok"] else None
        for cfg in args.configs[1:]:
            if best[cfg]["ok"] and (pick is None or best[cfg]["ms"] < best[pick]["ms"] - (args.min_gain_ms if pick == base else 0.0)):
                pick = cfg
        if pick is None:
            print("no configuration reproduced the reference: nothing written", file=sys.stderr)
            return
        knobs = [kv.split("=") for kv in pick.split() if kv and not kv.startswith("stage1_only")]
        with open(args.write_tuning, "w") as f:
            f.write("# tuned defaults of libmhx (read at mhx_create; explicit options and MHX_* environment variables win).\n"
                    "# Written by tools/ab_options.py from an A/B on the box: %s, %d x %d steps each; every line chooses between\n"
                    "# code paths with identical results (all configurations below matched the reference's digest).\n"
                    % (args.engine, args.rounds, args.steps))
            for cfg in args.configs:
                f.write("#   %-60s %.3f ms/step%s\n" % (cfg, best[cfg]["ms"], "" if best[cfg]["ok"] else "  (REJECTED: outputs differ)"))
            for name, v in knobs:
                f.write("%s = %d\n" % (name, int(v)))
        print("tuning: %s -> %s" % (pick, args.write_tuning), file=sys.stderr)


if __name__ == "__main__":
    main()
