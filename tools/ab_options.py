"""A/B of tuning knobs on ONE box, in ONE process: the 10 M-read library of bench.py is generated and uploaded once, then
every configuration (space-separated name=value lists, mhx_set_option knobs) runs warm-up + timed steps of read2sdbg with
the library's own per-kernel clocks, and the SdBG its last step left in HBM is digested against the reference's
(tests/golden/fullsize.json) — box-to-box variation is larger than most effects, and generating the reads costs more
than the steps.

    python tools/ab_options.py "sort_unit_runs=0" "sort_unit_runs=1" [--steps 6] [--engine read2sdbg|count] [--rounds 2]
    python tools/ab_options.py "s1_gen_blocked=0" --greedy "s1_gen_blocked s1_digit_hist_preload" --write-tuning megahit_amd/mhx_tuning.conf

--write-tuning FILE stores the winner as the installation's tuned defaults (libmhx reads mhx_tuning.conf beside libmhx.so
at mhx_create; include/mhx.h: mhx_get_option).  Only configurations whose outputs matched the reference's digest qualify.
--greedy KNOBS: starting from the first configuration, each knob of the list is switched on in turn ("name", or "name=value" for
another value than 1) and kept when the step gets faster by --min-gain-ms.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def parse(cfg):
    return dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in cfg.split() if kv)


def fmt(opts):
    return " ".join("%s=%d" % kv for kv in opts.items())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=float, default=10e6)
    ap.add_argument("--engine", default="read2sdbg")
    ap.add_argument("--rounds", type=int, default=1, help="repeat the whole list (drift of the box shows)")
    ap.add_argument("--greedy", default=None, metavar="KNOBS", help="knob names to switch on one after the other, starting from the first configuration")
    ap.add_argument("--write-tuning", default=None, metavar="FILE")
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
    lines = []

    def measure(opts, rnd=0, note=None):
        opts = dict(opts)
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
        line = {"config": fmt(opts), "round": rnd, "ms_per_step": round(dt / args.steps * 1e3, 3), "M_edges_per_s": round(E * args.steps / dt / 1e6, 1),
                "parity_checked": bool(par["checked"]) if par else None,
                "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] / args.steps >= 0.05}}
        if note:
            line["note"] = note
        lines.append(line)
        print(json.dumps(line), flush=True)
        return line

    pick, report = None, []
    if args.greedy:
        cur = parse(args.configs[0])
        trials = []  # (knob, value to try): "name" = switch it on, "name=8" = that value (its starting value belongs in the first configuration)
        for item in args.greedy.split():
            name, _, val = item.partition("=")
            trials.append((name, int(val) if val else 1))
            cur.setdefault(name, 0)
        base = measure(cur, note="greedy: start")
        t_cur, ok = base["ms_per_step"], base["parity_checked"] is True
        report.append((fmt(cur), t_cur, ok))
        if ok:
            for name, val in trials:
                trial = dict(cur)
                trial[name] = val
                ln = measure(trial, note="greedy: try %s=%d" % (name, val))
                good = ln["parity_checked"] is True and ln["ms_per_step"] < t_cur - args.min_gain_ms
                report.append((fmt(trial), ln["ms_per_step"], ln["parity_checked"] is True))
                if good:
                    cur, t_cur = trial, ln["ms_per_step"]
            last = measure(cur, rnd=1, note="greedy: result")
            if last["parity_checked"] is True:
                pick = cur
                report.append((fmt(cur) + "   (result, measured again)", last["ms_per_step"], True))
    else:
        for rnd in range(args.rounds):
            for cfg in args.configs:
                measure(parse(cfg), rnd)
        best = {}
        for line in lines:
            b = best.setdefault(line["config"], {"ms": 1e30, "ok": True})
            b["ms"] = min(b["ms"], line["ms_per_step"])
            b["ok"] = b["ok"] and line["parity_checked"] is True
        order = [fmt(parse(c)) for c in args.configs]
        base = order[0]
        choice = base if best[base]["ok"] else None
        for cfg in order[1:]:
            if best[cfg]["ok"] and (choice is None or best[cfg]["ms"] < best[choice]["ms"] - (args.min_gain_ms if choice == base else 0.0)):
                choice = cfg
        report = [(cfg, best[cfg]["ms"], best[cfg]["ok"]) for cfg in order]
        pick = parse(choice) if choice else None
    eng.close()
    if args.write_tuning:
        if pick is None:
            print("no configuration reproduced the reference: nothing written", file=sys.stderr)
            return
        pick.pop("stage1_only", None)
        with open(args.write_tuning, "w") as f:
            f.write("# tuned defaults of libmhx (read at mhx_create; explicit options and MHX_* environment variables win).\n"
                    "# Written by tools/ab_options.py from an A/B on the box (%s, %d timed steps per line); every knob chooses between\n"
                    "# code paths with identical results, and only configurations that matched the reference's digest qualify.\n"
                    % (args.engine, args.steps))
            for cfg, ms, ok in report:
                f.write("#   %8.3f ms/step%s  %s\n" % (ms, "" if ok else "  REJECTED (outputs differ)", cfg))
            for name, v in pick.items():
                f.write("%s = %d\n" % (name, int(v)))
        print("tuning: %s -> %s" % (fmt(pick), args.write_tuning), file=sys.stderr)


if __name__ == "__main__":
    main()
