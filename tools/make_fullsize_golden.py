"""Known answers of the REFERENCE at BASELINE configs[1] size (10 M synthetic 150 bp PE reads, k=21, m=2).

Runs oracle/_ref/ref_core (= the reference's own sources compiled in place) on the deterministic read library of
tools/e2e_cli.py / bench.py rank 0 (genome seed 1, read seeds 1001+i) and stores the digests of the canonical streams
in tests/golden/fullsize.json.  The library itself is regenerated on the GPU box (numpy is deterministic), only the
digests travel.  Needs /root/reference to have been compiled (make -C oracle ref); ~25 min of 8-thread CPU time.

    python tools/make_fullsize_golden.py [--reads 1e7] [--threads 8] [--keep DIR]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megahit_amd import canon, synth  # noqa: E402


def gen_library(prefix, n_reads):
    """The bench.py / e2e_cli.py workload: one genome (seed 1, 2.5 bp per read), PE blocks of 1 M pairs (seed 1001+i)."""
    synth.write_pe_library(prefix, n_reads, 1, 1001)  # (blocks made by worker processes, written in order)


META = {"reads": 40000000, "genomes": 160, "genome_len": 2500000, "seed": 3, "k": 27, "m": 1}


def gen_meta_library(prefix, n_reads=META["reads"]):
    """BASELINE configs[4] ("500 M x 150 bp high-diversity metagenome, meta-large preset: k=27, -m 1") at single-GPU-shard
    size: 40 M reads, and 160 of the config's 2 000 genomes (2.5 Mbp each, log-normal abundances, sigma 1) so that the
    shard keeps the config's ~15x coverage — what one GPU of eight sees of its own lv1 buckets after the exchange."""
    blocks = synth.gen_metagenome_library(n_reads, META["genomes"] * n_reads // META["reads"] or 1, META["genome_len"], seed=META["seed"])
    synth.write_read_lib(prefix, blocks)


def meta_golden(args):
    """tests/golden/fullsize_meta.json: the reference's read2sdbg -k 27 -m 1 (stage 1 skipped, main_sdbg_build.cpp:139-147)"""
    n = int(args.reads) // 2 * 2 if args.reads != 1e7 else META["reads"]
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_core")
    d = args.keep or tempfile.mkdtemp(prefix="mhx_meta_")
    os.makedirs(d, exist_ok=True)
    lib = os.path.join(d, "reads")
    if not os.path.exists(lib + ".bin"):
        gen_meta_library(lib, n)
    k, m = META["k"], META["m"]
    out = {"reads": n, "k": k, "m": m, "edges": n * (150 - k), "genomes": META["genomes"] * n // META["reads"] or 1, "generator": "tools/make_fullsize_golden.py --preset meta",
           "reference_threads": args.threads, "lib_bin_md5": canon.digest_file(lib + ".bin"), "cases": {}}
    dt, _ = run([ref, "read2sdbg", "-k", str(k), "-m", str(m), "--host_mem", "%g" % args.host_mem, "--num_cpu_threads", str(args.threads),
                 "--read_lib_file", lib, "--output_prefix", os.path.join(d, "r2s")])
    c = sdbg_summary(os.path.join(d, "r2s"))
    c["wall_s"] = round(dt, 1)
    out["cases"]["read2sdbg"] = c
    print("read2sdbg -k %d -m %d" % (k, m), c, flush=True)
    path = args.out if args.out.endswith("_meta.json") else os.path.join(ROOT, "tests", "golden", "fullsize_meta.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


CONFIGS2 = {"reads": 100000000, "genome_seed": 2, "read_seed0": 2001, "k": 21, "m": 2}


def gen_configs2_library(prefix, n_reads=CONFIGS2["reads"], procs=None):
    """BASELINE configs[2] / the north-star size: 100 M x 150 bp PE reads of one 250 Mbp genome (seed 2; blocks 2001+i)."""
    synth.write_pe_library(prefix, n_reads, CONFIGS2["genome_seed"], CONFIGS2["read_seed0"], procs=procs)


def sdbg_octants(prefix):
    """Digests of the eight contiguous eighths of the SdBG's lv1 buckets: what rank r of 8 owns (tools/config_bench.py owner8)."""
    return [canon.digest_sdbg(prefix, r * 8192, (r + 1) * 8192) for r in range(8)]


def configs2_golden(args):
    """tests/golden/fullsize_100M.json: the reference's read2sdbg, count, seq2sdbg --need_mercy at the north-star size."""
    n = int(args.reads) // 2 * 2 if args.reads != 1e7 else CONFIGS2["reads"]
    k, m = CONFIGS2["k"], CONFIGS2["m"]
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_core")
    d = args.keep or tempfile.mkdtemp(prefix="mhx_c2_")
    os.makedirs(d, exist_ok=True)
    lib = os.path.join(d, "reads")
    if not os.path.exists(lib + ".bin"):
        gen_configs2_library(lib, n)
    path = args.out if args.out.endswith("_100M.json") else os.path.join(ROOT, "tests", "golden", "fullsize_100M.json")
    out = {"reads": n, "k": k, "m": m, "edges": n * (150 - k), "generator": "tools/make_fullsize_golden.py --preset configs2",
           "reference_threads": args.threads, "reference_host": "build container (%d cores)" % (os.cpu_count() or 0),
           "lib_bin_md5": canon.digest_file(lib + ".bin"), "cases": {}}
    if os.path.exists(path):
        old = json.load(open(path))
        if old.get("lib_bin_md5") == out["lib_bin_md5"]:
            out["cases"] = old.get("cases", {})

    def save():
        with open(path, "w") as f:
            json.dump(out, f, indent=1)

    common = ["-k", str(k), "-m", str(m), "--host_mem", "%g" % args.host_mem, "--num_cpu_threads", str(args.threads), "--read_lib_file", lib]
    if "read2sdbg" not in out["cases"]:
        dt, _ = run([ref, "read2sdbg"] + common + ["--output_prefix", os.path.join(d, "r2s")])
        c = sdbg_summary(os.path.join(d, "r2s"))
        c.update(wall_s=round(dt, 1), counting_md5=canon.digest_file(os.path.join(d, "r2s.counting")), octants=sdbg_octants(os.path.join(d, "r2s")))
        out["cases"]["read2sdbg"] = c
        print("read2sdbg", c, flush=True)
        save()
    if "count" not in out["cases"]:
        dt, _ = run([ref, "count"] + common + ["--output_prefix", os.path.join(d, "cnt")])
        hdr, _rows = canon.read_edges_info(os.path.join(d, "cnt"))
        c = {"digest": canon.digest_edges(os.path.join(d, "cnt")), "n_edges": hdr["num_edges"], "wall_s": round(dt, 1),
             "counting_md5": canon.digest_file(os.path.join(d, "cnt.counting")),
             "cand_md5": canon.digest_file(os.path.join(d, "cnt.cand"))}
        out["cases"]["count"] = c
        print("count", c, flush=True)
        save()
    if "seq2sdbg_need_mercy" not in out["cases"]:
        s2s = ["-k", str(k), "--kmer_from", "0", "--host_mem", "%g" % args.host_mem, "--num_cpu_threads", str(args.threads),
               "--input_prefix", os.path.join(d, "cnt")]
        dt, _ = run([ref, "seq2sdbg"] + s2s + ["--need_mercy", "--output_prefix", os.path.join(d, "s2m")])
        c = sdbg_summary(os.path.join(d, "s2m"))
        c["wall_s"] = round(dt, 1)
        out["cases"]["seq2sdbg_need_mercy"] = c
        print("seq2sdbg_need_mercy", c, flush=True)
        save()
    print("wrote", path)


def varlen_golden(args):
    """tests/golden/fullsize_varlen.json: the reference's read2sdbg -k 21 -m 2 on the configs[1] library with its reads TRIMMED
    (synth.VARLEN_RULES: every read cut to U[100, 150]; 2 % of the reads cut) — libraries whose reads are not of one length"""
    n = int(args.reads) // 2 * 2
    k, m = 21, 2
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_core")
    d = args.keep or tempfile.mkdtemp(prefix="mhx_var_")
    os.makedirs(d, exist_ok=True)
    path = args.out if args.out.endswith("_varlen.json") else os.path.join(ROOT, "tests", "golden", "fullsize_varlen.json")
    out = {"reads": n, "k": k, "m": m, "generator": "tools/make_fullsize_golden.py --preset varlen", "reference_threads": args.threads, "rules": {}}
    for rule in synth.VARLEN_RULES:
        lib = os.path.join(d, "reads_" + rule)
        if not os.path.exists(lib + ".bin"):
            n_reads, n_bases = synth.write_var_read_lib(lib, synth.varlen_blocks(n, rule))
        else:
            n_bases = int(open(lib + ".lib_info").read().split()[0])
        dt, _ = run([ref, "read2sdbg", "-k", str(k), "-m", str(m), "--host_mem", "%g" % args.host_mem, "--num_cpu_threads", str(args.threads),
                     "--read_lib_file", lib, "--output_prefix", os.path.join(d, "r2s_" + rule)])
        c = sdbg_summary(os.path.join(d, "r2s_" + rule))
        c.update(wall_s=round(dt, 1), bases=n_bases, lib_bin_md5=canon.digest_file(lib + ".bin"))
        out["rules"][rule] = c
        print(rule, c, flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print("wrote", path)


def run(cmd):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-3000:])
        raise SystemExit("command failed: " + " ".join(cmd))
    return dt, p.stderr


def sdbg_summary(prefix):
    hdr, rows = canon.read_sdbg_info(prefix)
    items = sum(r[3] for r in rows if r[0] != canon.NULL_ID)
    tips = sum(r[4] for r in rows if r[0] != canon.NULL_ID)
    large = sum(r[5] for r in rows if r[0] != canon.NULL_ID)
    return {"digest": canon.digest_sdbg(prefix), "n_sdbg": items, "n_tips": tips, "n_large": large}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=float, default=1e7)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--keep", default=None, help="work directory to keep (default: a temp dir)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "fullsize.json"))
    ap.add_argument("--preset", choices=["configs1", "meta", "configs2", "varlen"], default="configs1")
    ap.add_argument("--host_mem", type=float, default=48e9)
    args = ap.parse_args()
    if args.preset == "meta":
        return meta_golden(args)
    if args.preset == "configs2":
        return configs2_golden(args)
    if args.preset == "varlen":
        return varlen_golden(args)
    n = int(args.reads) // 2 * 2
    k, m = 21, 2
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_core")
    d = args.keep or tempfile.mkdtemp(prefix="mhx_full_")
    os.makedirs(d, exist_ok=True)
    lib = os.path.join(d, "reads")
    if not os.path.exists(lib + ".bin"):
        gen_library(lib, n)
    out = {"reads": n, "k": k, "m": m, "edges": n * (150 - k), "generator": "tools/make_fullsize_golden.py",
           "reference_threads": args.threads, "lib_bin_md5": canon.digest_file(lib + ".bin"), "cases": {}}
    common = ["-k", str(k), "-m", str(m), "--host_mem", "48e9", "--num_cpu_threads", str(args.threads), "--read_lib_file", lib]

    dt, _ = run([ref, "read2sdbg"] + common + ["--output_prefix", os.path.join(d, "r2s")])
    c = sdbg_summary(os.path.join(d, "r2s"))
    c.update(wall_s=round(dt, 1), counting_md5=canon.digest_file(os.path.join(d, "r2s.counting")))
    out["cases"]["read2sdbg"] = c
    print("read2sdbg", c, flush=True)

    dt, _ = run([ref, "count"] + common + ["--output_prefix", os.path.join(d, "cnt")])
    hdr, _rows = canon.read_edges_info(os.path.join(d, "cnt"))
    c = {"digest": canon.digest_edges(os.path.join(d, "cnt")), "n_edges": hdr["num_edges"], "wall_s": round(dt, 1),
         "counting_md5": canon.digest_file(os.path.join(d, "cnt.counting")),
         "cand_md5": canon.digest_file(os.path.join(d, "cnt.cand"))}
    out["cases"]["count"] = c
    print("count", c, flush=True)

    s2s = ["-k", str(k), "--kmer_from", "0", "--host_mem", "48e9", "--num_cpu_threads", str(args.threads),
           "--input_prefix", os.path.join(d, "cnt")]
    dt, _ = run([ref, "seq2sdbg"] + s2s + ["--output_prefix", os.path.join(d, "s2s")])
    c = sdbg_summary(os.path.join(d, "s2s"))
    c["wall_s"] = round(dt, 1)
    out["cases"]["seq2sdbg"] = c
    print("seq2sdbg", c, flush=True)

    # the orchestrator's default route: count, then seq2sdbg --need_mercy on the same prefix (reads .cand + the read lib)
    dt, _ = run([ref, "seq2sdbg"] + s2s + ["--need_mercy", "--output_prefix", os.path.join(d, "s2m")])
    c = sdbg_summary(os.path.join(d, "s2m"))
    c["wall_s"] = round(dt, 1)
    out["cases"]["seq2sdbg_need_mercy"] = c
    print("seq2sdbg_need_mercy", c, flush=True)

    dt, log = run([ref, "read2sdbg"] + common + ["--need_mercy", "--output_prefix", os.path.join(d, "r2m")])
    c = sdbg_summary(os.path.join(d, "r2m"))
    c["wall_s"] = round(dt, 1)
    for line in log.splitlines():
        if "Number mercy" in line:
            c["number_mercy"] = int(line.split(":")[-1].strip().split()[0])
    out["cases"]["read2sdbg_need_mercy"] = c
    print("read2sdbg_need_mercy", c, flush=True)

    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
