"""Known answers of the REFERENCE for BASELINE configs[3] at single-GPU-shard size: the iterative k-list
21,29,39,59,79,99,119 on 12.5 M synthetic 150 bp PE reads (genome seed 2, read seeds 2001+i, ~60x).

The reference's UNMODIFIED orchestrator (oracle/_ref/harness/bin/megahit -> ref_megahit_core, all sub-programs the
reference's own) assembles the reads with --keep-tmp-files.  For every k > 21 the seq2sdbg inputs it produced (the contig
files of the previous k written by `assemble`/`local`, the unsorted edge file written by `iterate`) are packed into
oracle/_ref/klist/k<K>/ (git-ignored like every other artefact built from the reference; travels to the GPU box with
the tree) and the digest of the SdBG the reference's seq2sdbg built from them goes to tests/golden/klist.json.
tests/test_gpu_klist.py runs mhx_core seq2sdbg on the packed inputs and compares.

    python tools/make_klist_golden.py [--reads 12.5e6] [--threads 8] [--work DIR]
"""
import argparse
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megahit_amd import canon, synth  # noqa: E402

KLIST = [21, 29, 39, 59, 79, 99, 119]
REPEAT_FAMILIES = 8000  # exact repeats of 25..140 bp planted in the genome (synth.plant_repeats): work for every k of the list


def sdbg_summary(prefix):
    _hdr, rows = canon.read_sdbg_info(prefix)
    live = [r for r in rows if r[0] != canon.NULL_ID]
    return {"digest": canon.digest_sdbg(prefix), "n_sdbg": sum(r[3] for r in live), "n_tips": sum(r[4] for r in live), "n_large": sum(r[5] for r in live)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=float, default=12.5e6)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--work", default="/tmp/mhx_klist")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "klist.json"))
    ap.add_argument("--pack", default=os.path.join(ROOT, "oracle", "_ref", "klist"))
    args = ap.parse_args()
    n = int(args.reads) // 2 * 2
    os.makedirs(args.work, exist_ok=True)
    fa = os.path.join(args.work, "reads.fa")
    if not os.path.exists(fa):
        _genome, blocks = synth.gen_shard_library(n, 2, 2001, repeat_families=REPEAT_FAMILIES)
        synth.write_fasta_interleaved(fa, blocks)
        del blocks
    out = os.path.join(args.work, "out")
    if not os.path.exists(os.path.join(out, "final.contigs.fa")):
        shutil.rmtree(out, ignore_errors=True)
        cmd = [sys.executable, os.path.join(ROOT, "oracle", "_ref", "harness", "bin", "megahit"), "--12", fa, "--k-list",
               ",".join(map(str, KLIST)), "-t", str(args.threads), "--keep-tmp-files", "-o", out]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            sys.stderr.write(p.stdout[-4000:])
            raise SystemExit("reference pipeline failed")
        print("reference pipeline: %.0f s" % (time.perf_counter() - t0), flush=True)
    log = open(os.path.join(out, "log")).read()
    doc = {"reads": n, "klist": KLIST, "generator": "tools/make_klist_golden.py", "genome_seed": 2, "read_seed0": 2001, "repeat_families": REPEAT_FAMILIES,
           "reference_threads": args.threads, "cases": {}}
    shutil.rmtree(args.pack, ignore_errors=True)
    # k = 21: count + seq2sdbg --need_mercy on the read library itself (regenerated on the GPU box: only digests travel)
    k21 = os.path.join(out, "tmp", "k21", "21")
    hdr, _rows = canon.read_edges_info(k21)
    doc["lib_bin_md5"] = canon.digest_file(os.path.join(out, "tmp", "reads.lib.bin"))
    doc["k21"] = {"count": {"digest": canon.digest_edges(k21), "n_edges": hdr["num_edges"], "counting_md5": canon.digest_file(k21 + ".counting"),
                            "cand_md5": canon.digest_file(k21 + ".cand")},
                  "seq2sdbg_need_mercy": sdbg_summary(k21)}
    print("k=21", doc["k21"], flush=True)
    for k in KLIST[1:]:
        m = re.search(r"command \S+ (seq2sdbg .*? -k %d .*)" % k, log)
        assert m, "no seq2sdbg command for k=%d in the log" % k
        argv = m.group(1).split()
        d = os.path.join(args.pack, "k%d" % k)
        os.makedirs(d)
        files = {}
        packed_args = []
        i = 0
        while i < len(argv):
            a = argv[i]
            if a in ("--contig", "--bubble", "--addi_contig", "--local_contig"):
                src = argv[i + 1]
                name = os.path.basename(src)
                for s, dn in ((src, name), (src + ".info", name + ".info")):
                    with open(s, "rb") as fi, gzip.open(os.path.join(d, dn + ".gz"), "wb", compresslevel=6) as fo:
                        shutil.copyfileobj(fi, fo, 1 << 22)
                files[a] = name
                packed_args += [a, name]
                i += 2
            elif a == "--input_prefix":
                src = argv[i + 1]
                name = os.path.basename(src)
                for ext in (".edges.info", ".edges.0"):
                    with open(src + ext, "rb") as fi, gzip.open(os.path.join(d, name + ext + ".gz"), "wb", compresslevel=1) as fo:
                        shutil.copyfileobj(fi, fo, 1 << 22)
                packed_args += [a, name]
                i += 2
            elif a in ("--output_prefix", "--host_mem", "--num_cpu_threads", "--mem_flag"):
                i += 2
            elif a in ("seq2sdbg", "--need_mercy"):
                packed_args.append(a)
                i += 1
            else:  # -k, --kmer_from
                packed_args += [a, argv[i + 1]]
                i += 2
        prefix = os.path.join(out, "tmp", "k%d" % k, str(k))
        t = re.search(r"Real: ([0-9.]+)", log[m.end():])  # the Real: line of this very run (utils.h:152)
        c = sdbg_summary(prefix)
        c.update(args=packed_args, reference_real_s=float(t.group(1)) if t else None,
                 packed_bytes=sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)))
        doc["cases"]["k%d" % k] = c
        print("k=%d" % k, doc["cases"]["k%d" % k], flush=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
