"""Generate tests/golden/: inputs + known answers produced by the REFERENCE itself.

Runs oracle/_ref/ref_core (the reference's own sources compiled in place by oracle/Makefile) on
small inputs and records the canonical-stream digests of its outputs (megahit_amd/canon.py).
Needs /root/reference (for ref_core and test_data/r3_*.fa); the committed fixtures do not.

    python tools/make_golden.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megahit_amd import canon, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_core")
GOLD = os.path.join(ROOT, "tests", "golden")
REFDATA = "/root/reference/test_data"


def run(args, cwd):
    subprocess.run([REF] + args, cwd=cwd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def write_contigs(path, contigs, k):
    """contigs: list of (bases uint8, flag, multi). FASTA + .info sidecar as ContigWriter does
    (reference src/sequence/io/contig/contig_writer.h:26-34; .info = num_contigs num_bases)."""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "w") as f:
        for i, (s, flag, multi) in enumerate(contigs):
            f.write(">k%d_%d flag=%d multi=%.4f len=%d\n%s\n" % (k, i, flag, multi, len(s), lut[s].tobytes().decode()))
    with open(path + ".info", "w") as f:
        f.write("%d %d\n" % (len(contigs), sum(len(s) for s, _, _ in contigs)))


def main():
    os.makedirs(GOLD, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="mhx_golden_")
    gold = {"cases": []}

    # ---- inputs -------------------------------------------------------------
    # (1) BASELINE config 1: test_data/r3_1.fa + r3_2.fa through the reference's buildlib
    with open(os.path.join(tmp, "r3.txt"), "w") as f:
        f.write("r3\npe %s/r3_1.fa %s/r3_2.fa\n" % (REFDATA, REFDATA))
    run(["buildlib", "r3.txt", "r3"], tmp)
    # (2) high-coverage fixed-length reads (buckets > 64 items: exercises kmsort's radix path)
    hc = synth.gen_pe_reads(2000, 4000, read_len=100, frag=260, err=0.01, seed=21)
    synth.write_read_lib(os.path.join(tmp, "hc"), [hc])
    # (3) ragged: variable lengths incl. reads shorter than k+1, an empty read, low-complexity reads
    rng = np.random.default_rng(22)
    base = synth.gen_pe_reads(700, 3000, read_len=120, frag=300, err=0.02, seed=23)
    rag = [r[: rng.integers(0, 121)] for r in base]
    rag += [np.zeros(150, dtype=np.uint8)] * 30 + [np.tile(np.array([0, 3], dtype=np.uint8), 60)] * 20
    rag += [np.zeros(0, dtype=np.uint8), np.full(25, 2, dtype=np.uint8)]
    synth.write_read_lib(os.path.join(tmp, "rag"), [rag], paired=False)
    for name in ["r3", "hc", "rag"]:
        for ext in [".bin", ".lib_info"]:
            shutil.copy(os.path.join(tmp, name + ext), os.path.join(GOLD, name + ext))

    # ---- count / read2sdbg / seq2sdbg known answers --------------------------
    def record(case, prefix, kinds):
        ent = {"case": case}
        p = os.path.join(tmp, prefix)
        if "edges" in kinds:
            ent["edges"] = canon.digest_edges(p)
            ent["cand"] = canon.digest_file(p + ".cand")
            ent["counting"] = canon.digest_file(p + ".counting")
            hdr, data, _ = canon.canonical_edges(p)
            ent["n_edges"] = int(data.shape[0])
        if "sdbg" in kinds:
            ent["sdbg"] = canon.digest_sdbg(p)
            hdr, buckets = canon.canonical_sdbg(p)
            ent["n_sdbg"] = int(sum(b[1] for b in buckets))
        if "counting" in kinds:
            ent["counting"] = canon.digest_file(p + ".counting")
        if "mercy_cand" in kinds:
            import hashlib
            ent["mercy_cand_kmsort"] = hashlib.md5(canon.sorted_mercy_cand(p).tobytes()).hexdigest()
        gold["cases"].append(ent)

    common = ["--host_mem", "2e9", "--num_cpu_threads", "3"]
    for lib, k, m in [("r3", 21, 2), ("hc", 21, 2), ("hc", 27, 3), ("rag", 21, 2), ("rag", 31, 2), ("rag", 32, 2), ("hc", 61, 2)]:
        tag = "%s_k%d_m%d" % (lib, k, m)
        run(["count", "-k", str(k), "-m", str(m), "--read_lib_file", lib, "--output_prefix", "cnt_" + tag] + common, tmp)
        record({"prog": "count", "lib": lib, "k": k, "m": m}, "cnt_" + tag, ["edges"])
        run(["read2sdbg", "-k", str(k), "-m", str(m), "--read_lib_file", lib, "--output_prefix", "r2s_" + tag] + common, tmp)
        record({"prog": "read2sdbg", "lib": lib, "k": k, "m": m, "mercy": False}, "r2s_" + tag, ["sdbg", "counting", "mercy_cand"])
        run(["read2sdbg", "-k", str(k), "-m", str(m), "--read_lib_file", lib, "--output_prefix", "r2m_" + tag, "--need_mercy"] + common, tmp)
        record({"prog": "read2sdbg", "lib": lib, "k": k, "m": m, "mercy": True}, "r2m_" + tag, ["sdbg"])
        for mercy in (False, True):
            o = "s2s_%s_%d" % (tag, mercy)
            run(["seq2sdbg", "-k", str(k), "--kmer_from", "0", "--input_prefix", "cnt_" + tag, "--output_prefix", o] + common +
                (["--need_mercy"] if mercy else []), tmp)
            record({"prog": "seq2sdbg", "lib": lib, "k": k, "m": m, "mercy": mercy, "input": "count"}, o, ["sdbg"])
    for lib, k in [("rag", 21), ("hc", 27), ("r3", 21)]:
        tag = "%s_k%d_m1" % (lib, k)
        run(["read2sdbg", "-k", str(k), "-m", "1", "--read_lib_file", lib, "--output_prefix", "r2s_" + tag] + common, tmp)
        record({"prog": "read2sdbg", "lib": lib, "k": k, "m": 1, "mercy": False}, "r2s_" + tag, ["sdbg"])

    # ---- seq2sdbg at the next k: contigs (+ loop), bubble, addi, local + unsorted edges --------
    rng = np.random.default_rng(31)
    genome = rng.integers(0, 4, size=5000, dtype=np.uint8)

    def piece(a, L, rc=False):
        s = genome[a:a + L].copy()
        return (3 - s)[::-1].copy() if rc else s
    for k_from, k in [(21, 29), (29, 39), (59, 79), (99, 119)]:
        contigs = [(piece(rng.integers(0, 4000), int(rng.integers(k - 5, 700)), rng.random() < .5), 1 if i % 3 else 0,
                    float(rng.uniform(0.6, 300))) for i in range(40)]
        loop = piece(100, k + 40)
        contigs.append((loop, 2, 7.5))            # loop contig: extended by s[k_from..k) (contig_reader.h:73-86)
        contigs.append((piece(200, k + 1), 3, 70000.0))
        contigs.append((piece(300, k), 1, 5.0))   # too short: skipped
        bubble = [(piece(rng.integers(0, 4000), k + 30), 1, 2.25) for _ in range(5)]
        addi = [(piece(rng.integers(0, 4000), k + 60, True), 0, 1.49) for _ in range(6)]
        local = [(piece(rng.integers(0, 4000), 2 * k), 1, 254.5) for _ in range(4)]
        d = "ctg_k%d" % k
        for nm, cs in [("contigs", contigs), ("bubble", bubble), ("addi", addi), ("local", local)]:
            write_contigs(os.path.join(tmp, "%s.%s.fa" % (d, nm)), cs, k_from)
            shutil.copy(os.path.join(tmp, "%s.%s.fa" % (d, nm)), GOLD)
            shutil.copy(os.path.join(tmp, "%s.%s.fa.info" % (d, nm)), GOLD)
        # unsorted (k+1)-mer edges as `iterate` writes them (edge_writer.h:48-53,94-99)
        wpe = (2 * (k + 1) + 16 + 31) // 32
        n_e = 300
        ed = np.zeros((n_e, wpe), dtype=np.uint32)
        for i in range(n_e):
            s = piece(int(rng.integers(0, 4900 - k)), k + 1, rng.random() < .5)
            for j, c in enumerate(s):
                ed[i, j >> 4] |= np.uint32(int(c) << (30 - 2 * (j & 15)))
            ed[i, wpe - 1] |= np.uint32(int(rng.integers(1, 400)))
        ed.tofile(os.path.join(tmp, d + ".edges.0"))
        with open(os.path.join(tmp, d + ".edges.info"), "w") as f:
            f.write("kmer_size %d\nwords_per_edge %d\nnum_files 1\nnum_buckets 0\nnum_edges %d\nis_sorted 0\n" % (k, wpe, n_e))
        shutil.copy(os.path.join(tmp, d + ".edges.0"), GOLD)
        shutil.copy(os.path.join(tmp, d + ".edges.info"), GOLD)
        args = ["seq2sdbg", "-k", str(k), "--kmer_from", str(k_from), "--input_prefix", d, "--contig", d + ".contigs.fa", "--bubble",
                d + ".bubble.fa", "--addi_contig", d + ".addi.fa", "--local_contig", d + ".local.fa", "--output_prefix", d + "_out"]
        run(args + common, tmp)
        record({"prog": "seq2sdbg", "k": k, "k_from": k_from, "input": "contigs"}, d + "_out", ["sdbg"])
        run(["seq2sdbg", "-k", str(k), "--kmer_from", str(k_from), "--contig", d + ".contigs.fa", "--bubble", d + ".bubble.fa",
             "--output_prefix", d + "_out2"] + common, tmp)
        record({"prog": "seq2sdbg", "k": k, "k_from": k_from, "input": "contigs_only"}, d + "_out2", ["sdbg"])

    with open(os.path.join(GOLD, "golden.json"), "w") as f:
        json.dump(gold, f, indent=1)
    shutil.rmtree(tmp)
    print("wrote %d cases to %s" % (len(gold["cases"]), GOLD))


if __name__ == "__main__":
    main()
