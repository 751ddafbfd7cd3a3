"""Helpers for the tests in which REFERENCE code consumes the files this framework writes (test infrastructure).

oracle/_ref/ref_megahit_core = the complete reference megahit_core compiled from /root/reference/src with assertions
on; oracle/_ref/harness = the reference's unmodified orchestrator script + test data staged by oracle/Makefile."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FULL = os.path.join(ROOT, "oracle", "_ref", "ref_megahit_core")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "harness")

_COMP = str.maketrans("ACGT", "TGCA")


def read_fasta(path):
    recs, name, seq = [], None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(seq)))
                name, seq = line[1:], []
            elif line:
                seq.append(line)
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


def canonical_contigs(path, kk=31):
    """Order-, strand- and rotation-independent view of a contig file: (sorted (length, flag, multi) of the contigs, the
    set of canonical kk-mers of all of them).  The reference prints a circular contig as the cycle plus an overlap,
    starting wherever its traversal happened to start (thread timing), so whole sequences are not comparable; the
    k-mer set, the lengths and the multiplicities are."""
    meta, kmers = [], set()
    for name, seq in read_fasta(path):
        m = re.search(r"flag=(\d+) multi=([0-9.]+)", name)
        meta.append((len(seq), m.group(1) if m else "", m.group(2) if m else ""))
        rc = seq.translate(_COMP)[::-1]
        n = len(seq)
        for i in range(n - kk + 1):
            a, b = seq[i:i + kk], rc[n - kk - i:n - i]
            kmers.add(a if a < b else b)
    return sorted(meta), kmers


def run_orchestrator(bin_dir, out_dir, extra=(), env=None):
    """The reference's `megahit` script from harness/<bin_dir>; returns (summary line, canonical final contigs)."""
    e = dict(os.environ)
    e.update(env or {})
    cmd = [sys.executable, os.path.join(HARNESS, bin_dir, "megahit"), "--test", "-t", "4", "--keep-tmp-files", "-o", out_dir] + list(extra)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e)
    assert p.returncode == 0, p.stdout[-3000:]
    summary = [l.split(" - ", 1)[1] for l in p.stdout.splitlines() if " contigs, total " in l]
    assert summary, p.stdout[-2000:]
    return summary[-1].strip(), canonical_contigs(os.path.join(out_dir, "final.contigs.fa"))
