"""Inputs for the buildlib tests (FASTA / FASTQ texts of every shape the reference's kseq reader accepts)."""
import gzip
import os

import numpy as np


def rand_seq(rng, n, p_n=0.0, lower=False):
    s = "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    if p_n:
        s = "".join("N" if rng.random() < p_n else ch for ch in s)
    return s.lower() if lower else s


def wrap(s, width):
    return "\n".join(s[i:i + width] for i in range(0, len(s), width)) if s else ""


def make_cases(d, seed=1):
    """-> {name: (lib file text, needs_sequential_parser)}; files are written under d"""
    rng = np.random.default_rng(seed)
    cases = {}

    def w(name, text, gz=False):
        p = os.path.join(d, name)
        if gz:
            with gzip.open(p, "wb") as f:
                f.write(text.encode())
        else:
            with open(p, "w") as f:
                f.write(text)
        return p

    seqs = [rand_seq(rng, int(rng.integers(1, 200)), p_n=0.02 if i % 3 == 0 else 0.0, lower=i % 7 == 0) for i in range(400)]
    seqs[5] = "NNNNNNNN"          # trimmed to nothing -> one fake 'A'
    seqs[6] = "NNACGTNNACGTACGT"  # first N-free stretch only
    seqs[7] = "ACGTRYKMACGT"      # other letters map to 'A'
    fa1 = "".join(">r%d some comment\n%s\n" % (i, s) for i, s in enumerate(seqs))
    cases["se_fasta"] = ("lib1\nse %s\n" % w("se.fa", fa1), False)
    fa_ml = "".join(">r%d\n%s\n" % (i, wrap(s, 60)) for i, s in enumerate(seqs)) + ">empty_record\n>last\nACGT"
    cases["se_fasta_multiline_no_final_newline"] = ("lib multi line\nse %s\n" % w("ml.fa", fa_ml), False)
    fq = "".join("@q%d/1 x\n%s\n+\n%s\n" % (i, s, "".join(chr(33 + int(v)) for v in rng.integers(0, 41, size=len(s)))) for i, s in enumerate(seqs))
    fq = fq.replace("+\n", "+q0\n", 1)
    cases["se_fastq"] = ("fq\nse %s\n" % w("se.fq", fq), False)
    cases["se_fastq_gz"] = ("fqgz\nse %s\n" % w("se.fq.gz", fq, gz=True), False)
    m1 = [rand_seq(rng, 100) for _ in range(300)]
    m2 = [rand_seq(rng, 100, p_n=0.01) for _ in range(300)]
    p1 = w("pe_1.fa", "".join(">p%d/1\n%s\n" % (i, s) for i, s in enumerate(m1)))
    p2 = w("pe_2.fa", "".join(">p%d/2\n%s\n" % (i, s) for i, s in enumerate(m2)))
    cases["pe_fasta"] = ("pe lib\npe %s %s\n" % (p1, p2), False)
    il = "".join(">i%d/1\n%s\n>i%d/2\n%s\n" % (i, a, i, b) for i, (a, b) in enumerate(zip(m1, m2)))
    cases["interleaved_plus_se"] = ("il\ninterleaved %s\nse again\nse %s\n" % (w("il.fa", il), w("se2.fa", fa1)), False)
    # shapes only the sequential parser takes
    cases["crlf"] = ("crlf\nse %s\n" % w("crlf.fa", fa1.replace("\n", "\r\n")), True)
    fq_ml = "".join("@m%d\n%s\n+\n%s\n" % (i, wrap(s, 50), wrap("I" * len(s), 50)) for i, s in enumerate(seqs[:60]))
    cases["fastq_multiline"] = ("fqml\nse %s\n" % w("ml.fq", fq_ml), True)
    cases["junk_before_header"] = ("junk\nse %s\n" % w("junk.fa", "garbage line\n" + fa1), True)
    fq_bad = fq + "@broken\nACGTACGT\n+\nIIII\n@after\nACGT\n+\nIIII\n"
    cases["fastq_truncated_quality"] = ("bad\nse %s\n" % w("bad.fq", fq_bad), True)
    p2s = w("pe_2_short.fa", "".join(">p%d/2\n%s\n" % (i, s) for i, s in enumerate(m2[:250])))
    cases["pe_unequal"] = ("pe lib\npe %s %s\n" % (p1, p2s), True)
    return cases
