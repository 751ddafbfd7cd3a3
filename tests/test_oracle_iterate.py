"""CPU: the Python restatement of `iterate` (oracle/iterate_oracle.py) against the reference's own iterate
(oracle/_ref/ref_megahit_core, reference sources) on seeded inputs: equal edge sets, flank counts and aligned-read counts."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import consume_util as cu
from megahit_amd import canon, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import iterate_oracle as io  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(cu.REF_FULL), reason="oracle/_ref/ref_megahit_core not built")


def make_case(d, k, seed, n_reads=700, genome_len=6000):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    genome[4000:4300] = genome[1000:1300]  # a repeat
    reads = []
    for _ in range(n_reads):
        ln = int(rng.integers(40, 161))
        a = int(rng.integers(0, genome.size - ln))
        r = genome[a:a + ln].copy()
        e = rng.random(ln) < 0.004
        r[e] = rng.integers(0, 4, size=int(e.sum()), dtype=np.uint8)
        reads.append(r if rng.random() < 0.5 else (3 - r[::-1]).astype(np.uint8))
    synth.write_read_lib(os.path.join(d, "reads"), [reads])
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    n, pos = 0, 0
    with open(os.path.join(d, "c.fa"), "wb") as f:
        while pos + k + 2 < genome.size:
            ln = int(rng.integers(k + 1, 5 * k))
            piece = genome[pos:pos + ln]
            if rng.random() < 0.5:
                piece = (3 - piece[::-1]).astype(np.uint8)
            flag = 0 if rng.random() < 0.85 else int(rng.integers(1, 3))
            f.write(b">k%d_%d flag=%d multi=9.0000 len=%d\n" % (k, n, flag, piece.size))
            f.write(lut[piece].tobytes() + b"\n")
            n += 1
            pos += ln - k if rng.random() < 0.8 else ln + int(rng.integers(0, 4))
        # two contigs with one start: the longer extension must win; one of exactly k+1 bases; a palindromic flank
        base = genome[100:100 + 3 * k]
        f.write(b">k%d_%d flag=0 multi=2.0000 len=%d\n%s\n" % (k, n, k + 3, lut[base[:k + 3]].tobytes()))
        f.write(b">k%d_%d flag=0 multi=2.0000 len=%d\n%s\n" % (k, n + 1, base.size, lut[base].tobytes()))
        f.write(b">k%d_%d flag=0 multi=2.0000 len=%d\n%s\n" % (k, n + 2, k + 1, lut[genome[200:200 + k + 1]].tobytes()))
        half = genome[300:300 + (k + 1) // 2]
        pal = np.concatenate([half, (3 - half[::-1]).astype(np.uint8)])[:k + 1]
        f.write(b">k%d_%d flag=0 multi=2.0000 len=%d\n%s\n" % (k, n + 3, pal.size + 5, lut[np.concatenate([pal, genome[:5]])].tobytes()))
    with open(os.path.join(d, "b.fa"), "wb") as f:
        for i in range(6):
            a = int(rng.integers(0, genome.size - 3 * k))
            piece = genome[a:a + 2 * k]
            f.write(b">k%d_%d flag=0 multi=2.0000 len=%d\n%s\n" % (k, 9000 + i, piece.size, lut[piece].tobytes()))


@pytest.mark.parametrize("k,step,seed", [(21, 8, 1), (21, 28, 2), (31, 2, 3), (39, 20, 4), (22, 6, 5)])
def test_python_iterate_equals_the_reference(tmp_path, k, step, seed):
    d = str(tmp_path)
    make_case(d, k, seed)
    out = os.path.join(d, "ref")
    p = subprocess.run([cu.REF_FULL, "iterate", "-c", os.path.join(d, "c.fa"), "-b", os.path.join(d, "b.fa"), "-t", "3", "-k", str(k), "-s", str(step),
                        "-o", out, "-r", os.path.join(d, "reads.bin")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-1500:]
    hdr, edges, _ = canon.canonical_edges(out)
    edges = np.ascontiguousarray(edges)
    ref = edges[np.lexsort(edges.T[::-1])] if edges.size else edges
    rows, wpe, n_flanks, aligned = io.iterate(os.path.join(d, "c.fa"), os.path.join(d, "b.fa"), os.path.join(d, "reads.bin"), k, step)
    assert wpe == hdr["words_per_edge"] and hdr["kmer_size"] == k + step
    assert rows.shape == ref.shape and np.array_equal(rows, ref)
    assert rows.shape[0] > 20
    flanks = [int(x) for x in re.findall(r"Number of flank kmers: (\d+)", p.stderr)]
    assert flanks[-1] == n_flanks
    assert int(re.search(r"Total: \d+, aligned: (\d+)", p.stderr).group(1)) == aligned
