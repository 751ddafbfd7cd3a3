"""GPU: SURVEY.md section 8f N2 — `mhx_core iterate` (flank records sorted + searched, one thread per read, sort + unique)
writes the same iterative edges as the reference's iterate (oracle/_ref/ref_megahit_core: iterate/contig_flank_index.h,
iterate/kmer_collector.h, main_iterate.cpp).  The reference's output order is its hash set's iteration order, so the
edge files are compared as sorted record sets; the .edges.info headers must be equal."""
import os
import subprocess

import numpy as np
import pytest

import consume_util as cu
import golden_util as gu
from megahit_amd import canon, synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(cu.REF_FULL), reason="oracle/_ref/ref_megahit_core not built")]


def run_iterate(exe, contigs, bubble, reads_bin, k, step, out, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([exe, "iterate", "-c", contigs, "-b", bubble, "-t", "4", "-k", str(k), "-s", str(step), "-o", out, "-r", reads_bin],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    hdr, edges, _ = canon.canonical_edges(out)
    edges = np.ascontiguousarray(edges)
    order = np.lexsort(edges.T[::-1]) if edges.size else np.zeros(0, dtype=np.int64)
    return hdr, edges[order]


def compare(contigs, bubble, reads_bin, k, step, d):
    hr, er = run_iterate(cu.REF_FULL, contigs, bubble, reads_bin, k, step, os.path.join(d, "ref_%d" % k))
    hm, em = run_iterate(gu.MHX_CORE, contigs, bubble, reads_bin, k, step, os.path.join(d, "mhx_%d" % k))
    assert hm == hr
    assert em.shape == er.shape and np.array_equal(em, er)
    return er.shape[0]


@pytest.fixture(scope="module")
def pipeline(tmp_path_factory):
    """the reference pipeline on the reference's --test data, temporary files kept: contigs of every round + the read library"""
    if not os.path.exists(os.path.join(cu.HARNESS, "bin", "megahit")):
        pytest.skip("oracle/_ref/harness not staged")
    d = str(tmp_path_factory.mktemp("iter"))
    cu.run_orchestrator("bin", os.path.join(d, "out"))
    return os.path.join(d, "out")


@pytest.mark.parametrize("k,step", [(21, 8), (29, 10), (39, 20), (59, 20)])
def test_iterate_on_the_reference_test_pipeline(pipeline, tmp_path, k, step):
    ctg = os.path.join(pipeline, "intermediate_contigs", "k%d.contigs.fa" % k)
    bub = os.path.join(pipeline, "intermediate_contigs", "k%d.bubble_seq.fa" % k)
    if not os.path.exists(ctg):
        pytest.skip("the pipeline ended before k=%d" % k)
    compare(ctg, bub, os.path.join(pipeline, "tmp", "reads.lib.bin"), k, step, str(tmp_path))


@pytest.mark.parametrize("k,step,seed", [(21, 8, 31), (21, 28, 32), (27, 12, 33)])
def test_iterate_on_synthetic_contigs(tmp_path, k, step, seed):
    """a larger case: 20 K reads (errors, variable lengths) over a 50 kb genome with a repeat; contigs = the reference's own
    assemble output at k"""
    d = str(tmp_path)
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=50000, dtype=np.uint8)
    genome[30000:31500] = genome[5000:6500]  # a repeat longer than k+step: contigs break around it
    reads = synth.gen_pe_reads(10000, genome.size, read_len=120, frag=300, err=0.01, seed=seed, genome=genome)
    reads = [r[: int(rng.integers(60, 121))] for r in reads]
    synth.write_read_lib(os.path.join(d, "reads"), [reads])  # one block of variable-length reads
    common = ["-k", str(k), "-m", "2", "--host_mem", "4e9", "--num_cpu_threads", "3", "--read_lib_file", os.path.join(d, "reads")]
    subprocess.run([gu.REF_CORE, "read2sdbg"] + common + ["--output_prefix", os.path.join(d, "g")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([cu.REF_FULL, "assemble", "-s", os.path.join(d, "g"), "-o", os.path.join(d, "asm"), "-t", "2", "--min_standalone", "200",
                    "--prune_level", "2", "--merge_len", "20", "--merge_similar", "0.95", "--cleaning_rounds", "5", "--disconnect_ratio", "0.1",
                    "--low_local_ratio", "0.2", "--min_depth", "2", "--bubble_level", "2", "--max_tip_len", "-1", "--careful_bubble"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    n = compare(os.path.join(d, "asm.contigs.fa"), os.path.join(d, "asm.bubble_seq.fa"), os.path.join(d, "reads.bin"), k, step, d)
    assert n > 0


def write_overlapping_contigs(path_ctg, path_bub, genome, k, rng):
    """pieces of the genome that overlap by k bases (as unitigs do), half of them reverse-complemented, written as the
    assembler writes contigs; a few short 'bubble' sequences"""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    n, pos = 0, 0
    with open(path_ctg, "wb") as f:
        while pos + k + 2 < genome.size:
            ln = int(rng.integers(k + 1, 4 * k))
            piece = genome[pos:pos + ln]
            if rng.random() < 0.5:
                piece = (3 - piece[::-1]).astype(np.uint8)
            flag = 0 if rng.random() < 0.9 else int(rng.integers(1, 3))  # some standalone / loop contigs: dropped by iterate
            f.write(b">k%d_%d flag=%d multi=12.0000 len=%d\n" % (k, n, flag, piece.size))
            f.write(lut[piece].tobytes() + b"\n")
            n += 1
            pos += ln - k if rng.random() < 0.85 else ln + int(rng.integers(0, 5))
    with open(path_bub, "wb") as f:
        for i in range(20):
            a = int(rng.integers(0, genome.size - 3 * k))
            piece = genome[a:a + int(rng.integers(k + 1, 2 * k))]
            f.write(b">k%d_%d flag=0 multi=2.0000 len=%d\n" % (k, n + i, piece.size))
            f.write(lut[piece].tobytes() + b"\n")
    return n


@pytest.mark.parametrize("k,step,seed", [(79, 20, 41), (119, 22, 42), (99, 28, 43), (31, 2, 44)])
def test_iterate_wide_kmers(tmp_path, k, step, seed):
    """keys of up to 8 words, new k-mers of up to 9: contigs cut from the genome, reads of 150..220 bases with errors"""
    d = str(tmp_path)
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=30000, dtype=np.uint8)
    reads = []
    for _ in range(6000):
        ln = int(rng.integers(150, 221))
        a = int(rng.integers(0, genome.size - ln))
        r = genome[a:a + ln].copy()
        e = rng.random(ln) < 0.003
        r[e] = rng.integers(0, 4, size=int(e.sum()), dtype=np.uint8)
        reads.append(r if rng.random() < 0.5 else (3 - r[::-1]).astype(np.uint8))
    synth.write_read_lib(os.path.join(d, "reads"), [reads])
    write_overlapping_contigs(os.path.join(d, "c.fa"), os.path.join(d, "b.fa"), genome, k, rng)
    n = compare(os.path.join(d, "c.fa"), os.path.join(d, "b.fa"), os.path.join(d, "reads.bin"), k, step, d)
    assert n > 100


@pytest.mark.parametrize("k,step,seed", [(21, 8, 1), (22, 6, 5), (39, 20, 4)])
def test_iterate_equals_the_python_oracle(tmp_path, k, step, seed):
    """the same inputs as tests/test_oracle_iterate.py (where the restatement is pinned to the reference): contigs with
    competing flanks, a contig of exactly k+1 bases, a palindromic flank, dropped loop / standalone contigs"""
    import sys
    sys.path.insert(0, os.path.join(gu.ROOT, "oracle"))
    import iterate_oracle as io
    import test_oracle_iterate as toi
    d = str(tmp_path)
    toi.make_case(d, k, seed)
    hm, em = run_iterate(gu.MHX_CORE, os.path.join(d, "c.fa"), os.path.join(d, "b.fa"), os.path.join(d, "reads.bin"), k, step, os.path.join(d, "mhx"))
    rows, wpe, _n_flanks, _aligned = io.iterate(os.path.join(d, "c.fa"), os.path.join(d, "b.fa"), os.path.join(d, "reads.bin"), k, step)
    assert hm["words_per_edge"] == wpe and hm["kmer_size"] == k + step and hm["is_sorted"] == 0
    assert em.shape == rows.shape and np.array_equal(em, rows)
