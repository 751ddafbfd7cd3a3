"""Seeded inputs shared by the multi-rank tests (thread ranks: test_gpu_comm.py; process ranks: test_gpu_multiprocess.py)."""
import numpy as np


def reads_of(seed, n_pairs=250):
    from megahit_amd import synth
    rng = np.random.default_rng(seed)
    genome = np.random.default_rng(99).integers(0, 4, size=3000, dtype=np.uint8)
    r = synth.gen_pe_reads(n_pairs, 3000, read_len=100, frag=250, err=0.01, seed=seed, genome=genome)
    return [x[: rng.integers(10, 101)] for x in r]


def seqs_with_mult(rank_seed):
    """seq2sdbg input: (k+1)-mer-or-longer sequences with multiplicities, as edges/contigs would be"""
    rng = np.random.default_rng(rank_seed)
    genome = np.random.default_rng(7).integers(0, 4, size=4000, dtype=np.uint8)
    seqs, mult = [], []
    for _ in range(300):
        L = int(rng.integers(5, 90))
        o = int(rng.integers(0, genome.size - L))
        seqs.append(genome[o:o + L].copy())
        mult.append(int(rng.integers(1, 400)))
    return seqs, np.array(mult, dtype=np.uint16)
