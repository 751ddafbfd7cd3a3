"""Synthetic read sets for the SdBG-construction benchmarks.

Genome = i.i.d. uniform ACGT; PE fragments: start ~ U[0, G-frag), read 1 = forward `read_len`
bases at start, read 2 = reverse complement of the `read_len` bases ending at start+frag;
substitution errors i.i.d. with probability `err`, replaced by a uniformly different base.
Reads are written straight into the reference's read-library format
(`<prefix>.lib_info` + `<prefix>.bin`, reference src/sequence/io/sequence_lib.cpp:84-90 and
src/sequence/sequence_package.h:224-240): per read `uint32 len` + ceil(len/16) uint32 words,
base j in bits 31-2j..30-2j of word j/16, forward orientation.
"""
import numpy as np


def pack_reads(bases):
    """bases: uint8 [n, L] in 0..3 -> uint32 [n, ceil(L/16)] MSB-first."""
    n, L = bases.shape
    nw = (L + 15) // 16
    pad = np.zeros((n, nw * 16), dtype=np.uint32)
    pad[:, :L] = bases
    pad = pad.reshape(n, nw, 16)
    shifts = (30 - 2 * np.arange(16, dtype=np.uint32)).astype(np.uint32)
    return (pad << shifts).sum(axis=2, dtype=np.uint64).astype(np.uint32)


def pack_reads_concat(bases):
    """bases: uint8 [n, L] -> uint32 words of the gap-free concatenation (SequencePackage layout)."""
    flat = np.ascontiguousarray(bases).reshape(-1)
    pad = (-flat.size) % 16
    if pad:
        flat = np.concatenate([flat, np.zeros(pad, dtype=np.uint8)])
    out = np.zeros(flat.size // 16, dtype=np.uint32)
    f = flat.reshape(-1, 16)
    for j in range(16):
        out |= f[:, j].astype(np.uint32) << np.uint32(30 - 2 * j)
    return out


def gen_pe_reads(n_pairs, genome_len, read_len=150, frag=400, err=0.005, seed=1, genome=None):
    """-> uint8 [2*n_pairs, read_len] (reads interleaved r1,r2,r1,r2...)."""
    rng = np.random.default_rng(seed)
    if genome is None:
        genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    start = rng.integers(0, genome_len - frag, size=n_pairs)
    idx = np.arange(read_len)
    r1 = genome[start[:, None] + idx[None, :]]
    r2 = 3 - genome[(start + frag - 1)[:, None] - idx[None, :]]
    reads = np.empty((2 * n_pairs, read_len), dtype=np.uint8)
    reads[0::2] = r1
    reads[1::2] = r2
    if err > 0:
        mask = rng.random(reads.shape) < err
        delta = rng.integers(1, 4, size=int(mask.sum()), dtype=np.uint8)
        reads[mask] = (reads[mask] + delta) & 3
    return reads


def write_read_lib(prefix, reads_list, description="synthetic", paired=True):
    """reads_list: list of uint8 arrays [n_i, L_i] (one block per fixed length) or list of 1-D arrays."""
    total_bases = 0
    total_reads = 0
    max_len = 0
    with open(prefix + ".bin", "wb") as f:
        for block in reads_list:
            if isinstance(block, np.ndarray) and block.ndim == 2:
                n, L = block.shape
                packed = pack_reads(block)
                rec = np.empty((n, 1 + packed.shape[1]), dtype=np.uint32)
                rec[:, 0] = L
                rec[:, 1:] = packed
                rec.tofile(f)
                total_bases += n * L
                total_reads += n
                max_len = max(max_len, L)
            else:
                for r in block:
                    r = np.asarray(r, dtype=np.uint8)
                    L = len(r)
                    np.array([L], dtype=np.uint32).tofile(f)
                    if L:
                        pack_reads(r[None, :])[0].tofile(f)
                    total_bases += L
                    total_reads += 1
                    max_len = max(max_len, L)
    with open(prefix + ".lib_info", "w") as f:
        f.write("%d %d\n%s\n0 %d %d %d\n" % (total_bases, total_reads, description, total_reads, max_len, int(paired)))
    return total_reads, total_bases


def write_fasta(path, reads):
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            f.write(">r%d\n%s\n" % (i, lut[np.asarray(r)].tobytes().decode()))


def write_fasta_interleaved(path, blocks):
    """blocks: uint8 [n, L] arrays (r1,r2,r1,r2,...) -> one interleaved FASTA, vectorised (headers all '>r')."""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "wb") as f:
        for reads in blocks:
            n, L = reads.shape
            rec = np.empty((n, L + 4), dtype=np.uint8)
            rec[:, 0] = ord(">")
            rec[:, 1] = ord("r")
            rec[:, 2] = ord("\n")
            rec[:, 3:3 + L] = lut[reads]
            rec[:, 3 + L] = ord("\n")
            rec.tofile(f)


def plant_repeats(genome, seed, n_families, min_len=25, max_len=140, max_copies=4):
    """Exact repeats planted in place: family f = a random sequence of min_len..max_len bases written at 2..max_copies
    random positions.  A repeat of length >= k forks the de Bruijn graph of order k and is resolved once k exceeds it, so
    an iterative k-list (21 ... 119) has work to do at every step (a uniform random genome assembles completely at k=21:
    the reference's orchestrator then stops after k=29)."""
    rng = np.random.default_rng(seed)
    G = genome.size
    lens = rng.integers(min_len, max_len + 1, size=n_families)
    copies = rng.integers(2, max_copies + 1, size=n_families)
    for L, c in zip(lens, copies):
        unit = rng.integers(0, 4, size=int(L), dtype=np.uint8)
        for pos in rng.integers(0, G - int(L), size=int(c)):
            genome[pos:pos + int(L)] = unit
    return genome


def gen_shard_library(n_reads, genome_seed, read_seed0, read_len=150, frag=400, err=0.005, repeat_families=0):
    """The single-genome workload family of SURVEY.md section 8d (configs 2-4): genome of 2.5 bp per read (~60x), PE blocks
    of 1 M pairs with seeds read_seed0 + i.  -> (genome, list of uint8 [n_i, read_len] blocks)."""
    G = int(n_reads * 2.5)
    genome = np.random.default_rng(genome_seed).integers(0, 4, size=G, dtype=np.uint8)
    if repeat_families:
        plant_repeats(genome, genome_seed + 7919, repeat_families)
    blocks = []
    for i, lo in enumerate(range(0, n_reads // 2, 1000000)):
        c = min(1000000, n_reads // 2 - lo)
        blocks.append(gen_pe_reads(c, G, read_len=read_len, frag=frag, err=err, seed=read_seed0 + i, genome=genome))
    return genome, blocks


def gen_metagenome_library(n_reads, n_genomes, genome_len=2500000, seed=3, sigma=1.0, read_len=150, frag=400, err=0.005):
    """SURVEY.md section 8d config 5 ("high diversity"): `n_genomes` i.i.d. genomes of `genome_len` bases with log-normal
    abundances (sigma), PE fragments never spanning two genomes.  Blocks of 1 M pairs (seeds 1000*seed + 1 + i)."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=n_genomes * genome_len, dtype=np.uint8)
    ab = rng.lognormal(0.0, sigma, size=n_genomes)
    cdf = np.cumsum(ab / ab.sum())
    blocks = []
    for i, lo in enumerate(range(0, n_reads // 2, 1000000)):
        c = min(1000000, n_reads // 2 - lo)
        r = np.random.default_rng(1000 * seed + 1 + i)
        g = np.minimum(np.searchsorted(cdf, r.random(c)), n_genomes - 1)
        start = g * genome_len + r.integers(0, genome_len - frag, size=c)
        idx = np.arange(read_len)
        r1 = genome[start[:, None] + idx[None, :]]
        r2 = 3 - genome[(start + frag - 1)[:, None] - idx[None, :]]
        reads = np.empty((2 * c, read_len), dtype=np.uint8)
        reads[0::2] = r1
        reads[1::2] = r2
        mask = r.random(reads.shape) < err
        delta = r.integers(1, 4, size=int(mask.sum()), dtype=np.uint8)
        reads[mask] = (reads[mask] + delta) & 3
        blocks.append(reads)
    return blocks


# ---- large libraries: blocks made by worker processes, consumed in order as they arrive (configs[2]: 100 M reads) ----
# Workers are SPAWNED (not forked): callers may have a HIP runtime with its threads alive in the parent.  Every worker makes the
# genome once for itself (default_rng(seed) is deterministic: 1.4 s for 250 Mbp).

_GENOMES = {}


def _genome(seed, G):
    g = _GENOMES.get((seed, G))
    if g is None:
        _GENOMES.clear()
        g = _GENOMES[(seed, G)] = np.random.default_rng(seed).integers(0, 4, size=G, dtype=np.uint8)
    return g


def _pe_block(job):
    """Worker: one PE block -> `.bin` record bytes ("records") or the packed words of the REVERSED reads ("reversed")."""
    form, c, G, genome_seed, read_len, frag, err, seed = job
    reads = gen_pe_reads(c, G, read_len=read_len, frag=frag, err=err, seed=seed, genome=_genome(genome_seed, G))
    if form == "reversed":
        return pack_reads_concat(reads[:, ::-1])
    packed = pack_reads(reads)
    rec = np.empty((reads.shape[0], 1 + packed.shape[1]), dtype=np.uint32)
    rec[:, 0] = read_len
    rec[:, 1:] = packed
    return rec.tobytes()


def map_pe_blocks(jobs, procs=None):
    """-> iterator over _pe_block(job) in job order, computed by up to `procs` spawned workers (1: in this process)."""
    import multiprocessing as mp
    import os
    import sys
    procs = max(1, min(procs or (os.cpu_count() or 1), len(jobs), 64))
    main = sys.modules.get("__main__")
    if getattr(main, "__file__", None) is None and (not sys.argv or sys.argv[0] in ("", "-")):
        procs = 1  # a program read from stdin: spawned workers cannot import their parent's __main__ (they hang)
    if procs == 1:
        for j in jobs:
            yield _pe_block(j)
        return
    with mp.get_context("spawn").Pool(procs) as pool:
        for out in pool.imap(_pe_block, jobs, chunksize=1):
            yield out


def pe_jobs(form, n_reads, G, genome_seed, read_seed0, read_len=150, frag=400, err=0.005):
    return [(form, min(1000000, n_reads // 2 - lo), G, genome_seed, read_len, frag, err, read_seed0 + i)
            for i, lo in enumerate(range(0, n_reads // 2, 1000000))]


def write_pe_library(prefix, n_reads, genome_seed, read_seed0, procs=None, read_len=150, frag=400, err=0.005, description="synthetic"):
    """The single-genome family of SURVEY.md section 8d at any size, without holding the reads in memory: genome =
    default_rng(genome_seed) over 2.5 bp per read, PE blocks of 1 M pairs with seeds read_seed0 + i — byte for byte the
    file write_read_lib(gen_shard_library(...)) writes."""
    n_reads = n_reads // 2 * 2
    G = int(n_reads * 2.5)
    with open(prefix + ".bin", "wb") as f:
        for blob in map_pe_blocks(pe_jobs("records", n_reads, G, genome_seed, read_seed0, read_len, frag, err), procs):
            f.write(blob)
    with open(prefix + ".lib_info", "w") as f:
        f.write("%d %d\n%s\n0 %d %d %d\n" % (n_reads * read_len, n_reads, description, n_reads, read_len, 1))
    return n_reads, n_reads * read_len


# ---- libraries whose reads are NOT of one length (trimmed reads: every real library after quality / N trimming) ----
VARLEN_RULES = ("u100_150", "trim2pct")


def trimmed_lengths(n, read_len, rule, seed):
    """lengths of n reads of `read_len` bases after trimming: "u100_150" = every read cut to U[100, read_len];
    "trim2pct" = 2 % of the reads cut to U[30, read_len), the others whole (N-trimmed reads of a real library)"""
    rng = np.random.default_rng(seed)
    if rule == "u100_150":
        return rng.integers(min(100, read_len), read_len + 1, size=n).astype(np.uint32)
    if rule == "trim2pct":
        lens = np.full(n, read_len, dtype=np.uint32)
        cut = rng.random(n) < 0.02
        lens[cut] = rng.integers(min(30, read_len), read_len, size=int(cut.sum()))
        return lens
    raise ValueError(rule)


def varlen_blocks(n_reads, rule, genome_seed=1, read_seed0=1001, read_len=150, frag=400, err=0.005):
    """the library of write_pe_library / bench.py (same genome, same PE blocks of 1 M pairs) with every block's reads trimmed by `rule`
    (length seed = the block's read seed + 500000): -> iterator over (bases uint8 [n, read_len], lens uint32 [n])"""
    n_reads = n_reads // 2 * 2
    G = max(5000, int(n_reads * 2.5))
    genome = _genome(genome_seed, G)
    for i, lo in enumerate(range(0, n_reads // 2, 1000000)):
        c = min(1000000, n_reads // 2 - lo)
        reads = gen_pe_reads(c, G, read_len=read_len, frag=frag, err=err, seed=read_seed0 + i, genome=genome)
        yield reads, trimmed_lengths(reads.shape[0], read_len, rule, read_seed0 + i + 500000)


def pack_var_reversed(reads, lens):
    """reads uint8 [n, L], lens [n] -> (uint32 words of the gap-free concatenation of the REVERSED trimmed reads, as the reference's
    SequencePackage holds a library it loaded; base count) — vectorised"""
    n, L = reads.shape
    rev = reads[:, ::-1]
    keep = np.arange(L, dtype=np.uint32)[None, :] >= (L - lens)[:, None]  # the trimmed read = the first lens bases = the LAST lens of the reversal
    flat = rev[keep]
    n_bases = int(flat.size)
    pad = (-flat.size) % 16
    if pad:
        flat = np.concatenate([flat, np.zeros(pad, dtype=np.uint8)])
    out = np.zeros(flat.size // 16, dtype=np.uint32)
    f = flat.reshape(-1, 16)
    for j in range(16):
        out |= f[:, j].astype(np.uint32) << np.uint32(30 - 2 * j)
    return out, n_bases


def write_var_read_lib(prefix, blocks, description="synthetic, trimmed", paired=True):
    """blocks: iterable of (bases [n, L], lens [n]) -> `<prefix>.bin` / `.lib_info` in the reference's format (per read: uint32 length +
    ceil(length / 16) words, bits past the length zero) — vectorised"""
    total_bases = total_reads = max_len = 0
    with open(prefix + ".bin", "wb") as f:
        for reads, lens in blocks:
            n, L = reads.shape
            words = pack_reads(reads)  # [n, ceil(L / 16)]
            nw = words.shape[1]
            need = (lens + 15) // 16
            # zero the bits past the read's end in its last word
            last = np.maximum(need, 1) - 1
            rem = (lens % 16).astype(np.uint32)
            mask_last = np.where(rem == 0, np.uint32(0xFFFFFFFF), (np.uint32(0xFFFFFFFF) << (np.uint32(32) - 2 * rem)).astype(np.uint32))
            words[np.arange(n), last] &= mask_last
            rec = np.empty((n, 1 + nw), dtype=np.uint32)
            rec[:, 0] = lens
            rec[:, 1:] = words
            keep = np.arange(1 + nw, dtype=np.uint32)[None, :] <= need[:, None]
            rec[keep].tofile(f)
            total_bases += int(lens.sum())
            total_reads += n
            max_len = max(max_len, int(lens.max()) if n else 0)
    with open(prefix + ".lib_info", "w") as f:
        f.write("%d %d\n%s\n0 %d %d %d\n" % (total_bases, total_reads, description, total_reads, max_len, int(paired)))
    return total_reads, total_bases
