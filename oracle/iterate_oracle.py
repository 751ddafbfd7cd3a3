"""TEST INFRASTRUCTURE — not product code.  A plain-Python restatement of the reference's `iterate`:
the (k+step+1)-mer edges that reads support between the contigs of round k.  Pinned against the reference's own
`megahit_core iterate` (oracle/_ref/ref_megahit_core) by tests/test_oracle_iterate.py.  Only tests import this.

Follows, with the reference's line numbers (all under /root/reference/src):
  iterate/contig_flank_index.h:29-88    FeedBatchContigs: per contig and strand, the first (k+1)-mer + up to step-1 bases
                                        after it; of two flanks with one (k+1)-mer the longer, then the larger, extension wins
  iterate/contig_flank_index.h:90-213   FindNextKmersFromReads: flank hits in either orientation and their matching
                                        extensions set `exist` bits; every position that closes a run of step+1 set bits yields
                                        the canonical (k+step+1)-mer ending there
  iterate/kmer_collector.h:49-69        the record: the chosen k-mer with base k-1 FIRST (WriteToFile reads GetBase(k-1-j)),
                                        2 bits per base MSB-first, multiplicity 0 in the low 16 bits of the last word
  sequence/io/async_sequence_reader.h:80  loop and standalone contigs are not fed
Sequences are lists of ints 0..3; k-mers are Python ints with base 0 most significant (numeric order = Kmer::operator<)."""
import numpy as np

K_STANDALONE, K_LOOP = 1, 2


def read_fasta_contigs(path, discard_flags=K_STANDALONE | K_LOOP):
    """-> list of base lists; header `>name flag=F multi=M len=L` (contig_reader.h:60-70)"""
    out, name, seq = [], None, []
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}

    def flush():
        if name is None:
            return
        comment = name.split(" ", 1)[1] if " " in name else ""
        flag = int(comment[5]) if len(comment) > 5 else 0
        if not (flag & discard_flags):
            out.append([code.get(c, 2) for c in "".join(seq).encode()])
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                flush()
                name, seq = line[1:], []
            elif name is not None:
                seq.append(line)
    flush()
    return out


def read_bin_reads(path):
    """.bin records: uint32 length + ceil(len/16) words, base j in bits 31-2j..30-2j of word j/16 (sequence_package.h:224-240)"""
    w = np.fromfile(path, dtype=np.uint32)
    out, i = [], 0
    while i < w.size:
        n = int(w[i])
        nw = (n + 15) // 16
        words = w[i + 1:i + 1 + nw]
        out.append([int((int(words[j >> 4]) >> (30 - 2 * (j & 15))) & 3) for j in range(n)])
        i += 1 + nw
    return out


def to_int(bases):
    v = 0
    for b in bases:
        v = (v << 2) | b
    return v


def rc_int(v, n):
    r = 0
    for _ in range(n):
        r = (r << 2) | (3 - (v & 3))
        v >>= 2
    return r


def flank_index(contig_sets, k, step):
    index = {}
    for contigs in contig_sets:
        for seq in contigs:
            L = len(seq)
            if L < k + 1:
                continue
            for strand in (0, 1):
                def ch(j):
                    return seq[j] if strand == 0 else 3 - seq[L - 1 - j]
                kmer = to_int([ch(j) for j in range(k + 1)])
                if (k + 1) % 2 == 0 and rc_int(kmer, k + 1) == kmer:  # Kmer::IsPalindrome, kmer.h:168-172
                    continue
                ext_len = min(step - 1, L - (k + 1))
                ext_seq = 0
                for j in range(ext_len):
                    ext_seq |= ch(k + 1 + j) << (2 * j)
                old = index.get(kmer)
                if old is None or old[0] < ext_len or (old[0] == ext_len and old[1] < ext_seq):
                    index[kmer] = (ext_len, ext_seq)
                if L == k + 1:
                    break
    return index


def next_kmers(reads, index, k, step):
    """-> (set of canonical (k+step+1)-mers, number of reads that yielded one)"""
    nk = k + step + 1
    found, aligned = set(), 0
    for seq in reads:
        L = len(seq)
        if L < nk:
            continue
        exist = [False] * L
        cur = 0
        while cur + k + 1 <= L:
            nxt = cur + 1
            if not exist[cur]:
                kmer = to_int(seq[cur:cur + k + 1])
                hit = index.get(kmer)
                if hit is not None:
                    exist[cur] = True
                    ext_len, ext_seq = hit
                    j = 0
                    while j < ext_len and cur + k + 1 + j < L:
                        if seq[cur + k + 1 + j] == (ext_seq >> (2 * j)) & 3:
                            exist[cur + j + 1] = True
                        else:
                            break
                        j += 1
                        nxt += 1
                hit = index.get(rc_int(kmer, k + 1))
                if hit is not None:
                    exist[cur] = True
                    ext_len, ext_seq = hit
                    j = 0
                    while j < ext_len and cur >= j + 1:
                        if 3 - seq[cur - 1 - j] == (ext_seq >> (2 * j)) & 3:
                            exist[cur - 1 - j] = True
                        else:
                            break
                        j += 1
            if nxt + k + 1 <= L:
                cur = nxt
            else:
                break
        acc, any_new = 0, False
        for j in range(L - k):
            acc = acc + 1 if exist[j] else 0
            if acc >= step + 1:
                v = to_int(seq[j - step:j + k + 1])
                r = rc_int(v, nk)
                found.add(min(v, r))
                any_new = True
        aligned += any_new
    return found, aligned


def edge_records(kmers, nk):
    """sorted uint32 [n, words_per_edge]: each k-mer reversed (base nk-1 first), zero padded, multiplicity 0"""
    wpe = (2 * nk + 16 + 31) // 32
    rows = np.zeros((len(kmers), wpe), dtype=np.uint32)
    for i, v in enumerate(kmers):
        bases = [(v >> (2 * (nk - 1 - j))) & 3 for j in range(nk)][::-1]
        for j, b in enumerate(bases):
            rows[i, j >> 4] |= np.uint32(b << (30 - 2 * (j & 15)))
    if rows.shape[0]:
        rows = rows[np.lexsort(rows.T[::-1])]
    return rows


def iterate(contig_file, bubble_file, reads_bin, k, step):
    """-> (sorted edge records, words_per_edge, number of flank k-mers, number of aligned reads)"""
    index = flank_index([read_fasta_contigs(contig_file), read_fasta_contigs(bubble_file)], k, step)
    kmers, aligned = next_kmers(read_bin_reads(reads_bin), index, k, step)
    rows = edge_records(sorted(kmers), k + step + 1)
    return rows, rows.shape[1], len(index), aligned
