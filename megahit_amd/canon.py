"""Canonical (bucket-ordered) streams of the on-disk formats of the SdBG-construction path.

Per-file bytes of the reference are not deterministic (bucket -> file assignment follows OpenMP
dynamic scheduling, reference src/sorting/base_engine.cpp:323) but the bucket-ordered logical
stream is.  These helpers turn `<p>.edges.*` / `<p>.sdbg.*` of ANY producer
(reference, oracle, this framework) into that canonical stream so they can be compared.

Formats: reference src/sequence/io/edge/edge_io_meta.h:25-66, src/sdbg/sdbg_meta.cpp:51-75.
"""
import hashlib
import os

import numpy as np

NULL_ID = 18446744073709551615


def read_edges_info(prefix):
    with open(prefix + ".edges.info") as f:
        toks = f.read().split()
    hdr = {}
    for i in range(6):
        hdr[toks[2 * i]] = int(toks[2 * i + 1])
    rest = np.array(toks[12:], dtype=np.int64).reshape(-1, 4)
    return hdr, rest


def canonical_edges(prefix):
    """-> (header dict, uint32 array [n_edges, words_per_edge] in bucket-id order, per-bucket counts)."""
    hdr, rows = read_edges_info(prefix)
    wpe = hdr["words_per_edge"]
    files = [np.fromfile("%s.edges.%d" % (prefix, i), dtype=np.uint32) for i in range(hdr["num_files"])]
    if not hdr["is_sorted"]:
        return hdr, files[0][: hdr["num_edges"] * wpe].reshape(-1, wpe), None
    parts = []
    counts = np.zeros(hdr["num_buckets"], dtype=np.int64)
    for bid, fid, off, cnt in rows:
        if fid < 0:
            continue
        parts.append(files[fid][off * wpe:(off + cnt) * wpe])
        counts[bid] = cnt
    data = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)
    return hdr, data.reshape(-1, wpe), counts


def read_sdbg_info(prefix):
    with open(prefix + ".sdbg_info") as f:
        toks = f.read().split()
    hdr = {toks[0]: int(toks[1]), toks[2]: int(toks[3]), toks[4]: int(toks[5]), toks[6]: int(toks[7])}
    rows = [tuple(int(x) for x in toks[8 + 6 * i: 14 + 6 * i]) for i in range(hdr["num_buckets"])]
    return hdr, rows


def sdbg_bucket_nbytes(n_items, n_tips, n_large, words_per_tip_label):
    return 2 * n_items + 2 * n_large + 4 * words_per_tip_label * n_tips


def canonical_sdbg(prefix):
    """-> (header, list of (bucket_id, num_items, num_tips, num_large_mul, bytes) for ascending bucket id)."""
    hdr, rows = read_sdbg_info(prefix)
    wpt = hdr["words_per_tip_label"]
    files = {}
    out = []
    for bid, fid, off, n_items, n_tips, n_large in rows:
        if bid == NULL_ID or fid == NULL_ID:
            continue
        if fid not in files:
            files[fid] = np.fromfile("%s.sdbg.%d" % (prefix, fid), dtype=np.uint8)
        nb = sdbg_bucket_nbytes(n_items, n_tips, n_large, wpt)
        out.append((bid, n_items, n_tips, n_large, files[fid][off:off + nb].tobytes()))
    out.sort(key=lambda r: r[0])
    return hdr, out


def digest_edges(prefix):
    hdr, data, counts = canonical_edges(prefix)
    h = hashlib.md5()
    h.update(("k%d w%d n%d|" % (hdr["kmer_size"], hdr["words_per_edge"], data.shape[0])).encode())
    if counts is not None:
        h.update(counts.tobytes())
    h.update(np.ascontiguousarray(data).tobytes())
    return h.hexdigest()


def digest_sdbg(prefix, bucket_lo=0, bucket_hi=None):
    """md5 over the SdBG's buckets in ascending id order; with bucket_lo/bucket_hi only the buckets in [lo, hi)."""
    hdr, buckets = canonical_sdbg(prefix)
    h = hashlib.md5()
    h.update(("k%d w%d|" % (hdr["k"], hdr["words_per_tip_label"])).encode())
    for bid, ni, nt, nl, b in buckets:
        if bid < bucket_lo or (bucket_hi is not None and bid >= bucket_hi):
            continue
        h.update(np.array([bid, ni, nt, nl], dtype=np.uint64).tobytes())
        h.update(b)
    return h.hexdigest()


def digest_sdbg_buffers(k, sdbg_bytes, bucket_items, bucket_tips, bucket_large, bucket_offset):
    """The digest of digest_sdbg() computed from the library's result buffers (MHX_BUF_SDBG_BYTES + the per-bucket
    tables) instead of from files: lets a resident-in-HBM run be compared with a reference run's files."""
    wpt = (k + 15) // 16
    h = hashlib.md5()
    h.update(("k%d w%d|" % (k, wpt)).encode())
    items = np.asarray(bucket_items, dtype=np.uint64)
    tips = np.asarray(bucket_tips, dtype=np.uint64)
    large = np.asarray(bucket_large, dtype=np.uint64)
    off = np.asarray(bucket_offset, dtype=np.uint64)
    buf = memoryview(np.ascontiguousarray(sdbg_bytes, dtype=np.uint8))
    for bid in np.nonzero(items)[0]:
        ni, nt, nl = int(items[bid]), int(tips[bid]), int(large[bid])
        h.update(np.array([bid, ni, nt, nl], dtype=np.uint64).tobytes())
        o = int(off[bid])
        h.update(buf[o:o + sdbg_bucket_nbytes(ni, nt, nl, wpt)])
    return h.hexdigest()


def digest_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def sorted_mercy_cand(prefix):
    """All <prefix>.mercy_cand.* records, sorted (order inside the files is nondeterministic)."""
    parts = []
    i = 0
    while os.path.exists("%s.mercy_cand.%d" % (prefix, i)):
        parts.append(np.fromfile("%s.mercy_cand.%d" % (prefix, i), dtype=np.int64))
        i += 1
    if not parts:
        return np.zeros(0, dtype=np.int64)
    return np.sort(np.concatenate(parts))
