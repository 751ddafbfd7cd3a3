"""Memory-bounded runs: the reference's lv1 passes (reference src/sorting/base_engine.cpp:54-141,213-281) on top of
mhx_bucket_histogram / mhx_set_bucket_filter.  One call per contiguous range of lv1 buckets whose items fit the
given budget; per-range outputs (edges, SdBG records, per-bucket tables) are concatenated in bucket order, which is
the canonical order of the single-pass result.  `engine` is a megahit_amd.lib.Engine with the sequences loaded.
"""
import numpy as np

from . import lib

NUM_BUCKETS = 65536


def plan_ranges(hist, max_items):
    """Contiguous bucket ranges [(lo, hi, n_items)] with n_items <= max_items (a single bucket above the budget gets a
    range of its own: the reference aborts there, base_engine.cpp:96-99; here it is the caller's call).
    max_items < 0 means "about -max_items equal passes" (tests)."""
    hist = np.asarray(hist, dtype=np.uint64)
    if max_items < 0:
        max_items = int(hist.sum()) // (-max_items) + 1
    out, lo, acc = [], 0, 0
    for b in range(NUM_BUCKETS):
        h = int(hist[b])
        if acc and acc + h > max_items:
            out.append((lo, b, acc))
            lo, acc = b, 0
        acc += h
    out.append((lo, NUM_BUCKETS, acc))
    return out


def _mask(lo, hi):
    m = np.zeros(NUM_BUCKETS, dtype=np.uint8)
    m[lo:hi] = 1
    return m


def items_budget(free_bytes, item_bytes):
    """items per pass for a given amount of free HBM: two ping-pong buffers + the filtered copy + slack"""
    return max(1, int(free_bytes * 0.8) // (3 * item_bytes))


def count_in_passes(e, k, m, max_items, batch_bytes=0):
    ranges = plan_ranges(e.bucket_histogram(lib.STAGE_COUNT, k, m), max_items)
    edges, bcount = [], np.zeros(NUM_BUCKETS, dtype=np.uint64)
    try:
        for i, (lo, hi, n) in enumerate(ranges):
            e.set_bucket_filter(_mask(lo, hi), n, batch_bytes, accumulate=i > 0)
            r = e.count(k, m)
            edges.append(e.fetch(lib.BUF_EDGES, np.uint32).reshape(-1, r.words_per_edge))
            bcount += e.fetch(lib.BUF_BUCKET_COUNT, np.uint64)
    finally:
        e.set_bucket_filter(None)
    return dict(edges=np.concatenate(edges), bucket_count=bcount, hist=e.fetch(lib.BUF_MUL_HIST, np.int64),
                first_0_out=e.fetch(lib.BUF_FIRST_0_OUT, np.uint32), last_0_in=e.fetch(lib.BUF_LAST_0_IN, np.uint32),
                n_passes=len(ranges))


def _collect_sdbg(e, r, acc):
    byts = e.fetch(lib.BUF_SDBG_BYTES, np.uint8)
    items = e.fetch(lib.BUF_BUCKET_COUNT, np.uint64)
    off = e.fetch(lib.BUF_BUCKET_OFFSET, np.uint64)
    nz = items > 0
    acc["bucket_off"][nz] = off[nz] + np.uint64(acc["n_bytes"])
    acc["bytes"].append(byts)
    acc["n_bytes"] += byts.size
    acc["bucket_items"] += items
    acc["bucket_tips"] += e.fetch(lib.BUF_BUCKET_TIPS, np.uint64)
    acc["bucket_large"] += e.fetch(lib.BUF_BUCKET_LARGE, np.uint64)
    acc["w_count"] += e.fetch(lib.BUF_W_COUNT, np.uint64)[:10]
    acc["wpt"] = r.words_per_tip_label


def _new_sdbg():
    z = lambda: np.zeros(NUM_BUCKETS, dtype=np.uint64)
    return dict(bytes=[], n_bytes=0, bucket_off=z(), bucket_items=z(), bucket_tips=z(), bucket_large=z(),
                w_count=np.zeros(10, dtype=np.uint64), wpt=0)


def _finish_sdbg(acc, n_passes):
    acc["bytes"] = np.concatenate(acc["bytes"]) if acc["bytes"] else np.zeros(0, dtype=np.uint8)
    acc["n_passes"] = n_passes
    return acc


def read2sdbg_in_passes(e, k, m, max_items_s1, max_items_s2, need_mercy=0, batch_bytes=0):
    """-> (stage-1 summary, SdBG dict).  Stage 1 runs over its own ranges and accumulates is_solid / histogram /
    mercy candidates / aggregated items; the optional mercy block follows; stage 2 runs over ranges of ITS buckets."""
    s1 = None
    try:
        if m > 1:
            stage = lib.STAGE_S1_MERCY if need_mercy else lib.STAGE_S1
            ranges = plan_ranges(e.bucket_histogram(stage, k, m), max_items_s1)
            n_items = 0
            for i, (lo, hi, n) in enumerate(ranges):
                e.set_bucket_filter(_mask(lo, hi), n, batch_bytes, accumulate=i > 0)
                r = e.read2sdbg_s1(k, m, want_mercy=need_mercy)
                n_items += r.n_items
            e.set_bucket_filter(None)
            s1 = dict(n_items=n_items, n_solid=r.n_solid, n_mercy_cand=r.n_mercy_cand, n_passes=len(ranges))
            if need_mercy:
                s1["n_mercy"] = e.read2sdbg_add_mercy(k)
        ranges = plan_ranges(e.bucket_histogram(lib.STAGE_S2, k, m), max_items_s2)
        acc = _new_sdbg()
        for lo, hi, n in ranges:
            e.set_bucket_filter(_mask(lo, hi), n, batch_bytes)
            _collect_sdbg(e, e.read2sdbg_s2(k, m), acc)
    finally:
        e.set_bucket_filter(None)
    return s1, _finish_sdbg(acc, len(ranges))


def seq2sdbg_in_passes(e, k, max_items, batch_bytes=0):
    ranges = plan_ranges(e.bucket_histogram(lib.STAGE_SEQ2SDBG, k, 0), max_items)
    acc = _new_sdbg()
    try:
        for lo, hi, n in ranges:
            e.set_bucket_filter(_mask(lo, hi), n, batch_bytes)
            _collect_sdbg(e, e.seq2sdbg(k), acc)
    finally:
        e.set_bucket_filter(None)
    return _finish_sdbg(acc, len(ranges))
