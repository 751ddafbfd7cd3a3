"""GPU probe: the multi-GPU code path of ONE rank at full size, with read positions beyond 2^32 (what every rank of a
>= 3-GPU run sees: 16-byte stage-1 items with 64-bit positions, global bitmap, slice adoption), checked against the
single-GPU result of the same reads.  No process group: the all-to-all of a 1-rank world is a device copy.

    python tools/probe_bigpos.py [reads]
"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from megahit_amd import dist as mdist
from megahit_amd import lib

T0 = time.perf_counter()


def note(msg):
    print("[bigpos %.1f s] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


n_reads = (int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000) // 16 * 16
K, M = bench.K, bench.MIN_COUNT
packed = bench.make_reads(n_reads, 0, 1)
dev = torch.device("cuda", 0)
e = lib.Engine(0)
e.load_sequences(packed, n_reads, bench.READ_LEN, None)
note("loaded")
e.read2sdbg_s1(K, M)
r2 = e.read2sdbg_s2(K, M)
want = dict(md5=hashlib.md5(e.fetch(lib.BUF_SDBG_BYTES, np.uint8).tobytes()).hexdigest(), items=e.fetch(lib.BUF_BUCKET_COUNT, np.uint64),
            n=int(r2.n_sdbg), hist=e.fetch(lib.BUF_MUL_HIST, np.int64))
note("single-GPU reference done: %d records" % want["n"])

stride_words = (n_reads * bench.READ_LEN + 63) // 64
pos_base = 5 * (1 << 30) // 64 * 64           # > 2^32
global_bases = pos_base + 2 * stride_words * 64
e.set_partition(0, 1, np.array([0, 65536], dtype=np.uint32))
e.set_global_layout(pos_base, global_bases)


def move(stage):
    ptr, n, ib, counts = e.dist_extract(stage, K, M)
    assert int(counts[0]) == int(n)
    rptr = e.dist_recv_buffer(int(n), ib)
    e.as_tensor(rptr, int(n) * ib, dev).copy_(e.as_tensor(ptr, int(n) * ib, dev))
    torch.cuda.synchronize()
    return int(n), ib


e.profile(True)
e.profile_reset()
t0 = time.perf_counter()
n1, ib1 = move(mdist.STAGE_S1)
note("S1 items %d x %d B" % (n1, ib1))
r1 = e.dist_process_s1(K, M, n1)
note("S1 processed; kernels ms: %s" % {k2: round(v["ms"], 1) for k2, v in e.profile_get().items() if v["ms"] > 1})
bm = e.as_tensor(e.device_pointer(lib.BUF_IS_SOLID), (global_bases // 64) * 8, dev).view(torch.int64)
sl = bm[pos_base // 64: pos_base // 64 + stride_words].clone()
assert int(bm[: pos_base // 64].count_nonzero()) == 0
e.adopt_is_solid_slice(sl.data_ptr(), stride_words)
e.profile_reset()
n2, ib2 = move(mdist.STAGE_S2)
r2 = e.dist_process_s2(K, n2)
dt = time.perf_counter() - t0
note("S2 processed; kernels ms: %s" % {k2: round(v["ms"], 1) for k2, v in e.profile_get().items() if v["ms"] > 1})
got = dict(md5=hashlib.md5(e.fetch(lib.BUF_SDBG_BYTES, np.uint8).tobytes()).hexdigest(), items=e.fetch(lib.BUF_BUCKET_COUNT, np.uint64),
           n=int(r2.n_sdbg), hist=e.fetch(lib.BUF_MUL_HIST, np.int64))
ok = got["md5"] == want["md5"] and got["n"] == want["n"] and np.array_equal(got["items"], want["items"]) and np.array_equal(got["hist"], want["hist"])
print(json.dumps({"reads": n_reads, "pos_base": pos_base, "s1_item_bytes": ib1, "s2_item_bytes": ib2, "s1_items": n1, "s2_items": n2,
                  "seconds_one_rank_step": round(dt, 3), "sdbg_md5_equal": got["md5"] == want["md5"], "all_ok": bool(ok)}))
sys.exit(0 if ok else 1)
