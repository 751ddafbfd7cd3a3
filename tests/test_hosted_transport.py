"""CPU, world_size 2 and 3 on gloo: the byte movers behind libmhx's hosted communicator (megahit_amd/hosted.py — what the
multi-GPU drivers of comm.hip call back into when the ranks are processes without a shared RCCL world).  No libmhx compute
here (that needs a GPU: tests/test_gpu_multiprocess.py runs the same callbacks under mhx_dist_*); this checks the part that
runs anywhere: the in-place 64-bit reductions (including the "~0 - x" values libmhx max-reduces to find a minimum) and the
variable-size all-to-all built from point-to-point pairs, with empty and uneven segments."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _segments(src, dst, world):
    """what rank src sends to rank dst: a seeded, uneven, sometimes empty byte string"""
    rng = np.random.default_rng(1000 * src + dst)
    n = 0 if (src + 2 * dst) % 5 == 0 else int(rng.integers(1, 200000))
    return rng.integers(0, 256, size=n, dtype=np.uint8)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from megahit_amd import hosted
    t, _keep = hosted.make_transport(dist, rank, world)
    ok = True
    # reductions, through the C-callable function pointers exactly as libmhx calls them
    v = np.array([rank + 1, 10 * (rank + 1), (1 << 64) - 1 - 1000 * (rank + 1)], dtype=np.uint64)
    s = v.copy()
    assert t.all_reduce_u64(None, s.ctypes.data_as(C.c_void_p), s.size, 0) == 0
    want_sum = sum(np.array([r + 1, 10 * (r + 1), (1 << 64) - 1 - 1000 * (r + 1)], dtype=np.uint64) for r in range(world))  # wraps like uint64
    ok = ok and np.array_equal(s, want_sum)
    m = v.copy()
    assert t.all_reduce_u64(None, m.ctypes.data_as(C.c_void_p), m.size, 1) == 0
    # max: per element all values share a sign when read as int64, so the signed max is the unsigned max
    ok = ok and m[0] == world and m[1] == 10 * world and m[2] == (1 << 64) - 1 - 1000  # = ~0 - min(x): how libmhx finds the least free memory
    # all-to-all
    send = [(_segments(rank, p, world) if p != rank else np.zeros(0, dtype=np.uint8)) for p in range(world)]
    recv_n = [(_segments(p, rank, world).size if p != rank else 0) for p in range(world)]
    sb = (C.c_uint64 * world)(*[x.size for x in send])
    rb = (C.c_uint64 * world)(*recv_n)
    sbuf = np.concatenate(send) if sum(x.size for x in send) else np.zeros(1, dtype=np.uint8)
    rbuf = np.zeros(max(1, sum(recv_n)), dtype=np.uint8)
    assert t.all_to_all_bytes(None, sbuf.ctypes.data_as(C.c_void_p), sb, rbuf.ctypes.data_as(C.c_void_p), rb) == 0
    at = 0
    for p in range(world):
        if p != rank:
            ok = ok and np.array_equal(rbuf[at:at + recv_n[p]], _segments(p, rank, world))
        at += recv_n[p]
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_hosted_transport_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs == [(r, True) for r in range(world)]
