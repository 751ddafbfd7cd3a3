"""GPU: the multi-GPU path with real kernels.  A 1-GPU box cannot host two RCCL ranks (RCCL refuses
duplicate devices), so two processes share cuda:0, the process group is gloo and item buffers are
staged through host memory; extraction, owner partition, sort, reduction and SdBG emission are the
same HIP kernels the N-GPU bench runs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu


def _reads(seed):
    from test_dist_cpu import _reads as r
    return r(seed, n_pairs=600)


def _worker(rank, world, port, k, m, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from megahit_amd import dist as mdist
    from megahit_amd import lib
    pkg = ob.Package(_reads(100 + rank), reverse=True)
    eng = lib.Engine(0)
    eng.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
    runner = mdist.DistRead2Sdbg(eng, k, m, rank, world, torch.device("cuda", 0), staging="host")
    r1, r2 = runner.step()
    r1, r2 = runner.step()  # a second step must give the same answer (buffers are reused)
    lo, hi = int(runner.bucket_begin[rank]), int(runner.bucket_begin[rank + 1])
    q.put((rank, lo, hi, eng.fetch(lib.BUF_SDBG_BYTES, np.uint8).tobytes(), eng.fetch(lib.BUF_BUCKET_COUNT, np.uint64),
           eng.fetch(lib.BUF_BUCKET_TIPS, np.uint64), eng.fetch(lib.BUF_BUCKET_LARGE, np.uint64),
           eng.fetch(lib.BUF_MUL_HIST, np.int64) if m > 1 else None))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


@pytest.mark.parametrize("world,k,m", [(2, 21, 2), (2, 21, 1), (1, 27, 2), (3, 31, 2)])
def test_ranks_on_one_gpu_equal_single_process(world, k, m):
    import oracle_binding as ob
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, m, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    allreads = []
    for r in range(world):
        allreads += _reads(100 + r)
    pkg = ob.Package(allreads, reverse=True)
    if m > 1:
        s1 = ob.s1(pkg, k, m)
        want = ob.s2(pkg, k, m, s1["is_solid"])
        assert np.array_equal(sum(o[7] for o in outs), s1["hist"])
    else:
        want = ob.s2(pkg, k, 1, None)
    off = np.concatenate([want["bucket_off"], [len(want["bytes"])]]).astype(np.int64)
    for rank, lo, hi, byts, b_items, b_tips, b_large, _ in outs:
        assert np.array_equal(b_items[lo:hi], want["bucket_items"][lo:hi])
        assert np.array_equal(b_tips[lo:hi], want["bucket_tips"][lo:hi])
        assert np.array_equal(b_large[lo:hi], want["bucket_large"][lo:hi])
        assert b_items[:lo].sum() == 0 and b_items[hi:].sum() == 0
        assert byts == want["bytes"][off[lo]:off[hi]].tobytes()


def _worker2(rank, world, port, mode, k, m, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from megahit_amd import dist as mdist
    from megahit_amd import lib
    from test_dist_cpu import _seqs_with_mult
    dev = torch.device("cuda", 0)
    eng = lib.Engine(0)
    if mode == "seq2sdbg":
        seqs, mult = _seqs_with_mult(50 + rank)
        pkg = ob.Package(seqs, reverse=False)
        eng.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
        eng.load_multiplicity(mult)
        runner = mdist.DistSeq2Sdbg(eng, k, rank, world, dev, staging="host")
    else:
        pkg = ob.Package(_reads(100 + rank), reverse=True)
        eng.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
        got = dict(bytes=[], items=0, tips=0, large=0)

        def collect(p, r2):
            got["bytes"].append(eng.fetch(lib.BUF_SDBG_BYTES, np.uint8).tobytes())
            got["items"] = got["items"] + eng.fetch(lib.BUF_BUCKET_COUNT, np.uint64)
            got["tips"] = got["tips"] + eng.fetch(lib.BUF_BUCKET_TIPS, np.uint64)
            got["large"] = got["large"] + eng.fetch(lib.BUF_BUCKET_LARGE, np.uint64)

        if mode == "count":
            runner = mdist.DistCount(eng, k, m, rank, world, dev, staging="host")
        elif mode.startswith("passes"):
            runner = mdist.DistRead2Sdbg(eng, k, m, rank, world, dev, staging="host", need_mercy=1 if mode == "passes_mercy" else 0,
                                         n_passes=3, batch_bytes=8 << 20, on_s2_pass=collect)
            runner.step()
            lo, hi = int(runner.bucket_begin[rank]), int(runner.bucket_begin[rank + 1])
            q.put((rank, lo, hi, b"".join(got["bytes"]), got["items"], got["tips"], got["large"], runner.n_mercy))
            dist.barrier()
            dist.destroy_process_group()
            eng.close()
            return
        else:
            runner = mdist.DistRead2Sdbg(eng, k, m, rank, world, dev, staging="host", need_mercy=2 if mode == "mercy_exact" else 1)
    runner.step()
    runner.step()  # buffers are reused: same answer the second time
    lo, hi = int(runner.bucket_begin[rank]), int(runner.bucket_begin[rank + 1])
    if mode == "count":
        r = eng.fetch(lib.BUF_EDGES, np.uint32)
        q.put((rank, r, eng.fetch(lib.BUF_BUCKET_COUNT, np.uint64), eng.fetch(lib.BUF_MUL_HIST, np.int64),
               eng.fetch(lib.BUF_FIRST_0_OUT, np.uint32), eng.fetch(lib.BUF_LAST_0_IN, np.uint32)))
    else:
        q.put((rank, lo, hi, eng.fetch(lib.BUF_SDBG_BYTES, np.uint8).tobytes(), eng.fetch(lib.BUF_BUCKET_COUNT, np.uint64),
               eng.fetch(lib.BUF_BUCKET_TIPS, np.uint64), eng.fetch(lib.BUF_BUCKET_LARGE, np.uint64),
               getattr(runner, "n_mercy", 0)))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def _run2(mode, k, m, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker2, args=(r, world, port, mode, k, m, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return outs


@pytest.mark.parametrize("world,k,m", [(2, 21, 2), (3, 31, 3)])
def test_count_ranks_on_one_gpu(world, k, m):
    import oracle_binding as ob
    outs = _run2("count", k, m, world)
    allreads = []
    for r in range(world):
        allreads += _reads(100 + r)
    want = ob.count(ob.Package(allreads, reverse=True), k, m)
    wpe = want["wpe"]
    assert np.array_equal(np.concatenate([o[1] for o in outs]).reshape(-1, wpe), want["edges"])
    assert np.array_equal(sum(o[2] for o in outs), want["bucket_count"])
    assert np.array_equal(sum(o[3] for o in outs), want["hist"])
    assert np.array_equal(np.concatenate([o[4] for o in outs]), want["first_0_out"])
    assert np.array_equal(np.concatenate([o[5] for o in outs]), want["last_0_in"])


@pytest.mark.parametrize("world,k", [(2, 21), (2, 39), (3, 29)])
def test_seq2sdbg_ranks_on_one_gpu(world, k):
    import oracle_binding as ob
    from test_dist_cpu import _check_sdbg_ranges, _seqs_with_mult
    outs = _run2("seq2sdbg", k, 0, world)
    seqs, mult = [], []
    for r in range(world):
        s, m_ = _seqs_with_mult(50 + r)
        seqs += s
        mult.append(m_)
    want = ob.seq2sdbg(ob.Package(seqs, reverse=False), np.concatenate(mult), k)
    _check_sdbg_ranges(outs, want)


@pytest.mark.parametrize("world,k,m,mode", [(2, 21, 2, "mercy"), (3, 27, 3, "mercy"), (2, 21, 2, "mercy_exact")])
def test_read2sdbg_mercy_ranks_on_one_gpu(world, k, m, mode):
    import oracle_binding as ob
    from test_dist_cpu import _check_sdbg_ranges
    outs = _run2(mode, k, m, world)
    allreads = []
    for r in range(world):
        allreads += _reads(100 + r)
    pkg = ob.Package(allreads, reverse=True)
    s1 = ob.s1(pkg, k, m, tie_stable=mode == "mercy")
    n_want, solid = ob.s2_add_mercy(pkg, k, s1["is_solid"], s1["mercy"])
    assert sum(o[7] for o in outs) == n_want and n_want > 0
    _check_sdbg_ranges(outs, ob.s2(pkg, k, m, solid))


@pytest.mark.parametrize("world,k,m,mode", [(2, 21, 2, "passes"), (2, 27, 2, "passes_mercy"), (2, 31, 1, "passes")])
def test_read2sdbg_passes_ranks_on_one_gpu(world, k, m, mode):
    """memory-bounded operation on every rank: 3 bucket sub-range passes per stage"""
    import oracle_binding as ob
    from test_dist_cpu import _check_sdbg_ranges
    outs = _run2(mode, k, m, world)
    allreads = []
    for r in range(world):
        allreads += _reads(100 + r)
    pkg = ob.Package(allreads, reverse=True)
    if m > 1:
        s1 = ob.s1(pkg, k, m, tie_stable=True)
        solid = s1["is_solid"]
        if mode == "passes_mercy":
            n_want, solid = ob.s2_add_mercy(pkg, k, s1["is_solid"], s1["mercy"])
            assert sum(o[7] for o in outs) == n_want and n_want > 0
        want = ob.s2(pkg, k, m, solid)
    else:
        want = ob.s2(pkg, k, 1, None)
    _check_sdbg_ranges(outs, want)


@pytest.mark.parametrize("world,k,m", [(2, 21, 2), (3, 21, 3), (3, 26, 2), (3, 27, 2)])
def test_rank_tagged_compact_items(world, k, m, monkeypatch):
    """Past 2^32 global positions the compact stage-1 records keep rank-local positions and carry the source rank in
    spare key bits; MHX_S1_FORCE_TAGGED takes that path at test sizes (k=26 is the last k with 8 spare bits; k=27
    must fall back to plain global positions)."""
    import oracle_binding as ob
    monkeypatch.setenv("MHX_S1_FORCE_TAGGED", "1")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, m, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    allreads = []
    for r in range(world):
        allreads += _reads(100 + r)
    pkg = ob.Package(allreads, reverse=True)
    s1 = ob.s1(pkg, k, m)
    want = ob.s2(pkg, k, m, s1["is_solid"])
    assert np.array_equal(sum(o[7] for o in outs), s1["hist"])
    off = np.concatenate([want["bucket_off"], [len(want["bytes"])]]).astype(np.int64)
    for rank, lo, hi, byts, b_items, b_tips, b_large, _ in outs:
        assert np.array_equal(b_items[lo:hi], want["bucket_items"][lo:hi])
        assert byts == want["bytes"][off[lo]:off[hi]].tobytes()
