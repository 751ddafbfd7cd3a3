"""CPU, world_size 2 on gloo: the multi-GPU orchestration of megahit_amd/dist.py (bucket partition,
count + item all-to-all, is_solid reduction, per-rank SdBG emission) reproduces the single-process
result bucket by bucket.  The per-rank compute is the oracle-backed stand-in of tests/cpu_engine.py."""
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


def _reads(seed, n_pairs=250):
    from megahit_amd import synth
    rng = np.random.default_rng(seed)
    genome = np.random.default_rng(99).integers(0, 4, size=3000, dtype=np.uint8)
    r = synth.gen_pe_reads(n_pairs, 3000, read_len=100, frag=250, err=0.01, seed=seed, genome=genome)
    return [x[: rng.integers(10, 101)] for x in r]


def _worker(rank, world, port, k, m, balanced, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_engine import OracleEngine
    from megahit_amd import dist as mdist
    eng = OracleEngine(_reads(100 + rank))
    bb = None
    if balanced:  # skewed ownership: rank 0 owns few buckets
        bb = np.array([0, 9000, 65536], dtype=np.uint32) if world == 2 else None
    runner = mdist.DistRead2Sdbg(eng, k, m, rank, world, torch.device("cpu"), bucket_begin=bb)
    r1, r2 = runner.step()
    lo, hi = int(runner.bucket_begin[rank]), int(runner.bucket_begin[rank + 1])
    q.put((rank, lo, hi, eng.sdbg["bytes"].tobytes(), eng.sdbg["bucket_items"], eng.sdbg["bucket_tips"], eng.sdbg["bucket_large"],
           eng.hist if m > 1 else None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k,m,balanced", [(21, 2, False), (27, 2, True), (21, 1, False)])
def test_two_ranks_equal_single_process(k, m, balanced):
    import oracle_binding as ob
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, m, balanced, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference result on the concatenated read set
    allreads = _reads(100) + _reads(101)
    pkg = ob.Package(allreads, reverse=True)
    if m > 1:
        s1 = ob.s1(pkg, k, m)
        want = ob.s2(pkg, k, m, s1["is_solid"])
        hist = sum(o[7] for o in outs)
        assert np.array_equal(hist, s1["hist"])
    else:
        want = ob.s2(pkg, k, 1, None)
    off = np.concatenate([want["bucket_off"], [len(want["bytes"])]]).astype(np.int64)
    covered = 0
    for rank, lo, hi, byts, b_items, b_tips, b_large, _ in outs:
        assert np.array_equal(b_items[lo:hi], want["bucket_items"][lo:hi])
        assert np.array_equal(b_tips[lo:hi], want["bucket_tips"][lo:hi])
        assert np.array_equal(b_large[lo:hi], want["bucket_large"][lo:hi])
        assert b_items[:lo].sum() == 0 and b_items[hi:].sum() == 0
        assert byts == want["bytes"][off[lo]:off[hi]].tobytes()
        covered += hi - lo
    assert covered == 65536


def test_partitions():
    from megahit_amd import dist as mdist
    assert list(mdist.equal_partition(4)) == [0, 16384, 32768, 49152, 65536]
    w = np.zeros(65536)
    w[:100] = 10.0
    w[100:] = 1e-3
    bb = mdist.balanced_partition(w, 4)
    assert bb[0] == 0 and bb[-1] == 65536 and (np.diff(bb.astype(np.int64)) >= 0).all()
    assert bb[1] <= 30 and bb[2] <= 60
