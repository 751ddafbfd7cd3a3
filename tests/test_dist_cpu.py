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


def _seqs_with_mult(rank_seed):
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


def _worker2(rank, world, port, mode, k, m, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_engine import OracleEngine
    from megahit_amd import dist as mdist
    dev = torch.device("cpu")
    if mode.startswith("passes"):
        pass
    if mode == "count":
        eng = OracleEngine(_reads(100 + rank))
        runner = mdist.DistCount(eng, k, m, rank, world, dev)
        runner.step()
        q.put((rank, eng.count["edges"], eng.count["bucket_count"], eng.count["hist"], eng.first_0_out, eng.last_0_in))
    elif mode == "seq2sdbg":
        seqs, mult = _seqs_with_mult(50 + rank)
        eng = OracleEngine(seqs, reverse=False, mult=mult)
        runner = mdist.DistSeq2Sdbg(eng, k, rank, world, dev)
        runner.step()
        lo, hi = int(runner.bucket_begin[rank]), int(runner.bucket_begin[rank + 1])
        q.put((rank, lo, hi, eng.sdbg["bytes"].tobytes(), eng.sdbg["bucket_items"], eng.sdbg["bucket_tips"], eng.sdbg["bucket_large"]))
    elif mode in ("passes", "passes_mercy"):  # memory-bounded: every stage in 3 bucket sub-range passes
        eng = OracleEngine(_reads(100 + rank))
        got = dict(bytes=[], items=0, tips=0, large=0)

        def collect(p, r2):
            got["bytes"].append(eng.sdbg["bytes"].tobytes())
            got["items"] = got["items"] + eng.sdbg["bucket_items"]
            got["tips"] = got["tips"] + eng.sdbg["bucket_tips"]
            got["large"] = got["large"] + eng.sdbg["bucket_large"]

        runner = mdist.DistRead2Sdbg(eng, k, m, rank, world, dev, need_mercy=1 if mode == "passes_mercy" else 0, n_passes=3,
                                     on_s2_pass=collect)
        runner.step()
        lo, hi = int(runner.bucket_begin[rank]), int(runner.bucket_begin[rank + 1])
        q.put((rank, lo, hi, b"".join(got["bytes"]), got["items"], got["tips"], got["large"], runner.n_mercy))
    else:  # read2sdbg with mercy
        eng = OracleEngine(_reads(100 + rank))
        runner = mdist.DistRead2Sdbg(eng, k, m, rank, world, dev, need_mercy=1)
        runner.step()
        lo, hi = int(runner.bucket_begin[rank]), int(runner.bucket_begin[rank + 1])
        q.put((rank, lo, hi, eng.sdbg["bytes"].tobytes(), eng.sdbg["bucket_items"], eng.sdbg["bucket_tips"], eng.sdbg["bucket_large"],
               runner.n_mercy))
    dist.barrier()
    dist.destroy_process_group()


def _run2(mode, k, m, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker2, args=(r, world, port, mode, k, m, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return outs


def _check_sdbg_ranges(outs, want):
    off = np.concatenate([want["bucket_off"], [len(want["bytes"])]]).astype(np.int64)
    for o in outs:
        rank, lo, hi, byts, b_items, b_tips, b_large = o[:7]
        assert np.array_equal(b_items[lo:hi], want["bucket_items"][lo:hi])
        assert np.array_equal(b_tips[lo:hi], want["bucket_tips"][lo:hi])
        assert np.array_equal(b_large[lo:hi], want["bucket_large"][lo:hi])
        assert b_items[:lo].sum() == 0 and b_items[hi:].sum() == 0
        assert byts == want["bytes"][off[lo]:off[hi]].tobytes()


@pytest.mark.parametrize("k,m", [(21, 2), (31, 3)])
def test_two_ranks_count(k, m):
    """count: items to bucket owners, first_0_out/last_0_in events routed back to the read owners"""
    import oracle_binding as ob
    outs = _run2("count", k, m)
    r0, r1 = _reads(100), _reads(101)
    want = ob.count(ob.Package(r0 + r1, reverse=True), k, m)
    assert np.array_equal(np.concatenate([o[1] for o in outs]), want["edges"])  # rank order = bucket order = sorted order
    assert np.array_equal(sum(o[2] for o in outs), want["bucket_count"])
    assert np.array_equal(sum(o[3] for o in outs), want["hist"])
    assert np.array_equal(np.concatenate([o[4] for o in outs]), want["first_0_out"])
    assert np.array_equal(np.concatenate([o[5] for o in outs]), want["last_0_in"])
    assert (want["first_0_out"] != 0xFFFFFFFF).any() and (want["last_0_in"] != 0xFFFFFFFF).any()


@pytest.mark.parametrize("k", [21, 39])
def test_two_ranks_seq2sdbg(k):
    import oracle_binding as ob
    outs = _run2("seq2sdbg", k, 0)
    s0, m0 = _seqs_with_mult(50)
    s1, m1 = _seqs_with_mult(51)
    want = ob.seq2sdbg(ob.Package(s0 + s1, reverse=False), np.concatenate([m0, m1]), k)
    assert want["bucket_items"].sum() > 0
    _check_sdbg_ranges(outs, want)


@pytest.mark.parametrize("k,m", [(21, 2), (27, 3)])
def test_two_ranks_read2sdbg_with_mercy(k, m):
    """mercy candidates are produced by the bucket owners and routed to the ranks holding the reads"""
    import oracle_binding as ob
    outs = _run2("mercy", k, m)
    pkg = ob.Package(_reads(100) + _reads(101), reverse=True)
    s1 = ob.s1(pkg, k, m, tie_stable=True)
    n_want, solid = ob.s2_add_mercy(pkg, k, s1["is_solid"], s1["mercy"])
    assert sum(o[7] for o in outs) == n_want and n_want > 0
    _check_sdbg_ranges(outs, ob.s2(pkg, k, m, solid))


@pytest.mark.parametrize("mode,k,m", [("passes", 21, 2), ("passes_mercy", 27, 2), ("passes", 21, 1)])
def test_two_ranks_read2sdbg_in_passes(mode, k, m):
    """bucket sub-range passes on every rank (filtered extraction, accumulating stage 1) = the single-pass result"""
    import oracle_binding as ob
    outs = _run2(mode, k, m)
    pkg = ob.Package(_reads(100) + _reads(101), reverse=True)
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


def test_partitions():
    from megahit_amd import dist as mdist
    assert list(mdist.equal_partition(4)) == [0, 16384, 32768, 49152, 65536]
    w = np.zeros(65536)
    w[:100] = 10.0
    w[100:] = 1e-3
    bb = mdist.balanced_partition(w, 4)
    assert bb[0] == 0 and bb[-1] == 65536 and (np.diff(bb.astype(np.int64)) >= 0).all()
    assert bb[1] <= 30 and bb[2] <= 60


@pytest.mark.parametrize("world,item_bytes,max_msg", [(2, 12, 1000), (4, 16, 4096), (8, 8, 64), (3, 12, 1 << 30)])
def test_p2p_plan_is_consistent_across_ranks(world, item_bytes, max_msg):
    """The RCCL item exchange is chunked point-to-point traffic: simulate every rank's schedule and check that each
    round's sends are matched by receives of the same size on the peer, that all bytes are covered exactly once and
    that no message exceeds the limit."""
    from megahit_amd import dist as mdist
    rng = np.random.default_rng(world * 100 + item_bytes)
    counts = rng.integers(0, 400, size=(world, world))  # counts[s][d]: items s sends to d
    counts[rng.integers(0, world)][rng.integers(0, world)] = 0
    plans = [mdist.p2p_plan(counts[r], counts[:, r], item_bytes, r, world, max_msg) for r in range(world)]
    n_rounds = max(len(p[1]) for p in plans)
    sent = [np.zeros(int(counts[r].sum()) * item_bytes, dtype=np.int32) for r in range(world)]
    got = [np.zeros(int(counts[:, r].sum()) * item_bytes, dtype=np.int32) for r in range(world)]
    for r, (self_copy, _) in enumerate(plans):
        if self_copy:
            s_lo, r_lo, nb = self_copy
            assert nb == counts[r][r] * item_bytes
            sent[r][s_lo:s_lo + nb] += 1
            got[r][r_lo:r_lo + nb] += 1
    for c in range(n_rounds):
        sends, recvs = {}, {}
        for r, (_, rounds) in enumerate(plans):
            for kind, peer, lo, hi in (rounds[c] if c < len(rounds) else []):
                assert 0 < hi - lo <= max(item_bytes, max_msg) and (hi - lo) % item_bytes == 0
                if kind == "send":
                    assert (r, peer) not in sends
                    sends[(r, peer)] = hi - lo
                    sent[r][lo:hi] += 1
                else:
                    assert (peer, r) not in recvs
                    recvs[(peer, r)] = hi - lo
                    got[r][lo:hi] += 1
        assert sends == recvs  # same pairs, same sizes, same round
    for r in range(world):
        assert (sent[r] == 1).all() and (got[r] == 1).all()


def _p2p_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MHX_DIST_FORCE_P2P"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from megahit_amd import dist as mdist
    x = mdist.Exchanger(rank, world, torch.device("cpu"))
    x.MAX_MSG_BYTES = 100  # several rounds per pair
    item_bytes = 12
    counts = np.random.default_rng(7).integers(0, 40, size=(world, world))  # same matrix on every rank
    counts[0][1] = 0
    send = np.concatenate([np.full(int(counts[rank][d]) * item_bytes, 16 * rank + d, dtype=np.uint8) for d in range(world)])
    send[::7] ^= 0x80  # not constant inside a segment
    recv_counts = x.exchange_counts(counts[rank])
    assert list(recv_counts) == [int(counts[s][rank]) for s in range(world)]
    recv = torch.zeros(int(recv_counts.sum()) * item_bytes, dtype=torch.uint8)
    x.exchange_items(torch.from_numpy(send), counts[rank], recv, recv_counts, item_bytes)
    q.put((rank, send.tobytes(), recv.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_p2p_exchange_three_ranks():
    """The code path RCCL runs (self copy + batched isend/irecv in rounds of bounded messages), executed for real on
    gloo with 3 processes: every rank receives exactly the bytes its peers addressed to it."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_p2p_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts = np.random.default_rng(7).integers(0, 40, size=(world, world))
    counts[0][1] = 0
    sb = [np.concatenate([[0], np.cumsum(counts[r])]) * 12 for r in range(world)]
    for r in range(world):
        want = b"".join(outs[s][1][int(sb[s][r]):int(sb[s][r + 1])] for s in range(world))
        assert outs[r][2] == want


def test_two_ranks_with_p2p_exchange(monkeypatch):
    """the whole read2sdbg + mercy orchestration over the point-to-point exchange path (as on RCCL)"""
    monkeypatch.setenv("MHX_DIST_FORCE_P2P", "1")
    test_two_ranks_read2sdbg_with_mercy(21, 2)
    test_two_ranks_count(21, 2)


def test_three_ranks_read2sdbg_and_count():
    """odd world size: uneven bucket ranges (65536 / 3), three-way exchange and routing"""
    import oracle_binding as ob
    world, k, m = 3, 21, 2
    pkg = ob.Package(_reads(100) + _reads(101) + _reads(102), reverse=True)
    outs = _run2("mercy", k, m, world=world)
    s1 = ob.s1(pkg, k, m, tie_stable=True)
    n_want, solid = ob.s2_add_mercy(pkg, k, s1["is_solid"], s1["mercy"])
    assert sum(o[7] for o in outs) == n_want
    _check_sdbg_ranges(outs, ob.s2(pkg, k, m, solid))
    outs = _run2("count", k, m, world=world)
    want = ob.count(pkg, k, m)
    assert np.array_equal(np.concatenate([o[1] for o in outs]), want["edges"])
    assert np.array_equal(np.concatenate([o[4] for o in outs]), want["first_0_out"])
    assert np.array_equal(np.concatenate([o[5] for o in outs]), want["last_0_in"])
