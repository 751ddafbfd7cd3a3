"""GPU: the C++ multi-GPU drivers with the ranks as separate PROCESSES (world_size 2 and 3, all on cuda:0), the bytes moved by
torch.distributed over gloo through libmhx's hosted communicator (include/mhx.h mhx_comm_init_hosted, megahit_amd/hosted.py).
tests/test_gpu_comm.py runs the same drivers with thread ranks behind the in-process transport; RCCL needs one GPU per rank
and only runs in the driver's multi-GPU bench.  Every rank's output against the oracle on the concatenated input."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, mode, k, m, opts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from dist_inputs import reads_of, seqs_with_mult
    from megahit_amd import lib
    e = lib.Engine(0)
    for name, v in (opts or {}).items():
        e.set_option(name, v)
    if mode == "seq2sdbg":
        seqs, mult = seqs_with_mult(50 + rank)
        pkg = ob.Package(seqs, reverse=False)
        e.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
        e.load_multiplicity(mult)
    else:
        pkg = ob.Package(reads_of(100 + rank, n_pairs=600), reverse=True)
        e.load_sequences(pkg.words(), pkg.n_seqs, 0, pkg.start())
    cm = lib.Comm.hosted(e, dist, rank, world)
    out = {"rank": rank}
    if mode == "read2sdbg":
        cm.setup(1, k, m)
        r1, r2, _nm = cm.read2sdbg(k, m)
        out["hist"] = e.fetch(lib.BUF_MUL_HIST, np.int64) if m > 1 else None
        out["n_solid"] = int(r1.n_solid)
        out["plan"] = e.last_s1_plan()
    elif mode == "count":
        cm.setup(3, k, m)
        cm.count(k, m)
        out["edges"] = e.fetch(lib.BUF_EDGES, np.uint32)
        out["hist"] = e.fetch(lib.BUF_MUL_HIST, np.int64)
        out["first"] = e.fetch(lib.BUF_FIRST_0_OUT, np.uint32)
        out["last"] = e.fetch(lib.BUF_LAST_0_IN, np.uint32)
    else:
        cm.setup(0, k, 0)
        cm.seq2sdbg(k)
    if mode != "count":
        out["bytes"] = e.fetch(lib.BUF_SDBG_BYTES, np.uint8).tobytes()
        out["items"] = e.fetch(lib.BUF_BUCKET_COUNT, np.uint64)
        out["tips"] = e.fetch(lib.BUF_BUCKET_TIPS, np.uint64)
    else:
        out["items"] = e.fetch(lib.BUF_BUCKET_COUNT, np.uint64)
    q.put(out)
    dist.barrier()
    cm.close()
    e.close()
    dist.destroy_process_group()


def run_world(world, mode, k, m, opts=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, k, m, opts, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=600) for _ in range(world)], key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return outs


@pytest.mark.parametrize("world,k,m,opts", [(2, 21, 2, None), (3, 21, 2, {"dist_max_items": 40000}), (2, 27, 1, None), (2, 21, 2, {"dist_presort": 0}),
                                            # round 6: super-k-mer records exchanged by bin (comm.hip dist_s1_skm), rank processes over the hosted transport
                                            (2, 21, 2, {"s1_skm": 2, "s1_skm_max_bin": 1 << 30, "s1_var_min_fill": 5}),
                                            (3, 22, 2, {"s1_skm": 2, "s1_skm_max_bin": 1 << 30, "s1_var_min_fill": 5, "s1_stream_fill": 40})])
def test_read2sdbg_rank_processes(world, k, m, opts):
    import oracle_binding as ob
    from dist_inputs import reads_of
    outs = run_world(world, "read2sdbg", k, m, opts)
    pkg = ob.Package(sum((reads_of(100 + r, n_pairs=600) for r in range(world)), []), reverse=True)
    if opts and opts.get("s1_skm") == 2:
        assert all(o["plan"].startswith("super-k-mers") and "exchanged by bin" in o["plan"] for o in outs), [o["plan"] for o in outs]
    if m > 1:
        s1 = ob.s1(pkg, k, m)
        want = ob.s2(pkg, k, m, s1["is_solid"])
        assert np.array_equal(sum(o["hist"] for o in outs), s1["hist"])
        assert sum(o["n_solid"] for o in outs) == int(sum(bin(int(x)).count("1") for x in s1["is_solid"]))
    else:
        want = ob.s2(pkg, k, 1, None)
    assert b"".join(o["bytes"] for o in outs) == want["bytes"].tobytes()
    assert np.array_equal(sum(o["items"] for o in outs), want["bucket_items"])
    assert np.array_equal(sum(o["tips"] for o in outs), want["bucket_tips"])


def test_count_rank_processes():
    import oracle_binding as ob
    from dist_inputs import reads_of
    world, k, m = 2, 21, 2
    outs = run_world(world, "count", k, m)
    want = ob.count(ob.Package(sum((reads_of(100 + r, n_pairs=600) for r in range(world)), []), reverse=True), k, m)
    assert np.array_equal(np.concatenate([o["edges"] for o in outs]).reshape(-1, want["wpe"]), want["edges"])
    assert np.array_equal(sum(o["hist"] for o in outs), want["hist"])
    assert np.array_equal(np.concatenate([o["first"] for o in outs]), want["first_0_out"])
    assert np.array_equal(np.concatenate([o["last"] for o in outs]), want["last_0_in"])


def test_seq2sdbg_rank_processes():
    import oracle_binding as ob
    from dist_inputs import seqs_with_mult
    world, k = 3, 39
    outs = run_world(world, "seq2sdbg", k, 0)
    seqs, mult = [], []
    for r in range(world):
        s, m_ = seqs_with_mult(50 + r)
        seqs += s
        mult.append(m_)
    want = ob.seq2sdbg(ob.Package(seqs, reverse=False), np.concatenate(mult), k)
    assert b"".join(o["bytes"] for o in outs) == want["bytes"].tobytes()
    assert np.array_equal(sum(o["items"] for o in outs), want["bucket_items"])
