"""Multi-GPU count / read2sdbg / seq2sdbg: one process per GPU, lv1 buckets sharded over ranks, items moved with an
all-to-all (RCCL over xGMI through torch.distributed; `gloo` on CPU for tests).

Per stage (S1, then S2) every rank
  1. extracts the items of ITS reads and partitions them by bucket owner   (engine.dist_extract)
  2. exchanges per-owner counts, then the items themselves                (all_to_all_single)
  3. sorts and reduces the buckets it owns                                (engine.dist_process_*)
After S1 the global is_solid bitmap (every bit is set by exactly one rank, because every (k+1)-mer
occurrence lives in exactly one (k-1)-mer item: reference src/sorting/read_to_sdbg_s1.cpp:259-292,464)
is summed over ranks and each rank keeps the slice covering its own reads for S2.

The reference has no counterpart: it shards buckets over OpenMP threads inside one process
(reference src/sorting/base_engine.cpp:213-223,318-327).  `engine` is anything with the phased
interface of megahit_amd.lib.Engine; tests drive the same class on CPU tensors with an oracle-backed
stand-in, so the communication logic is covered without GPUs.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

_T0 = time.time()


def _dbg(msg):
    if os.environ.get("MHX_DIST_DEBUG"):
        print("[dist %.2f s rank %s] %s" % (time.time() - _T0, os.environ.get("RANK", "?"), msg), file=sys.stderr, flush=True)


def _sync(t):
    """Collectives run on torch's streams, the library on its own: finish the collective before libmhx touches the data."""
    if t.is_cuda:
        torch.cuda.synchronize(t.device)

NUM_BUCKETS = 65536
STAGE_S1 = 1
STAGE_S2 = 2
STAGE_COUNT = 3
STAGE_SEQ2SDBG = 4
STAGE_S1_MERCY = 5
ROUTE_COUNT_EVENTS = 1
ROUTE_MERCY_CAND = 2
BUF_IS_SOLID = 6


class _DevPtr:
    """Expose a raw device pointer to torch through the CUDA array interface (works on ROCm builds)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes(ptr, nbytes, device):
    """uint8 tensor aliasing [ptr, ptr+nbytes) of device memory owned by libmhx."""
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(_DevPtr(ptr, nbytes), device=device)


def equal_partition(world):
    """bucket_begin[world+1]: contiguous, equally wide bucket ranges."""
    return np.array([(NUM_BUCKETS * r) // world for r in range(world + 1)], dtype=np.uint32)


def balanced_partition(bucket_weight, world):
    """Contiguous bucket ranges with ~equal total weight (e.g. the global lv1 bucket histogram)."""
    w = np.asarray(bucket_weight, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    begin = [0]
    for r in range(1, world):
        begin.append(int(np.searchsorted(cum, total * r / world, side="left")))
    begin.append(NUM_BUCKETS)
    begin = np.maximum.accumulate(np.minimum(np.array(begin), NUM_BUCKETS))
    begin[0] = 0
    return begin.astype(np.uint32)


def p2p_plan(send_counts, recv_counts, item_bytes, rank, world, max_msg_bytes):
    """Chunked point-to-point schedule of one all-to-all of byte buffers whose per-peer segments lie back to back.
    -> (self_copy (send_offset, recv_offset, n_bytes) or None, rounds); a round is a list of
    ("send" | "recv", peer, lo, hi) byte ranges.  Chunk c of the segment for/from a peer goes in round c, pairs in ring
    order (send to rank+d, receive from rank-d), so every message a rank sends in round c is received by its peer in
    the same round with the same size."""
    sb = np.concatenate([[0], np.cumsum(np.asarray(send_counts, dtype=np.int64))]) * item_bytes
    rb = np.concatenate([[0], np.cumsum(np.asarray(recv_counts, dtype=np.int64))]) * item_bytes
    n_self = int(sb[rank + 1] - sb[rank])
    assert n_self == int(rb[rank + 1] - rb[rank])
    self_copy = (int(sb[rank]), int(rb[rank]), n_self) if n_self else None
    chunk = max(item_bytes, max_msg_bytes // item_bytes * item_bytes)
    n_rounds = 0
    for p in range(world):
        if p != rank:
            n_rounds = max(n_rounds, -(-int(sb[p + 1] - sb[p]) // chunk), -(-int(rb[p + 1] - rb[p]) // chunk))
    rounds = []
    for c in range(n_rounds):
        plan = []
        for d in range(1, world):
            to, frm = (rank + d) % world, (rank - d) % world
            lo, hi = int(sb[to]) + c * chunk, min(int(sb[to + 1]), int(sb[to]) + (c + 1) * chunk)
            if lo < hi:
                plan.append(("send", to, lo, hi))
            lo, hi = int(rb[frm]) + c * chunk, min(int(rb[frm + 1]), int(rb[frm]) + (c + 1) * chunk)
            if lo < hi:
                plan.append(("recv", frm, lo, hi))
        if plan:
            rounds.append(plan)
    return self_copy, rounds


class Exchanger:
    """The collectives of the distributed path, on whatever device the tensors live."""

    def __init__(self, rank, world, device, group=None):
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.force_p2p = bool(os.environ.get("MHX_DIST_FORCE_P2P"))  # tests: the RCCL code path on gloo

    def exchange_counts(self, send_counts):
        """send_counts[p] = items this rank sends to p  ->  recv_counts[p] = items p sends to this rank."""
        s = torch.as_tensor(np.asarray(send_counts, dtype=np.int64), device=self.device)
        r = torch.empty_like(s)
        _dbg("exchange_counts %s" % list(np.asarray(send_counts)))
        dist.all_to_all_single(r, s, group=self.group)
        return r.cpu().numpy().astype(np.uint64)

    MAX_MSG_BYTES = 1 << 28  # 256 MiB per message (a 0.5 GB all-to-all ran fine on RCCL 2.26, a 16 GB one hung)

    def exchange_items(self, send, send_counts, recv, recv_counts, item_bytes):
        """send/recv: uint8 tensors holding the per-peer segments back to back (peers ascending); counts in items."""
        _dbg("exchange_items %d B/item, send %s recv %s" % (item_bytes, [int(c) for c in send_counts], [int(c) for c in recv_counts]))
        if dist.get_backend(self.group) != "nccl" and not self.force_p2p:
            if item_bytes % 8 == 0:
                q, dt = item_bytes // 8, torch.int64
            else:  # 12-byte records (compact stage-1 items)
                assert item_bytes % 4 == 0
                q, dt = item_bytes // 4, torch.int32
            dist.all_to_all_single(recv.view(dt), send.view(dt), [int(c) * q for c in recv_counts], [int(c) * q for c in send_counts],
                                   group=self.group)
            _sync(recv)
            return
        # RCCL: the rank's own segment is a device copy; every other segment goes as point-to-point messages of at most
        # 256 MiB (one 16 GB message hung RCCL 2.26 on MI355X), all pairs of a round grouped in one batch.  Sender and
        # receiver derive the same chunking from the exchanged counts, so no further agreement is needed.
        self_copy, rounds = p2p_plan(send_counts, recv_counts, item_bytes, self.rank, self.world, self.MAX_MSG_BYTES)
        if self_copy is not None:
            s_lo, r_lo, nb = self_copy
            recv[r_lo:r_lo + nb].copy_(send[s_lo:s_lo + nb])
        n_rounds = len(rounds)
        for plan in rounds:
            ops = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, (send if kind == "send" else recv)[lo:hi], peer, group=self.group)
                   for kind, peer, lo, hi in plan]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        _sync(recv)
        _dbg("exchange_items done (%d rounds)" % n_rounds)

    def sum_bitmap_and_take_slice(self, bitmap_i64, words_per_rank):
        """bitmap_i64: int64 tensor of world*words_per_rank words.  Returns this rank's summed slice."""
        if dist.get_backend(self.group) == "nccl":
            out = torch.empty(words_per_rank, dtype=torch.int64, device=bitmap_i64.device)
            _dbg("reduce_scatter of %d words" % bitmap_i64.numel())
            dist.reduce_scatter_tensor(out, bitmap_i64, op=dist.ReduceOp.SUM, group=self.group)
            _sync(out)
            _dbg("reduce_scatter done")
            return out
        dist.all_reduce(bitmap_i64, op=dist.ReduceOp.SUM, group=self.group)  # gloo: no reduce_scatter
        _sync(bitmap_i64)
        return bitmap_i64[self.rank * words_per_rank:(self.rank + 1) * words_per_rank].clone()

    def max_int(self, v):
        t = torch.tensor([int(v)], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())


class _DistBase:
    """Shared plumbing: bucket partition, global read layout (rank r's bases start at r*stride, stride a multiple of
    64 bits) and the two kinds of all-to-all (items to bucket owners, position-keyed records back to read owners)."""

    def __init__(self, engine, rank, world, device, bucket_begin=None, staging=None, global_layout=True):
        self.e, self.rank, self.world, self.device = engine, rank, world, device
        self.x = Exchanger(rank, world, device)
        # engines that keep items in GPU memory while the process group is CPU-only (gloo) stage through host
        self.staging = staging
        self.bucket_begin = equal_partition(world) if bucket_begin is None else np.asarray(bucket_begin, dtype=np.uint32)
        engine.set_partition(rank, world, self.bucket_begin)
        if global_layout:
            stride = self.x.max_int(engine.n_bases)
            self.stride_words = (stride + 63) // 64
            self.stride = self.stride_words * 64
            engine.set_global_layout(rank * self.stride, world * self.stride)

    def plan_passes(self, stage, k, m, n_passes):
        """Split every owner's bucket range into n_passes contiguous sub-ranges of about equal GLOBAL weight; pass p
        handles sub-range p of every owner, so each pass keeps all ranks busy with 1/n_passes of the items.
        -> [(keep mask uint8[65536], items of the LOCAL reads in it)]"""
        local = np.asarray(self.e.bucket_histogram(stage, k, m), dtype=np.int64)
        t = torch.as_tensor(local.copy(), device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.x.group)
        glob = t.cpu().numpy().astype(np.float64)
        masks = [np.zeros(NUM_BUCKETS, dtype=np.uint8) for _ in range(n_passes)]
        for r in range(self.world):
            lo, hi = int(self.bucket_begin[r]), int(self.bucket_begin[r + 1])
            if hi <= lo:
                continue
            cum = np.concatenate([[0.0], np.cumsum(glob[lo:hi])])
            cuts = [lo + int(np.searchsorted(cum, cum[-1] * p / n_passes, side="left")) for p in range(1, n_passes)]
            cuts = np.maximum.accumulate(np.clip(np.array([lo] + cuts + [hi]), lo, hi))
            for p in range(n_passes):
                masks[p][cuts[p]:cuts[p + 1]] = 1
        return [(mk, int(local[mk.astype(bool)].sum())) for mk in masks]

    def _move(self, ptr, n_items, item_bytes, counts):
        recv_counts = self.x.exchange_counts(counts)
        n_recv = int(recv_counts.sum())
        rptr = self.e.dist_recv_buffer(n_recv, item_bytes)
        send = self.e.as_tensor(ptr, int(n_items) * item_bytes, self.device)
        recv = self.e.as_tensor(rptr, n_recv * item_bytes, self.device)
        if self.staging == "host":  # GPU engine + gloo group: bounce through host memory
            hs, hr = send.cpu(), torch.empty(n_recv * item_bytes, dtype=torch.uint8)
            self.x.exchange_items(hs, counts, hr, recv_counts, item_bytes)
            recv.copy_(hr)
        else:
            self.x.exchange_items(send, counts, recv, recv_counts, item_bytes)
        return n_recv

    def _alltoall(self, stage, k, m):
        """items of the local sequences -> their bucket owners"""
        return self._move(*self.e.dist_extract(stage, k, m))

    def _route(self, which):
        """position-keyed records of the owned buckets -> the ranks holding those reads"""
        n = self._move(*self.e.dist_route_records(which, self.stride))
        self.e.dist_apply_routed(which, n)
        return n


class DistCount(_DistBase):
    """count over `world` ranks (reference KmerCounter, src/sorting/kmer_counter.cpp).  After step(): the engine holds
    the solid edges / bucket counts / multiplicity histogram of this rank's bucket range (the histogram is summed over
    ranks by the caller) and first_0_out / last_0_in of this rank's reads."""

    def __init__(self, engine, k, min_count, rank, world, device, bucket_begin=None, staging=None):
        super().__init__(engine, rank, world, device, bucket_begin, staging)
        self.k, self.m = k, min_count

    def step(self):
        n = self._alltoall(STAGE_COUNT, self.k, self.m)
        r = self.e.dist_process_count(self.k, self.m, n)
        self._route(ROUTE_COUNT_EVENTS)
        return r


class DistSeq2Sdbg(_DistBase):
    """seq2sdbg over `world` ranks (reference SeqToSdbg, src/sorting/seq_to_sdbg.cpp): every rank loads any share of
    the edges/contigs (+ multiplicities); items carry no positions, so one all-to-all is the whole exchange."""

    def __init__(self, engine, k, rank, world, device, bucket_begin=None, staging=None):
        super().__init__(engine, rank, world, device, bucket_begin, staging, global_layout=False)
        self.k = k

    def step(self):
        n = self._alltoall(STAGE_SEQ2SDBG, self.k, 0)
        return self.e.dist_process_seq2sdbg(self.k, n)


class DistRead2Sdbg(_DistBase):
    """read2sdbg (S1 [+ mercy] + S2) over `world` ranks.  After step(): engine holds this rank's SdBG
    records (its bucket range) exactly as the single-GPU engine would for those buckets.
    need_mercy: 0 none, 1 stable tie order, 2 reference-exact tie order (as mhx_read2sdbg_s1).
    n_passes > 1: memory-bounded operation (megahit_amd/passes.py): every stage runs once per sub-range of the owned
    buckets; stage-1 state accumulates, the stage-2 output of pass p is handed to on_s2_pass(p, result) (the
    engine's result buffers then hold that sub-range) before the next pass overwrites it."""

    def __init__(self, engine, k, min_count, rank, world, device, bucket_begin=None, staging=None, need_mercy=0, n_passes=1,
                 batch_bytes=0, on_s2_pass=None):
        super().__init__(engine, rank, world, device, bucket_begin, staging)
        self.k, self.m, self.need_mercy = k, min_count, int(need_mercy)
        self.n_passes, self.batch_bytes, self.on_s2_pass = int(n_passes), batch_bytes, on_s2_pass
        self.n_mercy = 0

    def _passes(self, stage):
        if self.n_passes <= 1:
            return [(None, 0)]
        return self.plan_passes(stage, self.k, self.m, self.n_passes)

    def step(self):
        r1 = None
        try:
            if self.m > 1:  # stage 1 is skipped when every edge is solid (reference main_sdbg_build.cpp:139-147)
                stage = STAGE_S1_MERCY if self.need_mercy else STAGE_S1
                for p, (mask, expected) in enumerate(self._passes(stage)):
                    if mask is not None:
                        self.e.set_bucket_filter(mask, expected, self.batch_bytes, accumulate=p > 0)
                    n1 = self._alltoall(stage, self.k, self.m)
                    r1 = self.e.dist_process_s1(self.k, self.m, n1, self.need_mercy)
                self.e.set_bucket_filter(None)
                n_words = self.world * self.stride_words
                bm = self.e.as_tensor(self.e.device_pointer(BUF_IS_SOLID), n_words * 8, self.device).view(torch.int64)
                if self.staging == "host":
                    h = bm.cpu()
                    sl = self.x.sum_bitmap_and_take_slice(h, self.stride_words).to(bm.device)
                else:
                    sl = self.x.sum_bitmap_and_take_slice(bm, self.stride_words)
                self.e.adopt_is_solid_slice(sl.data_ptr(), self.stride_words)
                self._keep = sl
                if self.need_mercy:  # candidates -> read owners; the mercy block of Read2SdbgS2::Initialize runs there
                    self._route(ROUTE_MERCY_CAND)
                    self.n_mercy = self.e.read2sdbg_add_mercy(self.k)
            r2 = None
            for p, (mask, expected) in enumerate(self._passes(STAGE_S2)):
                if mask is not None:
                    self.e.set_bucket_filter(mask, expected, self.batch_bytes)
                n2 = self._alltoall(STAGE_S2, self.k, self.m)
                r2 = self.e.dist_process_s2(self.k, n2)
                if self.on_s2_pass is not None:
                    self.on_s2_pass(p, r2)
        finally:
            if self.n_passes > 1:
                self.e.set_bucket_filter(None)
        return r1, r2
