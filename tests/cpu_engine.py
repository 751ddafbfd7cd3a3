"""TEST INFRASTRUCTURE: the phased multi-GPU interface of megahit_amd.lib.Engine, implemented on the
CPU with the oracle, so that megahit_amd/dist.py (partition, all-to-all, bitmap reduction) can be
exercised with world_size 2 on the gloo backend without GPUs.  Never used by the product."""
import ctypes as C

import numpy as np
import torch

import oracle_binding as ob

BUF_IS_SOLID = 6


class Res:
    pass


class OracleEngine:
    def __init__(self, reads, reverse=True, mult=None):
        self.pkg = ob.Package(reads, reverse=reverse)
        self.mult = mult
        self.n_bases = int(self.pkg.start()[-1])
        self.bufs = {}
        self.next_handle = 1
        self.pos_base = 0
        self.global_bases = 0
        self.local_solid = None
        self.sdbg = None
        self.keep = None
        self.accumulate = False
        self.mercy_global = np.zeros(0, dtype=np.int64)

    def _new(self, arr):
        h = self.next_handle
        self.next_handle += 1
        self.bufs[h] = arr
        return h

    def set_partition(self, my_part, n_parts, bucket_begin):
        self.my_part, self.n_parts = my_part, n_parts
        self.lut = np.zeros(65536, dtype=np.int64)
        for p in range(n_parts):
            self.lut[int(bucket_begin[p]):int(bucket_begin[p + 1])] = p
        self.bucket_begin = np.asarray(bucket_begin)

    def set_global_layout(self, pos_base, global_bases):
        self.pos_base, self.global_bases = pos_base, global_bases

    def _items(self, stage, k, m):
        if stage in (1, 5):
            return ob.s1_items(self.pkg, k, self.pos_base)
        if stage == 3:
            return ob.count_items(self.pkg, k, self.pos_base)
        if stage == 4:
            return ob.seq2sdbg_items(self.pkg, self.mult, k)
        return ob.s2_items(self.pkg, k, m, self.local_solid if m > 1 else None)

    def bucket_histogram(self, stage, k, m):
        it = self._items(stage, k, m)
        return np.bincount(it[:, 0] >> 16, minlength=65536).astype(np.uint64)

    def set_bucket_filter(self, keep, expected_items=0, batch_bytes=0, accumulate=False):
        self.keep = None if keep is None else np.asarray(keep).astype(bool)
        self.expected = expected_items
        self.accumulate = bool(accumulate)

    def dist_extract(self, stage, k, m):
        if self.keep is not None:
            items = self._items(stage, k, m)
            items = items[self.keep[items[:, 0] >> 16]]
            assert len(items) <= self.expected
        elif stage in (1, 5):
            items = ob.s1_items(self.pkg, k, self.pos_base)
        elif stage == 3:
            items = ob.count_items(self.pkg, k, self.pos_base)
        elif stage == 4:
            items = ob.seq2sdbg_items(self.pkg, self.mult, k)
        else:
            items = ob.s2_items(self.pkg, k, m, self.local_solid if m > 1 else None)
        owner = self.lut[items[:, 0] >> 16] if len(items) else np.zeros(0, dtype=np.int64)
        order = np.argsort(owner, kind="stable")
        items = np.ascontiguousarray(items[order])
        counts = np.bincount(owner, minlength=self.n_parts).astype(np.uint64)
        return self._new(items.view(np.uint8).reshape(-1)), items.shape[0], items.shape[1] * 4, counts

    def dist_recv_buffer(self, n_items, item_bytes):
        self.recv = np.zeros(n_items * item_bytes, dtype=np.uint8)
        return self._new(self.recv)

    def as_tensor(self, handle, nbytes, device):
        return torch.from_numpy(self.bufs[handle][:nbytes])

    def device_pointer(self, which):
        assert which == BUF_IS_SOLID
        return self._new(self.global_bits.view(np.uint8))

    def dist_process_s1(self, k, m, n_items, want_mercy=0):
        w = (2 * (k - 1) + 6 + 31) // 32 + 2
        items = self.recv.view(np.uint32).reshape(-1, w)[:n_items]
        assert (self.lut[items[:, 0] >> 16] == self.my_part).all()
        mercy = np.zeros(0, dtype=np.int64)
        if want_mercy:
            bits, hist, mercy = ob.s1_reduce_mercy(items, k, m, self.global_bases, tie_stable=want_mercy != 2)
        else:
            bits, hist = ob.s1_reduce(items, k, m, self.global_bases)
        if self.accumulate:  # later bucket-range pass: continue from the earlier ones
            bits = bits | self.global_bits
            hist = hist + self.hist
            mercy = np.concatenate([self.mercy_global, mercy])
        self.global_bits = np.ascontiguousarray(bits)
        self.hist = hist
        self.mercy_global = mercy
        r = Res()
        r.n_items = n_items
        return r

    def adopt_is_solid_slice(self, ptr, n_words):
        arr = np.ctypeslib.as_array((C.c_uint64 * n_words).from_address(ptr)).copy()
        self.local_solid = arr[: (self.n_bases + 63) // 64]

    def dist_process_s2(self, k, n_items):
        w = (2 * k + 4 + 31) // 32
        items = self.recv.view(np.uint32).reshape(-1, w)[:n_items]
        assert (self.lut[items[:, 0] >> 16] == self.my_part).all()
        self.sdbg = ob.sdbg_from_items(items, k, False)
        r = Res()
        r.n_items = n_items
        r.n_sdbg = int(self.sdbg["bucket_items"].sum())
        return r

    def dist_process_count(self, k, m, n_items):
        w = (2 * (k + 1) + 31) // 32 + 2
        items = self.recv.view(np.uint32).reshape(-1, w)[:n_items]
        assert (self.lut[items[:, 0] >> 16] == self.my_part).all()
        self.count = ob.count_reduce(items, k, m)
        r = Res()
        r.n_items = n_items
        r.n_edges = len(self.count["edges"])
        return r

    def dist_process_seq2sdbg(self, k, n_items):
        w = (2 * k + 20 + 31) // 32
        items = self.recv.view(np.uint32).reshape(-1, w)[:n_items]
        assert (self.lut[items[:, 0] >> 16] == self.my_part).all()
        self.sdbg = ob.sdbg_from_items(items, k, True)
        r = Res()
        r.n_items = n_items
        return r

    def dist_route_records(self, which, stride):
        rec = np.sort(self.count["events"]) if which == 1 else np.sort(self.mercy_global.astype(np.uint64))
        shift = 1 if which == 1 else 2
        owner = ((rec >> np.uint64(shift)) // np.uint64(stride)).astype(np.int64)
        counts = np.bincount(owner, minlength=self.n_parts).astype(np.uint64)
        assert len(counts) == self.n_parts
        return self._new(np.ascontiguousarray(rec).view(np.uint8).reshape(-1)), len(rec), 8, counts

    def dist_apply_routed(self, which, n):
        rec = self.recv.view(np.uint64)[:n]
        if which == 1:
            self.first_0_out, self.last_0_in = ob.count_apply_events(self.pkg, self.pos_base, rec)
        else:
            self.mercy_local = np.sort(rec.astype(np.int64) - (self.pos_base << 2))

    def read2sdbg_add_mercy(self, k):
        n, self.local_solid = ob.s2_add_mercy(self.pkg, k, self.local_solid, self.mercy_local)
        return n
