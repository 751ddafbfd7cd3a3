"""ctypes binding of libmhx.so (C ABI in include/mhx.h).

There is deliberately no fallback: if the HIP library is missing or no GPU is usable every
call raises.  The oracle under oracle/ is test infrastructure and is never imported here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MHX_LIBRARY: load another build of the same library (the phase-timing debug build of tools/probe_phases.py)
LIB_PATH = os.environ.get("MHX_LIBRARY") or os.path.join(_HERE, "libmhx.so")

NUM_BUCKETS = 65536
MAX_MUL = 65535

BUF_EDGES = 1
BUF_BUCKET_COUNT = 2
BUF_FIRST_0_OUT = 3
BUF_LAST_0_IN = 4
BUF_MUL_HIST = 5
BUF_IS_SOLID = 6
BUF_MERCY_CAND = 7
BUF_SDBG_BYTES = 8
BUF_BUCKET_OFFSET = 9
BUF_BUCKET_TIPS = 10
BUF_BUCKET_LARGE = 11
BUF_SORTED_ITEMS = 12
(BUF_SDBG_W, BUF_SDBG_LAST, BUF_SDBG_TIP, BUF_SDBG_INVALID, BUF_SDBG_SMALL_MUL, BUF_SDBG_MUL, BUF_SDBG_TIP_LABELS, BUF_SDBG_PREFIX_LKT,
 BUF_SDBG_RS_W_L2, BUF_SDBG_RS_W_L1, BUF_SDBG_RS_W_SEL, BUF_SDBG_RS_LAST_L2, BUF_SDBG_RS_LAST_L1, BUF_SDBG_RS_LAST_SEL, BUF_SDBG_RS_TIP_L2,
 BUF_SDBG_RS_TIP_L1) = range(20, 36)
BUF_W_COUNT = 13
BUF_LIB_RECORDS = 40


class MhxError(RuntimeError):
    pass


class CountResult(C.Structure):
    _fields_ = [("n_items", C.c_uint64), ("n_distinct", C.c_uint64), ("n_edges", C.c_uint64),
                ("words_per_edge", C.c_uint32), ("item_words", C.c_uint32)]


class S1Result(C.Structure):
    _fields_ = [("n_items", C.c_uint64), ("n_solid", C.c_uint64), ("n_mercy_cand", C.c_uint64),
                ("item_words", C.c_uint32)]


class SdbgResult(C.Structure):
    _fields_ = [("n_items", C.c_uint64), ("n_sdbg", C.c_uint64), ("n_tips", C.c_uint64), ("n_large", C.c_uint64),
                ("sdbg_bytes", C.c_uint64), ("words_per_tip_label", C.c_uint32), ("item_words", C.c_uint32)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint32), ("total_ms", C.c_double),
                ("algo_bytes", C.c_double)]


class FastxResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_words", C.c_uint64), ("max_len", C.c_uint32), ("status", C.c_int)]


class IterateResult(C.Structure):
    _fields_ = [("n_flanks", C.c_uint64), ("n_kmers", C.c_uint64), ("n_edges", C.c_uint64), ("words_per_edge", C.c_uint32)]


class SdbgIndexInfo(C.Structure):
    _fields_ = [("n_items", C.c_uint64), ("n_tips", C.c_uint64), ("n_large", C.c_uint64), ("k", C.c_uint32),
                ("words_per_tip_label", C.c_uint32), ("use_full_mul", C.c_int), ("num_l1_w", C.c_uint64), ("num_l2_w", C.c_uint64),
                ("num_l1_bits", C.c_uint64), ("num_l2_bits", C.c_uint64), ("w_char_count", C.c_uint64 * 9),
                ("w_sel_offset", C.c_uint64 * 10), ("ones_in_last", C.c_uint64), ("ones_in_tip", C.c_uint64),
                ("last_sel_count", C.c_uint64), ("f", C.c_longlong * 6), ("rank_f", C.c_longlong * 6)]


class DistItems(C.Structure):
    _fields_ = [("d_items", C.c_void_p), ("n_items", C.c_uint64), ("item_bytes", C.c_uint32)]


STAGE_S1 = 1
STAGE_S2 = 2
STAGE_COUNT = 3
STAGE_SEQ2SDBG = 4
STAGE_S1_MERCY = 5

# every symbol include/mhx.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "mhx_last_error": (C.c_char_p, []),
    "mhx_version": (C.c_char_p, []),
    "mhx_device_count": (C.c_int, []),
    "mhx_create": (_P, [C.c_int]),
    "mhx_destroy": (None, [_P]),
    "mhx_trim": (C.c_int, [_P]),
    "mhx_synchronize": (C.c_int, [_P]),
    "mhx_set_option": (C.c_int, [_P, C.c_char_p, C.c_longlong]),
    "mhx_get_option": (C.c_longlong, [_P, C.c_char_p, C.c_longlong]),
    "mhx_load_sequences": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    "mhx_load_bin_records": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_int]),
    "mhx_append_sequences": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint32, _P, _P]),
    "mhx_load_multiplicity": (C.c_int, [_P, _P, C.c_uint64]),
    "mhx_load_edges": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_uint32]),
    "mhx_num_sequences": (C.c_uint64, [_P]),
    "mhx_fixed_length": (C.c_uint32, [_P]),
    "mhx_num_bases": (C.c_uint64, [_P]),
    "mhx_buffer_bytes": (C.c_uint64, [_P, C.c_int]),
    "mhx_fetch": (C.c_int, [_P, C.c_int, _P, C.c_uint64, C.c_uint64]),
    "mhx_count": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(CountResult)]),
    "mhx_read2sdbg_s1": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(S1Result)]),
    "mhx_read2sdbg_add_mercy": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint64)]),
    "mhx_set_is_solid": (C.c_int, [_P, _P, C.c_uint64]),
    "mhx_read2sdbg_s2": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(SdbgResult)]),
    "mhx_seq2sdbg": (C.c_int, [_P, C.c_uint32, C.POINTER(SdbgResult)]),
    "mhx_gen_mercy_edges": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    "mhx_sort_records": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_uint32]),
    "mhx_set_partition": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "mhx_set_global_layout": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "mhx_dist_extract": (C.c_int, [_P, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(DistItems), _P]),
    "mhx_dist_recv_buffer": (_P, [_P, C.c_uint64, C.c_uint32]),
    "mhx_dist_process_s1": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.POINTER(S1Result)]),
    "mhx_dist_process_s2": (C.c_int, [_P, C.c_uint32, C.c_uint64, C.POINTER(SdbgResult)]),
    "mhx_dist_process_count": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(CountResult)]),
    "mhx_dist_process_seq2sdbg": (C.c_int, [_P, C.c_uint32, C.c_uint64, C.POINTER(SdbgResult)]),
    "mhx_dist_route_records": (C.c_int, [_P, C.c_int, C.c_uint64, C.POINTER(DistItems), _P]),
    "mhx_dist_apply_routed": (C.c_int, [_P, C.c_int, C.c_uint64]),
    "mhx_device_pointer": (_P, [_P, C.c_int]),
    "mhx_adopt_is_solid_slice": (C.c_int, [_P, _P, C.c_uint64]),
    "mhx_iterate": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_uint64, _P, C.POINTER(IterateResult)]),
    "mhx_fastx_to_records": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(FastxResult)]),
    "mhx_sdbg_build_index": (C.c_int, [_P, C.c_uint32, C.POINTER(SdbgIndexInfo)]),
    "mhx_sdbg_load_bytes": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P, _P]),
    "mhx_sdbg_remove_tips": (C.c_int, [_P, C.POINTER(SdbgIndexInfo), C.c_int, _P]),
    "mhx_comm_unique_id": (C.c_int, [_P]),
    "mhx_comm_init_rank": (_P, [_P, _P, C.c_int, C.c_int]),
    "mhx_comm_local_group": (C.c_int, [C.c_int, _P, _P]),
    "mhx_comm_destroy": (None, [_P]),
    "mhx_comm_rank": (C.c_int, [_P]),
    "mhx_comm_size": (C.c_int, [_P]),
    "mhx_comm_barrier": (C.c_int, [_P]),
    "mhx_comm_bytes_sent": (C.c_uint64, [_P, C.c_int]),
    "mhx_comm_all_reduce_u64": (C.c_int, [_P, _P, C.c_uint64, C.c_int]),
    "mhx_dist_setup": (C.c_int, [_P, _P, C.c_int, C.c_uint32, C.c_uint32]),
    "mhx_dist_read2sdbg": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(S1Result), C.POINTER(SdbgResult), _P]),
    "mhx_dist_count": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.POINTER(CountResult)]),
    "mhx_dist_seq2sdbg": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(SdbgResult)]),
    "mhx_dist_gen_mercy_edges": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint64, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    "mhx_device_free_bytes": (C.c_uint64, [_P]),
    "mhx_bucket_histogram": (C.c_int, [_P, C.c_int, C.c_uint32, C.c_uint32, _P]),
    "mhx_set_bucket_filter": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_int]),
    "mhx_stage_pass_bytes": (C.c_uint64, [_P, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64]),
    "mhx_alloc_stats": (None, [_P, _P, _P, _P]),
    "mhx_comm_init_hosted": (_P, [_P, C.c_int, C.c_int, _P]),
    "mhx_reset": (C.c_int, [_P]),
    "mhx_last_s1_plan": (C.c_char_p, [_P]),
    "mhx_s1_self_planned": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_int]),
    "mhx_count_self_planned": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "mhx_profile_enable": (C.c_int, [_P, C.c_int]),
    "mhx_profile_reset": (C.c_int, [_P]),
    "mhx_profile_get": (C.c_int, [_P, C.POINTER(KernelStat), C.c_int]),
}

_lib = None


def load():
    """Load libmhx.so (no compute).  Raises MhxError if the HIP extension was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MhxError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    """One GPU, one HIP stream: mirrors the reference's engine objects (KmerCounter, Read2SdbgS1/S2,
    SeqToSdbg — reference src/sorting/*.h) on top of the C ABI."""

    def __init__(self, device=0):
        self.lib = load()
        self.h = self.lib.mhx_create(device)
        if not self.h:
            raise MhxError(self.lib.mhx_last_error().decode())

    def close(self):
        if self.h:
            self.lib.mhx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MhxError(self.lib.mhx_last_error().decode())

    # ---- inputs
    def load_sequences(self, packed, n_seqs, fixed_len=0, start_pos=None):
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        if start_pos is not None:
            start_pos = np.ascontiguousarray(start_pos, dtype=np.uint64)
        self._keep = (packed, start_pos)
        self._chk(self.lib.mhx_load_sequences(self.h, _ptr(packed), packed.size, n_seqs, fixed_len, _ptr(start_pos)))

    def load_bin_records(self, records, n_seqs, reverse=True):
        records = np.ascontiguousarray(records, dtype=np.uint32)
        self._chk(self.lib.mhx_load_bin_records(self.h, _ptr(records), records.size, n_seqs, int(reverse)))

    def append_sequences(self, packed, n_seqs, fixed_len=0, start_pos=None, mult=None):
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        if start_pos is not None:
            start_pos = np.ascontiguousarray(start_pos, dtype=np.uint64)
        if mult is not None:
            mult = np.ascontiguousarray(mult, dtype=np.uint16)
        self._chk(self.lib.mhx_append_sequences(self.h, _ptr(packed), packed.size, n_seqs, fixed_len, _ptr(start_pos), _ptr(mult)))

    def load_multiplicity(self, mult):
        mult = np.ascontiguousarray(mult, dtype=np.uint16)
        self._chk(self.lib.mhx_load_multiplicity(self.h, _ptr(mult), mult.size))

    def set_is_solid(self, bits):
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        self._chk(self.lib.mhx_set_is_solid(self.h, _ptr(bits), bits.size))

    @property
    def n_seqs(self):
        return self.lib.mhx_num_sequences(self.h)

    @property
    def n_bases(self):
        return self.lib.mhx_num_bases(self.h)

    # ---- engines
    def count(self, k, m):
        r = CountResult()
        self._chk(self.lib.mhx_count(self.h, k, m, C.byref(r)))
        return r

    def read2sdbg_s1(self, k, m, want_mercy=False):
        """want_mercy: False/0 none, True/1 stable tie order, 2 reference-exact (kmsort) tie order."""
        r = S1Result()
        self._chk(self.lib.mhx_read2sdbg_s1(self.h, k, m, int(want_mercy), C.byref(r)))
        return r

    def read2sdbg_add_mercy(self, k):
        n = C.c_uint64(0)
        self._chk(self.lib.mhx_read2sdbg_add_mercy(self.h, k, C.byref(n)))
        return n.value

    def read2sdbg_s2(self, k, m):
        r = SdbgResult()
        self._chk(self.lib.mhx_read2sdbg_s2(self.h, k, m, C.byref(r)))
        return r

    def seq2sdbg(self, k):
        r = SdbgResult()
        self._chk(self.lib.mhx_seq2sdbg(self.h, k, C.byref(r)))
        return r

    def gen_mercy_edges(self, k, cand_packed, n_cand, cand_start):
        cand_packed = np.ascontiguousarray(cand_packed, dtype=np.uint32)
        cand_start = np.ascontiguousarray(cand_start, dtype=np.uint64)
        n = C.c_uint64(0)
        self._chk(self.lib.mhx_gen_mercy_edges(self.h, k, _ptr(cand_packed), cand_packed.size, n_cand, _ptr(cand_start),
                                               C.byref(n)))
        return n.value

    def sort_records(self, items, key_words):
        """items: uint32 [n, w]; sorted in place by the first key_words words."""
        assert items.dtype == np.uint32 and items.flags.c_contiguous and items.ndim == 2
        self._chk(self.lib.mhx_sort_records(self.h, _ptr(items), items.shape[0], key_words, items.shape[1] - key_words))
        return items

    # ---- multi-GPU phases, one call each (the C++ drivers of comm.hip compose them: lib.Comm)
    def set_partition(self, my_part, n_parts, bucket_begin):
        bb = np.ascontiguousarray(bucket_begin, dtype=np.uint32)
        self._chk(self.lib.mhx_set_partition(self.h, my_part, n_parts, _ptr(bb)))
        self.n_parts = n_parts

    def set_global_layout(self, pos_base, global_bases):
        self._chk(self.lib.mhx_set_global_layout(self.h, pos_base, global_bases))

    def dist_extract(self, stage, k, m):
        """-> (device pointer, n_items, item_bytes, counts per owner)"""
        out = DistItems()
        counts = np.zeros(self.n_parts, dtype=np.uint64)
        self._chk(self.lib.mhx_dist_extract(self.h, stage, k, m, C.byref(out), _ptr(counts)))
        return out.d_items, out.n_items, out.item_bytes, counts

    def dist_recv_buffer(self, n_items, item_bytes):
        p = self.lib.mhx_dist_recv_buffer(self.h, n_items, item_bytes)
        if not p:
            raise MhxError(self.lib.mhx_last_error().decode())
        return p

    def dist_process_s1(self, k, m, n_items, want_mercy=0):
        r = S1Result()
        self._chk(self.lib.mhx_dist_process_s1(self.h, k, m, int(want_mercy), n_items, C.byref(r)))
        return r

    def dist_process_count(self, k, m, n_items):
        r = CountResult()
        self._chk(self.lib.mhx_dist_process_count(self.h, k, m, n_items, C.byref(r)))
        return r

    def dist_process_seq2sdbg(self, k, n_items):
        r = SdbgResult()
        self._chk(self.lib.mhx_dist_process_seq2sdbg(self.h, k, n_items, C.byref(r)))
        return r

    def dist_route_records(self, which, stride_bases):
        """-> (device pointer, n_records, 8, counts per read-owning rank)"""
        out = DistItems()
        counts = np.zeros(self.n_parts, dtype=np.uint64)
        self._chk(self.lib.mhx_dist_route_records(self.h, which, stride_bases, C.byref(out), _ptr(counts)))
        return out.d_items, out.n_items, out.item_bytes, counts

    def dist_apply_routed(self, which, n_records):
        self._chk(self.lib.mhx_dist_apply_routed(self.h, which, n_records))

    def dist_process_s2(self, k, n_items):
        r = SdbgResult()
        self._chk(self.lib.mhx_dist_process_s2(self.h, k, n_items, C.byref(r)))
        return r

    def device_pointer(self, which):
        return self.lib.mhx_device_pointer(self.h, which)

    def adopt_is_solid_slice(self, ptr, n_words):
        self._chk(self.lib.mhx_adopt_is_solid_slice(self.h, ptr, n_words))

    # ---- memory-bounded passes (see megahit_amd/passes.py)
    def bucket_histogram(self, stage, k, m):
        h = np.zeros(NUM_BUCKETS, dtype=np.uint64)
        self._chk(self.lib.mhx_bucket_histogram(self.h, stage, k, m, _ptr(h)))
        return h

    def set_bucket_filter(self, keep, expected_items=0, batch_bytes=0, accumulate=False):
        if keep is None:
            self._chk(self.lib.mhx_set_bucket_filter(self.h, None, 0, batch_bytes, int(accumulate)))
            return
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        assert keep.size == NUM_BUCKETS
        self._chk(self.lib.mhx_set_bucket_filter(self.h, _ptr(keep), int(expected_items), int(batch_bytes), int(accumulate)))

    # ---- outputs
    def fetch(self, which, dtype):
        nbytes = self.lib.mhx_buffer_bytes(self.h, which)
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        if nbytes:
            self._chk(self.lib.mhx_fetch(self.h, which, _ptr(out), 0, nbytes))
        return out

    def synchronize(self):
        self._chk(self.lib.mhx_synchronize(self.h))

    def trim(self):
        self._chk(self.lib.mhx_trim(self.h))

    def fastx_to_records(self, text1, text2=None):
        """FASTA/FASTQ text -> read-library records on the GPU; returns (FastxResult, uint32 records or None)."""
        r = FastxResult()
        self._chk(self.lib.mhx_fastx_to_records(self.h, text1, len(text1), text2, len(text2) if text2 is not None else 0, C.byref(r)))
        return r, (self.fetch(BUF_LIB_RECORDS, np.uint32) if r.status == 0 else None)

    def sdbg_build_index(self, k):
        """W/last/tip/mul arrays + rank/select tables on the device (include/mhx.h: mhx_sdbg_build_index)."""
        info = SdbgIndexInfo()
        self._chk(self.lib.mhx_sdbg_build_index(self.h, k, C.byref(info)))
        return info

    def sdbg_remove_tips(self, info, max_tip_len):
        """sdbg_pruning::RemoveTips on the device-resident graph; returns the number of tips removed."""
        n = C.c_uint64(0)
        self._chk(self.lib.mhx_sdbg_remove_tips(self.h, C.byref(info), int(max_tip_len), C.byref(n)))
        return int(n.value)

    def sdbg_load_bytes(self, data, offset, items, tips, large):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        tabs = [np.ascontiguousarray(t, dtype=np.uint64) for t in (offset, items, tips, large)]
        self._chk(self.lib.mhx_sdbg_load_bytes(self.h, _ptr(data), data.size, *[_ptr(t) for t in tabs]))

    def set_option(self, name, value):
        """Tuning / diagnostic knob of this handle (include/mhx.h: mhx_set_option)."""
        self._chk(self.lib.mhx_set_option(self.h, name.encode(), int(value)))

    def s1_self_planned(self, k, min_count, want_mercy=False):
        """stage 1 would run on super-k-mer records and cut the job into passes by itself (mhx_s1_self_planned)"""
        return bool(self.lib.mhx_s1_self_planned(self.h, k, min_count, int(want_mercy)))

    def count_self_planned(self, k, min_count):
        """count would run on super-k-mer records and cut the job into passes by itself (mhx_count_self_planned)"""
        return bool(self.lib.mhx_count_self_planned(self.h, k, min_count))

    def last_s1_plan(self):
        """what the last stage 1 of this handle ran as (mhx_last_s1_plan)"""
        return (self.lib.mhx_last_s1_plan(self.h) or b"").decode()

    def get_option(self, name, default):
        """Effective value of a knob for this handle (include/mhx.h: mhx_get_option): explicit option, environment, tuned default."""
        return int(self.lib.mhx_get_option(self.h, name.encode(), int(default)))

    # ---- profiling
    def profile(self, on=True):
        self._chk(self.lib.mhx_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._chk(self.lib.mhx_profile_reset(self.h))

    def profile_get(self):
        arr = (KernelStat * 128)()
        n = self.lib.mhx_profile_get(self.h, arr, 128)
        if n < 0:
            raise MhxError(self.lib.mhx_last_error().decode())
        return {arr[i].name.decode(): dict(launches=arr[i].launches, ms=arr[i].total_ms, bytes=arr[i].algo_bytes)
                for i in range(min(n, 128))}


COMM_ID_BYTES = 128


def comm_unique_id():
    """RCCL unique id (bytes) created by one rank and shipped to the others (mhx_comm_unique_id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    if load().mhx_comm_unique_id(buf) != 0:
        raise MhxError(load().mhx_last_error().decode())
    return buf.raw


class Comm:
    """Communicator of the C++ multi-GPU drivers (include/mhx.h: mhx_comm_*).  One per rank, bound to the rank's Engine."""

    def __init__(self, engine, handle):
        self.lib, self.e, self.h = engine.lib, engine, handle

    @classmethod
    def rccl(cls, engine, unique_id, rank, n_ranks):
        h = engine.lib.mhx_comm_init_rank(engine.h, unique_id, rank, n_ranks)
        if not h:
            raise MhxError(engine.lib.mhx_last_error().decode())
        return cls(engine, h)

    @classmethod
    def hosted(cls, engine, dist, rank, n_ranks):
        """The caller's torch.distributed group (any backend that moves CPU tensors, e.g. gloo) moves the bytes through host
        memory (megahit_amd/hosted.py): rank processes that cannot share an RCCL world, e.g. several on one GPU."""
        from . import hosted as H
        t, keep = H.make_transport(dist, rank, n_ranks)
        engine.lib.mhx_comm_init_hosted.restype = C.c_void_p
        engine.lib.mhx_comm_init_hosted.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(H.HostTransport)]
        h = engine.lib.mhx_comm_init_hosted(engine.h, rank, n_ranks, C.byref(t))
        if not h:
            raise MhxError(engine.lib.mhx_last_error().decode())
        cm = cls(engine, h)
        cm._keep = (t, keep)  # the callbacks must outlive the communicator
        return cm

    @classmethod
    def local_group(cls, engines):
        """In-process group (no RCCL): the ranks are threads of this process and may share GPUs."""
        n = len(engines)
        ctxs = (C.c_void_p * n)(*[e.h for e in engines])
        out = (C.c_void_p * n)()
        if engines[0].lib.mhx_comm_local_group(n, ctxs, out) != 0:
            raise MhxError(engines[0].lib.mhx_last_error().decode())
        return [cls(e, out[i]) for i, e in enumerate(engines)]

    def close(self):
        if self.h:
            self.lib.mhx_comm_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise MhxError(self.lib.mhx_last_error().decode())

    def barrier(self):
        self._chk(self.lib.mhx_comm_barrier(self.h))

    def bytes_sent(self, reset=False):
        """payload bytes this rank has handed to other ranks through the item / record exchanges since the last reset"""
        return int(self.lib.mhx_comm_bytes_sent(self.h, int(reset)))

    def all_reduce(self, values, is_max=False):
        v = np.ascontiguousarray(values, dtype=np.uint64).copy()
        self._chk(self.lib.mhx_comm_all_reduce_u64(self.h, _ptr(v), v.size, int(is_max)))
        return v

    def setup(self, balance_stage=0, k=21, m=2):
        self._chk(self.lib.mhx_dist_setup(self.e.h, self.h, balance_stage, k, m))

    def read2sdbg(self, k, m, need_mercy=0):
        r1, r2, nm = S1Result(), SdbgResult(), C.c_uint64(0)
        self._chk(self.lib.mhx_dist_read2sdbg(self.e.h, self.h, k, m, int(need_mercy), C.byref(r1), C.byref(r2), C.byref(nm)))
        return r1, r2, int(nm.value)

    def count(self, k, m):
        r = CountResult()
        self._chk(self.lib.mhx_dist_count(self.e.h, self.h, k, m, C.byref(r)))
        return r

    def seq2sdbg(self, k):
        r = SdbgResult()
        self._chk(self.lib.mhx_dist_seq2sdbg(self.e.h, self.h, k, C.byref(r)))
        return r

    def gen_mercy_edges(self, k, cand_packed, n_cand, cand_start):
        """collective: every rank passes ALL candidate reads; returns the number of all mercy edges"""
        cand_packed = np.ascontiguousarray(cand_packed, dtype=np.uint32)
        cand_start = np.ascontiguousarray(cand_start, dtype=np.uint64)
        n = C.c_uint64(0)
        self._chk(self.lib.mhx_dist_gen_mercy_edges(self.e.h, self.h, k, _ptr(cand_packed), cand_packed.size, n_cand, _ptr(cand_start), C.byref(n)))
        return n.value
