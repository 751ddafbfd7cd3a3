"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
NB = 65536


class Pkg(C.Structure):
    _fields_ = [("words", C.POINTER(C.c_uint32)), ("n_words_cap", C.c_uint64), ("start", C.POINTER(C.c_uint64)),
                ("n_seqs", C.c_uint64), ("seq_cap", C.c_uint64)]


class Vec(C.Structure):
    _fields_ = [("d", C.POINTER(C.c_uint32)), ("n", C.c_uint64), ("cap", C.c_uint64), ("w", C.c_int)]


class CountOut(C.Structure):
    _fields_ = [("words_per_edge", C.c_int), ("edges", Vec), ("bucket_count", C.c_int64 * NB),
                ("first_0_out", C.POINTER(C.c_uint32)), ("last_0_in", C.POINTER(C.c_uint32)),
                ("hist", C.c_int64 * 65536), ("n_items", C.c_int64)]


class S1Out(C.Structure):
    _fields_ = [("is_solid", C.POINTER(C.c_uint64)), ("n_bits", C.c_uint64), ("hist", C.c_int64 * 65536),
                ("mercy", C.POINTER(C.c_int64)), ("n_mercy", C.c_uint64), ("mercy_cap", C.c_uint64),
                ("n_items", C.c_int64)]


class SdbgOut(C.Structure):
    _fields_ = [("k", C.c_int), ("words_per_tip_label", C.c_int), ("bytes", C.POINTER(C.c_uint8)),
                ("n_bytes", C.c_uint64), ("cap", C.c_uint64), ("bucket_off", C.c_uint64 * NB),
                ("bucket_items", C.c_uint64 * NB), ("bucket_tips", C.c_uint64 * NB), ("bucket_large", C.c_uint64 * NB),
                ("w_count", C.c_uint64 * 9), ("ones_in_last", C.c_uint64), ("n_sort_items", C.c_int64)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_pkg_init.argtypes = [C.POINTER(Pkg)]
        L.orc_pkg_free.argtypes = [C.POINTER(Pkg)]
        L.orc_pkg_append_packed.argtypes = [C.POINTER(Pkg), C.c_void_p, C.c_uint32, C.c_int]
        L.orc_count.argtypes = [C.POINTER(Pkg), C.c_int, C.c_int, C.POINTER(CountOut)]
        L.orc_count_free.argtypes = [C.POINTER(CountOut)]
        L.orc_s1.argtypes = [C.POINTER(Pkg), C.c_int, C.c_int, C.c_int, C.POINTER(S1Out)]
        L.orc_s1_free.argtypes = [C.POINTER(S1Out)]
        L.orc_s2_add_mercy.argtypes = [C.POINTER(Pkg), C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_s2_add_mercy.restype = C.c_int64
        L.orc_s2.argtypes = [C.POINTER(Pkg), C.c_int, C.c_int, C.c_void_p, C.POINTER(SdbgOut)]
        L.orc_seq2sdbg.argtypes = [C.POINTER(Pkg), C.c_void_p, C.c_int, C.POINTER(SdbgOut)]
        L.orc_sdbg_free.argtypes = [C.POINTER(SdbgOut)]
        L.orc_s1_items.argtypes = [C.POINTER(Pkg), C.c_int, C.c_uint64, C.POINTER(Vec)]
        L.orc_s1_reduce.argtypes = [C.c_void_p, C.POINTER(Vec), C.c_int, C.c_int, C.c_int, C.POINTER(S1Out)]
        L.orc_s2_items.argtypes = [C.POINTER(Pkg), C.c_int, C.c_int, C.c_void_p, C.POINTER(Vec)]
        L.orc_sdbg_from_items.argtypes = [C.POINTER(Vec), C.c_int, C.c_int, C.POINTER(SdbgOut)]
        L.orc_gen_mercy_edges.argtypes = [C.POINTER(Pkg), C.POINTER(C.POINTER(C.c_uint16)), C.POINTER(C.c_uint64),
                                          C.POINTER(Pkg), C.c_int]
        L.orc_gen_mercy_edges.restype = C.c_int64
        L.orc_sort_items.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int]
        L.orc_s1_reduce_ex.argtypes = [C.c_void_p, C.POINTER(Vec), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(S1Out)]
        L.orc_count_items.argtypes = [C.POINTER(Pkg), C.c_int, C.c_uint64, C.POINTER(Vec)]
        L.orc_count_reduce.argtypes = [C.POINTER(Vec), C.c_int, C.c_int, C.POINTER(CountOut), C.POINTER(C.POINTER(C.c_uint64)),
                                       C.POINTER(C.c_uint64)]
        L.orc_count_apply_events.argtypes = [C.POINTER(Pkg), C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_seq2sdbg_items.argtypes = [C.POINTER(Pkg), C.c_void_p, C.c_int, C.POINTER(Vec)]
        _lib = L
    return _lib


class Package:
    """A packed sequence set owned by the oracle library."""

    def __init__(self, seqs=None, reverse=False):
        self.L = lib()
        self.p = Pkg()
        self.L.orc_pkg_init(C.byref(self.p))
        if seqs is not None:
            from megahit_amd.synth import pack_reads
            for s in seqs:
                s = np.asarray(s, dtype=np.uint8)
                if len(s):
                    w = np.ascontiguousarray(pack_reads(s[None, :])[0])
                else:
                    w = np.zeros(1, dtype=np.uint32)
                self.L.orc_pkg_append_packed(C.byref(self.p), w.ctypes.data, len(s), int(reverse))

    @property
    def n_seqs(self):
        return self.p.n_seqs

    def start(self):
        return np.ctypeslib.as_array(self.p.start, shape=(self.p.n_seqs + 1,)).copy()

    def words(self):
        nb = int(self.start()[-1])
        return np.ctypeslib.as_array(self.p.words, shape=((nb + 15) // 16 + 1,)).copy()

    def __del__(self):
        try:
            self.L.orc_pkg_free(C.byref(self.p))
        except Exception:
            pass


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


def count(pkg, k, m):
    L = lib()
    o = CountOut()
    L.orc_count(C.byref(pkg.p), k, m, C.byref(o))
    res = dict(
        wpe=o.words_per_edge, n_items=o.n_items,
        edges=_arr(o.edges.d, o.edges.n * o.words_per_edge, np.uint32).reshape(-1, o.words_per_edge),
        bucket_count=np.array(o.bucket_count, dtype=np.uint64),
        first_0_out=_arr(o.first_0_out, pkg.n_seqs, np.uint32), last_0_in=_arr(o.last_0_in, pkg.n_seqs, np.uint32),
        hist=np.array(o.hist, dtype=np.int64))
    L.orc_count_free(C.byref(o))
    return res


def s1(pkg, k, m, tie_stable=True):
    L = lib()
    o = S1Out()
    L.orc_s1(C.byref(pkg.p), k, m, 0 if tie_stable else 1, C.byref(o))
    res = dict(n_items=o.n_items, is_solid=_arr(o.is_solid, (o.n_bits + 63) // 64, np.uint64),
               hist=np.array(o.hist, dtype=np.int64), mercy=_arr(o.mercy, o.n_mercy, np.int64))
    L.orc_s1_free(C.byref(o))
    return res


def s2_add_mercy(pkg, k, is_solid, cands):
    L = lib()
    is_solid = np.ascontiguousarray(is_solid, dtype=np.uint64).copy()
    buf = np.concatenate([is_solid, np.zeros(2, dtype=np.uint64)])
    cands = np.ascontiguousarray(cands, dtype=np.int64)
    n = L.orc_s2_add_mercy(C.byref(pkg.p), k, buf.ctypes.data, cands.ctypes.data, cands.size)
    return n, buf[:is_solid.size]


def _sdbg(o):
    res = dict(k=o.k, wpt=o.words_per_tip_label, bytes=_arr(o.bytes, o.n_bytes, np.uint8),
               bucket_off=np.array(o.bucket_off, dtype=np.uint64), bucket_items=np.array(o.bucket_items, dtype=np.uint64),
               bucket_tips=np.array(o.bucket_tips, dtype=np.uint64), bucket_large=np.array(o.bucket_large, dtype=np.uint64),
               w_count=np.array(o.w_count, dtype=np.uint64), ones_in_last=o.ones_in_last, n_sort_items=o.n_sort_items)
    lib().orc_sdbg_free(C.byref(o))
    return res


def s2(pkg, k, m, is_solid):
    o = SdbgOut()
    p = None
    if is_solid is not None:
        buf = np.concatenate([np.ascontiguousarray(is_solid, dtype=np.uint64), np.zeros(2, dtype=np.uint64)])
        p = buf.ctypes.data
    lib().orc_s2(C.byref(pkg.p), k, m, p, C.byref(o))
    return _sdbg(o)


def seq2sdbg(pkg, mult, k):
    o = SdbgOut()
    mult = np.ascontiguousarray(mult, dtype=np.uint16)
    lib().orc_seq2sdbg(C.byref(pkg.p), mult.ctypes.data, k, C.byref(o))
    return _sdbg(o)


def sort_items(items, key_words, kmsort=False):
    items = np.ascontiguousarray(items, dtype=np.uint32).copy()
    lib().orc_sort_items(items.ctypes.data, items.shape[0], items.shape[1], key_words, 1 if kmsort else 0)
    return items


def gen_mercy_edges(edge_pkg, mult, cand_pkg, k):
    """Appends mercy edges to edge_pkg (in place); returns (n_mercy, new multiplicity array)."""
    L = lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    mult = np.ascontiguousarray(mult, dtype=np.uint16)
    raw = libc.malloc(max(2, mult.size * 2))
    C.memmove(raw, mult.ctypes.data, mult.size * 2)
    mp = C.cast(raw, C.POINTER(C.c_uint16))
    nm = C.c_uint64(mult.size)
    n = L.orc_gen_mercy_edges(C.byref(edge_pkg.p), C.byref(mp), C.byref(nm), C.byref(cand_pkg.p), k)
    out = np.ctypeslib.as_array(mp, shape=(nm.value,)).copy() if nm.value else np.zeros(0, dtype=np.uint16)
    libc.free.argtypes = [C.c_void_p]
    libc.free(C.cast(mp, C.c_void_p))
    return n, out


def _take_vec(v):
    out = _arr(v.d, v.n * v.w, np.uint32).reshape(-1, v.w) if v.n else np.zeros((0, v.w), dtype=np.uint32)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(C.cast(v.d, C.c_void_p))
    return out


def s1_items(pkg, k, pos_base=0):
    v = Vec()
    lib().orc_s1_items(C.byref(pkg.p), k, pos_base, C.byref(v))
    return _take_vec(v)


def s2_items(pkg, k, m, is_solid):
    v = Vec()
    p = None
    if is_solid is not None:
        buf = np.concatenate([np.ascontiguousarray(is_solid, dtype=np.uint64), np.zeros(2, dtype=np.uint64)])
        p = buf.ctypes.data
    lib().orc_s2_items(C.byref(pkg.p), k, m, p, C.byref(v))
    return _take_vec(v)


def _malloc_copy(arr):
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    raw = libc.malloc(max(8, arr.nbytes))
    C.memmove(raw, arr.ctypes.data, arr.nbytes)
    return raw


def count_items(pkg, k, pos_base=0):
    v = Vec()
    lib().orc_count_items(C.byref(pkg.p), k, pos_base, C.byref(v))
    return _take_vec(v)


def count_reduce(items, k, m):
    """items: uint32 [n, W+2] -> dict(edges, bucket_count, hist, events uint64[(pos << 1) | which])."""
    items = np.ascontiguousarray(items, dtype=np.uint32)
    raw = _malloc_copy(items)
    v = Vec(C.cast(raw, C.POINTER(C.c_uint32)), items.shape[0], items.shape[0], items.shape[1])
    o = CountOut()
    ev = C.POINTER(C.c_uint64)()
    n_ev = C.c_uint64(0)
    lib().orc_count_reduce(C.byref(v), k, m, C.byref(o), C.byref(ev), C.byref(n_ev))  # frees raw
    res = dict(wpe=o.words_per_edge, n_items=o.n_items,
               edges=_arr(o.edges.d, o.edges.n * o.words_per_edge, np.uint32).reshape(-1, o.words_per_edge),
               bucket_count=np.array(o.bucket_count, dtype=np.uint64), hist=np.array(o.hist, dtype=np.int64),
               events=_arr(ev, n_ev.value, np.uint64))
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(C.cast(ev, C.c_void_p))
    libc.free(C.cast(o.edges.d, C.c_void_p))
    return res


def count_apply_events(pkg, pos_base, events):
    n = pkg.n_seqs
    first = np.full(max(n, 1), 0xFFFFFFFF, dtype=np.uint32)
    last = np.full(max(n, 1), 0xFFFFFFFF, dtype=np.uint32)
    events = np.ascontiguousarray(events, dtype=np.uint64)
    lib().orc_count_apply_events(C.byref(pkg.p), pos_base, events.ctypes.data, events.size, first.ctypes.data, last.ctypes.data)
    return first[:n], last[:n]


def seq2sdbg_items(pkg, mult, k):
    v = Vec()
    mult = np.ascontiguousarray(mult, dtype=np.uint16)
    lib().orc_seq2sdbg_items(C.byref(pkg.p), mult.ctypes.data, k, C.byref(v))
    return _take_vec(v)


def s1_reduce_mercy(items, k, m, n_bits, tie_stable=True):
    """As s1_reduce, plus the mercy candidates (global positions) computed without the reads."""
    items = np.ascontiguousarray(items, dtype=np.uint32)
    v = Vec(items.ctypes.data_as(C.POINTER(C.c_uint32)), items.shape[0], items.shape[0], items.shape[1])
    o = S1Out()
    nw = (n_bits + 63) // 64
    bits = np.zeros(nw + 2, dtype=np.uint64)
    o.is_solid = bits.ctypes.data_as(C.POINTER(C.c_uint64))
    o.n_bits = n_bits
    lib().orc_s1_reduce_ex(None, C.byref(v), k, m, 0 if tie_stable else 1, 1, C.byref(o))
    mercy = _arr(o.mercy, o.n_mercy, np.int64)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(C.cast(o.mercy, C.c_void_p))
    return bits[:nw], np.array(o.hist, dtype=np.int64), mercy


def s1_reduce(items, k, m, n_bits, tie_stable=True):
    """items: uint32 [n, W+2] (any order) -> (is_solid uint64[ceil(n_bits/64)], hist)."""
    items = np.ascontiguousarray(items, dtype=np.uint32)
    v = Vec(items.ctypes.data_as(C.POINTER(C.c_uint32)), items.shape[0], items.shape[0], items.shape[1])
    o = S1Out()
    nw = (n_bits + 63) // 64
    bits = np.zeros(nw + 2, dtype=np.uint64)
    o.is_solid = bits.ctypes.data_as(C.POINTER(C.c_uint64))
    o.n_bits = n_bits
    lib().orc_s1_reduce(None, C.byref(v), k, m, 0 if tie_stable else 1, C.byref(o))
    return bits[:nw], np.array(o.hist, dtype=np.int64)


def sdbg_from_items(items, k, is_seq2sdbg=False):
    items = np.ascontiguousarray(items, dtype=np.uint32)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    raw = libc.malloc(max(8, items.nbytes))
    C.memmove(raw, items.ctypes.data, items.nbytes)
    v = Vec(C.cast(raw, C.POINTER(C.c_uint32)), items.shape[0], items.shape[0], items.shape[1])
    o = SdbgOut()
    lib().orc_sdbg_from_items(C.byref(v), k, int(is_seq2sdbg), C.byref(o))  # frees raw
    return _sdbg(o)
