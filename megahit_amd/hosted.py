"""Byte movers for the hosted communicator of libmhx (include/mhx.h: mhx_host_transport, mhx_comm_init_hosted): the ranks
are processes of a torch.distributed group on any backend that moves CPU tensors (gloo), libmhx stages the per-peer
segments in host memory and calls back into these two functions.  Used where RCCL cannot be: several rank PROCESSES on
one GPU (tests/test_gpu_multiprocess.py), hosts without a librccl.  Control plane = data plane here; with RCCL
(lib.Comm.rccl) torch.distributed only carries the unique id and the barriers."""
import ctypes as C

import numpy as np


def all_reduce_u64(dist, ptr, n, is_max):
    """in place over all ranks: element-wise sum (wrapping, as unsigned) or max (of the values read as signed 64-bit)"""
    import torch
    if n == 0:
        return 0
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(int(n),))
    t = torch.from_numpy(a)  # shares the memory
    dist.all_reduce(t, op=dist.ReduceOp.MAX if is_max else dist.ReduceOp.SUM)
    return 0


def all_to_all_bytes(dist, rank, world, send_ptr, send_bytes, recv_ptr, recv_bytes):
    """segments for / from rank 0, 1, ... back to back; gloo has no all_to_all: point-to-point pairs, all posted at once"""
    import torch
    sb = [int(send_bytes[p]) for p in range(world)]
    rb = [int(recv_bytes[p]) for p in range(world)]
    send = np.ctypeslib.as_array(C.cast(send_ptr, C.POINTER(C.c_uint8)), shape=(max(1, sum(sb)),)) if sum(sb) else None
    recv = np.ctypeslib.as_array(C.cast(recv_ptr, C.POINTER(C.c_uint8)), shape=(max(1, sum(rb)),)) if sum(rb) else None
    ops, so, ro = [], 0, 0
    keep = []
    for p in range(world):
        if p != rank and rb[p]:
            t = torch.from_numpy(recv[ro:ro + rb[p]])
            keep.append(t)
            ops.append(dist.P2POp(dist.irecv, t, p))
        ro += rb[p]
    for p in range(world):
        if p != rank and sb[p]:
            t = torch.from_numpy(send[so:so + sb[p]])
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, p))
        so += sb[p]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return 0


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int)
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))


class HostTransport(C.Structure):
    _fields_ = [("all_reduce_u64", REDUCE_FN), ("all_to_all_bytes", A2A_FN), ("user", C.c_void_p)]


def make_transport(dist, rank, world):
    """-> (HostTransport, keep-alive objects): the callbacks of a torch.distributed process group"""
    def red(_user, ptr, n, is_max):
        try:
            return all_reduce_u64(dist, ptr, n, is_max)
        except Exception as ex:  # noqa: BLE001 - an exception must not cross the C boundary
            print("hosted all_reduce failed:", ex, flush=True)
            return 1

    def a2a(_user, send, send_bytes, recv, recv_bytes):
        try:
            return all_to_all_bytes(dist, rank, world, send, send_bytes, recv, recv_bytes)
        except Exception as ex:  # noqa: BLE001
            print("hosted all_to_all failed:", ex, flush=True)
            return 1

    r, a = REDUCE_FN(red), A2A_FN(a2a)
    return HostTransport(r, a, None), (r, a)
