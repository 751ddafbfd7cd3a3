"""GPU probe: where the stage-1 tile kernel spends its time.  Needs the debug build
(make -C megahit_amd/csrc timing) and MHX_LIBRARY=megahit_amd/libmhx_timing.so.

    MHX_LIBRARY=$PWD/megahit_amd/libmhx_timing.so python tools/probe_phases.py [reads]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from megahit_amd import lib

PHASES = ["begin_block", "stage tile", "flags+scan", "tail", "ctx", "unit_count", "scan+base", "unit_emit", "item_final", "end_block"]


def main():
    n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000000
    packed = bench.make_reads(n_reads // 16 * 16, 0, 1)
    e = lib.Engine(0)
    e.load_sequences(packed, n_reads // 16 * 16, bench.READ_LEN, None)
    e.read2sdbg_s1(bench.K, bench.MIN_COUNT)
    L = lib.load()
    L.mhx_debug_tile_phases.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 16)()
    L.mhx_debug_tile_phases(buf, 1)
    e.profile(True)
    e.profile_reset()
    e.read2sdbg_s1(bench.K, bench.MIN_COUNT)
    st = e.profile_get()
    L.mhx_debug_tile_phases(buf, 1)
    tot = sum(buf[i] for i in range(10))
    print("s1_groups %.3f ms, s1_sample %.3f ms" % (st["s1_groups"]["ms"], st.get("s1_sample", {"ms": 0})["ms"]))
    for i, name in enumerate(PHASES):
        print("  %-12s %6.2f %%  (%d ticks)" % (name, 100.0 * buf[i] / max(tot, 1), buf[i]))
    # the bucket-streaming kernel (k_s1_stream): clocks of thread 0 of every workgroup, summed
    STREAM = ["bucket start (descriptor, first source)", "inserts up to barrier A", "next round decided, its first trip requested (+ second-read marks)", "table walk (statistics, marks, items, wipe)", "barrier B"]
    tot = sum(buf[10 + i] for i in range(5))
    for i, name in enumerate(STREAM):
        print("  stream %-38s %6.2f %%  (%d ticks)" % (name, 100.0 * buf[10 + i] / max(tot, 1), buf[10 + i]))



if __name__ == "__main__":
    main()
