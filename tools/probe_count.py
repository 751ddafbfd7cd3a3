"""Quick GPU probe: count engine on N synthetic reads with per-kernel HIP-event timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megahit_amd import lib, synth

n_pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
t0 = time.time()
G = max(2000, n_pairs * 5)
reads = synth.gen_pe_reads(n_pairs, G, read_len=150, frag=400, err=0.005, seed=1)
reads = reads[:, ::-1]  # stored reversed
packed = synth.pack_reads_concat(reads)
print("gen %.1fs, %d reads" % (time.time() - t0, reads.shape[0]), flush=True)
e = lib.Engine(0)
t0 = time.time()
e.load_sequences(packed, reads.shape[0], 150, None)
print("load %.2fs" % (time.time() - t0), flush=True)
for it in range(3):
    e.profile(True); e.profile_reset()
    t0 = time.time()
    r = e.count(k, 2)
    dt = time.time() - t0
    st = e.profile_get(); e.profile(False)
    print("iter %d: %.3fs  items=%d distinct=%d edges=%d  -> %.1f M edges/s" % (it, dt, r.n_items, r.n_distinct, r.n_edges, r.n_items / dt / 1e6), flush=True)
tot = sum(v["ms"] for v in st.values())
for name, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-18s x%-3d %9.3f ms  %6.1f%%  %8.1f GB/s algo" % (name, v["launches"], v["ms"], 100 * v["ms"] / tot, v["bytes"] / v["ms"] / 1e6 if v["ms"] else 0))
print("kernel total %.1f ms" % tot)
