import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megahit_amd import lib, synth
n_pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3000
reads = synth.gen_pe_reads(n_pairs, max(4000, n_pairs * 5), read_len=100, frag=250, err=0.01, seed=11)[:, ::-1]
packed = synth.pack_reads_concat(reads)
e = lib.Engine(0)
e.load_sequences(packed, reads.shape[0], 100, None)
for mode in (1, 2, 2):
    e.profile(True); e.profile_reset()
    t0 = time.time(); r = e.read2sdbg_s1(21, 2, want_mercy=mode); dt = time.time() - t0
    st = e.profile_get(); e.profile(False)
    print("mode %d: %.3f s, %d items, %d cands" % (mode, dt, r.n_items, r.n_mercy_cand))
    for name, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])[:5]:
        print("    %-20s x%-3d %10.3f ms" % (name, v["launches"], v["ms"]))
