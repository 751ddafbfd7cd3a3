"""GPU probe: radix sort of random records through mhx_sort_records, per-kernel HIP-event timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megahit_amd import lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 26
e = lib.Engine(0)
rng = np.random.default_rng(0)
for kw, aux in [(2, 2), (2, 0)]:
    items = rng.integers(0, 2 ** 32, size=(n, kw + aux), dtype=np.uint64).astype(np.uint32)
    e.sort_records(items.copy(), kw)  # warm-up (allocations)
    e.profile(True); e.profile_reset()
    e.sort_records(items, kw)
    st = e.profile_get(); e.profile(False)
    for name, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]):
        if v["ms"] > 0.05:
            print("  %dB %-22s x%-3d %8.3f ms/launch %8.1f GB/s algo" % ((kw + aux) * 4, name, v["launches"], v["ms"] / v["launches"], v["bytes"] / v["ms"] / 1e6))
    assert (np.diff(items[:, 0].astype(np.int64)) >= 0).all()
