"""GPU probe: ms per chained-scan pass for 8-, 12- and 16-byte records at the headline's record count — is a scatter pass's time a
matter of bytes written or of runs written?  (round 6: the SQ counters show the 12-byte passes waiting, not issuing)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megahit_amd import lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 29
e = lib.Engine(0)
rng = np.random.default_rng(0)
out = {"records": n, "passes": {}}
for kw, aux in [(2, 0), (2, 1), (2, 2)]:
    items = rng.integers(0, 2 ** 32, size=(n, kw + aux), dtype=np.uint32)
    e.profile(True); e.profile_reset()
    e.sort_records(items, kw)
    st = e.profile_get(); e.profile(False)
    for name, v in st.items():
        if name.startswith("radix_scatter"):
            out["passes"]["%dB" % ((kw + aux) * 4)] = {"kernel": name, "launches": v["launches"], "ms_per_launch": round(v["ms"] / v["launches"], 3),
                                                      "GBs_read_plus_written": round(v["bytes"] / v["ms"] / 1e6, 1), "ps_per_record": round(v["ms"] / v["launches"] * 1e9 / n, 2)}
    del items
print(json.dumps(out))
