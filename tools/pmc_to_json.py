"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv passes -> per-kernel HBM bytes per launch (JSON).

    python tools/pmc_to_json.py <dir_fetch> <dir_write> > profiles/rNN_pmc_traffic.json

Units and corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes actually fetched — verified here against kernels of known traffic in the SAME
runs: k_pack_solid reads the 1 B/base byte map with 16-byte loads (x2.00), the tile kernels stage 8-byte
records (x1.98), radix_hist touches every 64 B sector with 4-byte loads (x2.00); WRITE_SIZE matches known
write volumes 1:1 (s1_extract: 21.46 GB reported vs 21.28 GB written; fillBuffer 1.50 vs 1.50 GB)."""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megahit_amd.buildid import build_id, lib_id  # noqa: E402

import pandas as pd


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    df = pd.read_csv(f)
    df = df[df["Counter_Name"] == counter]
    df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("mhx::", "")
    g = df.groupby("k")["Counter_Value"].agg(["count", "mean"])
    return {k: (int(r["count"]), float(r["mean"])) for k, r in g.iterrows()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_doc": "HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (see tools/pmc_to_json.py)",
       "build_id": build_id(), "lib_id": lib_id(), "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fr = 2 * 1024 * fetch.get(k, (0, 0.0))[1]
    wr = 1024 * write.get(k, (0, 0.0))[1]
    out["kernels"][k] = {"launches": fetch.get(k, write.get(k))[0], "read_bytes": round(fr), "write_bytes": round(wr), "hbm_bytes": round(fr + wr)}
print(json.dumps(out, indent=1))
