"""rocprofv3 --pmc SQ_* csv passes -> per-kernel counter means (JSON): where the wave cycles of the leading kernels go.

    python tools/pmc_sq.py <dir_pass1> [<dir_pass2> ...] > profiles/rNN_pmc_sq.json

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY +
ACTIVE_INST_ANY ~ WAVE_CYCLES); SQ_INSTS_* count wave instructions."""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megahit_amd.buildid import build_id, lib_id  # noqa: E402

import pandas as pd

out = {"build_id": build_id(), "lib_id": lib_id(), "command": "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e", "kernels": {}}
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    df = pd.read_csv(fs[0])
    df["k"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("mhx::", "")
    g = df.groupby(["k", "Counter_Name"])["Counter_Value"].agg(["count", "mean"])
    for (k, cn), r in g.iterrows():
        e = out["kernels"].setdefault(k, {"launches": int(r["count"])})
        e[cn] = float(r["mean"])
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]
out["kernels"] = {k: v for k, v in top}
for k, v in out["kernels"].items():
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        v["frac_of_wave_cycles"] = {c: round(v[c] / wc, 4) for c in v if c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE")}
print(json.dumps(out, indent=1))
