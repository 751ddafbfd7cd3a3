#!/bin/bash
mkdir -p gpurun_out/s6
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline --no-e2e > gpurun_out/s6/bench_force_dist.json 2> gpurun_out/s6/bench_force_dist.err
tail -c 1500 gpurun_out/s6/bench_force_dist.json; tail -3 gpurun_out/s6/bench_force_dist.err
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/s6/bench_full.json 2> gpurun_out/s6/bench_full.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/s6/bench_full.json"))
print(d["ms_per_step"], d["value"], d["parity_checked"])
print(json.dumps(d.get("cpu_baseline"), indent=1))
print(json.dumps(d.get("e2e"), indent=1))
PY
tail -3 gpurun_out/s6/bench_full.err
