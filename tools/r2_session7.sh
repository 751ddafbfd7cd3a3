#!/bin/bash
mkdir -p gpurun_out/s7
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/s7/pytest_all.log 2>&1
echo "rc=$?" >> gpurun_out/s7/pytest_all.log
tail -8 gpurun_out/s7/pytest_all.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s7/bench.json 2> gpurun_out/s7/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/s7/bench.json"))
print(d["ms_per_step"], d["value"], d["parity_checked"])
print(json.dumps(d.get("e2e"), indent=1))
PY
