#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r3e
(timeout 1200 python -m pytest tests/test_gpu_comm.py tests/test_gpu_server.py -x -q > $O/${T}_tests1.log 2>&1; echo "rc=$?" >> $O/${T}_tests1.log)
tail -25 $O/${T}_tests1.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 900 python bench.py --steps 8 --warmup 2 > $O/${T}_bench.json 2> $O/${T}_bench.err
python - <<P
import json
d=json.loads(open("$O/${T}_bench.json").read().splitlines()[0]); print("bench", d["ms_per_step"], d.get("parity_checked")); print(json.dumps(d["e2e"], indent=0)[:3000])
P
timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_fd.json 2> $O/${T}_fd.err
python - <<P
import json
try:
    d=json.loads(open("$O/${T}_fd.json").read().splitlines()[0]); print("force-dist", d["ms_per_step"], d.get("parity_checked"))
except Exception as e: print("FAILED", e); print(open("$O/${T}_fd.err").read()[-1500:])
P
