#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r3h
(timeout 1200 python -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_sdbg_index.py tests/test_gpu_iterate.py tests/test_gpu_next_rows_golden.py tests/test_gpu_comm.py -x -q > $O/${T}_tests1.log 2>&1; echo "rc=$?" >> $O/${T}_tests1.log)
tail -12 $O/${T}_tests1.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 600 python tools/next_rows_bench.py > $O/${T}_next_rows.json 2> $O/${T}_next_rows.err; python - <<P
import json
try:
    d=json.load(open("$O/${T}_next_rows.json"))
    for k,v in d.items():
        if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a!="kernels"}); print("   ", json.dumps(v.get("kernels"))[:700])
except Exception as e: print("next rows FAILED", e); print(open("$O/${T}_next_rows.err").read()[-1500:])
P
