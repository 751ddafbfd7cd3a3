#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r3k
(timeout 900 python -m pytest tests/test_gpu_sdbg.py tests/test_gpu_comm.py tests/test_gpu_multiprocess.py -x -q > $O/${T}_tests1.log 2>&1; echo "rc=$?" >> $O/${T}_tests1.log)
tail -4 $O/${T}_tests1.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "read2sdbg" > $O/${T}_tests2.log 2>&1; echo "rc=$?" >> $O/${T}_tests2.log)
tail -3 $O/${T}_tests2.log
i=0
for v in "X=1" "MHX_S1_GEN_ANY_ORDER=0" "MHX_S1_DIGIT_HIST_BLOCKED=0" "X=2"; do
i=$((i+1))
env $v timeout 300 python bench.py --steps 8 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_b$i.json 2> $O/${T}_b$i.err
python - <<P
import json
try:
    d=json.loads(open("$O/${T}_b$i.json").read().splitlines()[0]); print("$v", d["ms_per_step"], d.get("parity_checked"), d["roofline"]["kernel"], d["roofline"]["frac"], json.dumps(d["roofline"]["kernel_ms_per_step"])[:420])
except Exception as e: print("FAILED", e); print(open("$O/${T}_b$i.err").read()[-1500:])
P
done
