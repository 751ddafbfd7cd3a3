#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r3f
(timeout 900 python -m pytest tests/test_gpu_sdbg.py tests/test_gpu_count.py tests/test_gpu_passes.py tests/test_gpu_comm.py tests/test_gpu_dist.py -x -q > $O/${T}_tests1.log 2>&1; echo "rc=$?" >> $O/${T}_tests1.log)
tail -6 $O/${T}_tests1.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "read2sdbg or count" > $O/${T}_tests2.log 2>&1; echo "rc=$?" >> $O/${T}_tests2.log)
tail -3 $O/${T}_tests2.log
i=0
for v in "X=1" "MHX_S1_FUSED_FIRST_PASS=0" "X=2" "MHX_S1_FUSED_FIRST_PASS=0"; do
  i=$((i+1))
  env $v timeout 300 python bench.py --steps 8 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_ab$i.json 2> $O/${T}_ab$i.err
  python - <<P
import json
try:
    d=json.loads(open("$O/${T}_ab$i.json").read().splitlines()[0]); print("$v", d["ms_per_step"], d.get("parity_checked"), json.dumps(d["roofline"]["kernel_ms_per_step"]))
except Exception as e: print("$v", "FAILED", e); print(open("$O/${T}_ab$i.err").read()[-1500:])
P
done
for v in "X=1" "MHX_S1_FUSED_FIRST_PASS=0"; do
env $v timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_fd.json 2> $O/${T}_fd.err
python - <<P
import json
try:
    d=json.loads(open("$O/${T}_fd.json").read().splitlines()[0]); print("force-dist $v", d["ms_per_step"], d.get("parity_checked"), json.dumps(d["roofline"]["kernel_ms_per_step"]))
except Exception as e: print("FAILED", e); print(open("$O/${T}_fd.err").read()[-1500:])
P
done
