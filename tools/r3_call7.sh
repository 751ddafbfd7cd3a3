#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r3g
(timeout 900 python -m pytest tests/test_gpu_sdbg.py tests/test_gpu_comm.py tests/test_gpu_buildlib.py tests/test_gpu_fullsize.py -x -q > $O/${T}_tests1.log 2>&1; echo "rc=$?" >> $O/${T}_tests1.log)
tail -6 $O/${T}_tests1.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 300 python tools/mercy_prof.py 10e6 > $O/${T}_mercy_stage1.json 2> $O/${T}_mercy.err; head -c 900 $O/${T}_mercy_stage1.json
timeout 600 python tools/next_rows_bench.py > $O/${T}_next_rows.json 2> $O/${T}_next_rows.err; python - <<P
import json
try:
    d=json.load(open("$O/${T}_next_rows.json"))
    for k,v in d.items():
        if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a!="kernels"}); print("   ", json.dumps(v.get("kernels"))[:900])
except Exception as e: print("next rows FAILED", e); print(open("$O/${T}_next_rows.err").read()[-1500:])
P
timeout 300 python tools/buildlib_bench.py > $O/${T}_buildlib.json 2> $O/${T}_buildlib.err; cat $O/${T}_buildlib.json | head -60
timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_fd.json 2> $O/${T}_fd.err
python - <<P
import json
try:
    d=json.loads(open("$O/${T}_fd.json").read().splitlines()[0]); print("force-dist", d["ms_per_step"], d.get("parity_checked"), json.dumps(d["roofline"]["kernel_ms_per_step"]))
except Exception as e: print("FAILED", e); print(open("$O/${T}_fd.err").read()[-1500:])
P
