#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r3d
(timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_dist.py -x -q > $O/${T}_tests1.log 2>&1; echo "rc=$?" >> $O/${T}_tests1.log)
tail -15 $O/${T}_tests1.log
for v in "X=1" "MHX_DIST_PRESORT=0" "MHX_DIST_MARKS_ONE_PASS=0"; do
  env $v timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_fd.json 2> $O/${T}_fd.err
  python - <<P
import json
try:
    d=json.loads(open("$O/${T}_fd.json").read().splitlines()[0]); print("force-dist $v", d["ms_per_step"], d.get("parity_checked"), json.dumps(d["roofline"]["kernel_ms_per_step"]))
except Exception as e: print("$v", "FAILED", e); print(open("$O/${T}_fd.err").read()[-1500:])
P
  cp $O/${T}_fd.json $O/${T}_fd_$(echo $v | tr '=' '_').json
done
timeout 300 python bench.py --steps 6 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_single.json 2> $O/${T}_single.err
python - <<P
import json
d=json.loads(open("$O/${T}_single.json").read().splitlines()[0]); print("single", d["ms_per_step"], d.get("parity_checked"), json.dumps(d["roofline"]["kernel_ms_per_step"]))
P
(timeout 1500 python -m pytest tests/test_gpu_fullsize_meta.py -x -q > $O/${T}_tests2.log 2>&1; echo "rc=$?" >> $O/${T}_tests2.log)
tail -15 $O/${T}_tests2.log
timeout 1200 python tools/config_bench.py meta > $O/${T}_meta.json 2> $O/${T}_meta.err
tail -5 $O/${T}_meta.err
