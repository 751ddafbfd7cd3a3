#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r3b
(timeout 600 python -m pytest tests/test_gpu_hybrid_sort.py tests/test_gpu_sdbg.py -x -q > $O/${T}_tests1.log 2>&1; echo "rc=$?" >> $O/${T}_tests1.log)
tail -3 $O/${T}_tests1.log
(timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "read2sdbg or count" > $O/${T}_tests2.log 2>&1; echo "rc=$?" >> $O/${T}_tests2.log)
tail -3 $O/${T}_tests2.log
i=0
for v in "X=1" "MHX_S1_EXTRACT_ITEMS=1" "MHX_S1_EXTRACT_ITEMS=2" "MHX_S1_EXTRACT_ITEMS=8" "MHX_SORT_HYBRID=0" "MHX_SORT_HYBRID_AVG=4" "MHX_SORT_HYBRID_AVG=40"; do
  i=$((i+1))
  env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_ab$i.json 2> $O/${T}_ab$i.err
  python - <<P
import json
try:
    d=json.loads(open("$O/${T}_ab$i.json").read().splitlines()[0]); print("$v", d["ms_per_step"], d.get("parity_checked"), json.dumps(d["roofline"]["kernel_ms_per_step"]))
except Exception as e: print("$v", "FAILED", e)
P
done
MHX_LIBRARY=$R/megahit_amd/libmhx_timing.so timeout 300 python tools/probe_phases.py > $O/${T}_phases.txt 2> $O/${T}_phases.err
cat $O/${T}_phases.txt
timeout 600 python tools/churn_probe.py > $O/${T}_churn.json 2> $O/${T}_churn.err
cat $O/${T}_churn.err | tail -5
timeout 600 python tools/config_bench.py klist > $O/${T}_klist.json 2> $O/${T}_klist.err
tail -8 $O/${T}_klist.err
