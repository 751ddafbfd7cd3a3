#!/bin/bash
# round 3, GPU call 1: parity of the new paths, bench + A/B of every new knob, k-list config bench
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r3a
(timeout 900 python -m pytest tests/test_gpu_hybrid_sort.py tests/test_gpu_sdbg.py tests/test_gpu_count.py tests/test_gpu_cli.py -x -q > $O/${T}_tests1.log 2>&1; echo "rc=$?" >> $O/${T}_tests1.log)
tail -5 $O/${T}_tests1.log
(timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_klist.py -q > $O/${T}_tests2.log 2>&1; echo "rc=$?" >> $O/${T}_tests2.log)
tail -8 $O/${T}_tests2.log
timeout 900 python bench.py --steps 10 --warmup 2 > $O/${T}_bench.json 2> $O/${T}_bench.err
cat $O/${T}_bench.json | cut -c1-1500
i=0
for v in "MHX_S1_EXTRACT_FAST=0" "MHX_S1_STREAM_UNROLL=1" "MHX_S1_STREAM_UNROLL=2" "MHX_SORT_HYBRID=0" "MHX_S2_AGG_IN_PLACE=0" "MHX_S1_EXTRACT_FAST=0 MHX_S1_STREAM_UNROLL=1 MHX_SORT_HYBRID=0 MHX_S2_AGG_IN_PLACE=0"; do
  i=$((i+1))
  env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-e2e --no-cpu-baseline > $O/${T}_ab$i.json 2> $O/${T}_ab$i.err
  echo "$v" >> $O/${T}_ab$i.json
  python - <<P
import json
l=open("$O/${T}_ab$i.json").read().splitlines()
try:
    d=json.loads(l[0]); print("$v", d["ms_per_step"], d.get("parity_checked"), json.dumps(d["roofline"]["kernel_ms_per_step"]))
except Exception as e: print("$v", "FAILED", e)
P
done
timeout 900 python tools/config_bench.py klist > $O/${T}_klist.json 2> $O/${T}_klist.err
tail -8 $O/${T}_klist.err
