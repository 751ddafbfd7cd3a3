#!/bin/bash
# One gpurun call: tests of the new knobs, A/B of the ranking on the box -> megahit_amd/mhx_tuning.conf, then the evidence set
# (tools/r3_final.sh) under the tuned defaults that will ship.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
rm -f megahit_amd/mhx_tuning.conf
timeout 200 python -m pytest tests/test_gpu_tuning.py tests/test_gpu_sort_unit_runs.py tests/test_gpu_sdbg.py -m gpu -x -q > $O/e3_tests.log 2>&1; T=$?
echo "tests rc=$T"; grep -E "passed|failed|error" $O/e3_tests.log | tail -3
if [ $T -eq 0 ]; then
  timeout 200 python tools/ab_options.py "sort_rank_atomic=0" "sort_rank_atomic=1" --rounds 2 --write-tuning megahit_amd/mhx_tuning.conf > $O/e3_ab.jsonl 2> $O/e3_ab.err; echo "ab rc=$?"
  python - <<'P'
import json
for l in open("gpurun_out/e3_ab.jsonl"):
    d=json.loads(l); k=d["kernel_ms_per_step"]
    print(d["config"][:40], "|", d["ms_per_step"], d["parity_checked"], {x:k[x] for x in k if k[x]>2.0})
P
  tail -1 $O/e3_ab.err
  [ -f megahit_amd/mhx_tuning.conf ] && cp megahit_amd/mhx_tuning.conf $O/mhx_tuning.conf && cat megahit_amd/mhx_tuning.conf
else
  tail -30 $O/e3_tests.log
fi
bash tools/r3_final.sh r03
