#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sort_unit_runs.py tests/test_gpu_sdbg.py -m gpu -x -q > $O/e2_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|error" $O/e2_tests.log | tail -3
B="sort_unit_waves=3 s1_stream_debug=0"
timeout 500 python tools/ab_options.py "s1_stream_used_list=0 $B" "s1_stream_used_list=1 $B" "s1_stream_used_list=1 sort_unit_waves=4 s1_stream_debug=0" "s1_stream_used_list=0 $B" "s1_stream_used_list=1 $B" \
   "stage1_only=1 s1_stream_used_list=1 sort_unit_waves=3 s1_stream_debug=0" "stage1_only=1 s1_stream_used_list=1 sort_unit_waves=3 s1_stream_debug=1" "stage1_only=1 s1_stream_used_list=1 sort_unit_waves=3 s1_stream_debug=2" \
   "s1_stream_used_list=1 $B" > $O/e2_ab.jsonl 2> $O/e2_ab.err; echo "ab rc=$?"
python - <<'P'
import json
for l in open("gpurun_out/e2_ab.jsonl"):
    d=json.loads(l); k=d["kernel_ms_per_step"]
    print(d["config"][:75], "|", d["ms_per_step"], d["parity_checked"], {x:k[x] for x in k if k[x]>0.6})
P
tail -2 $O/e2_ab.err
