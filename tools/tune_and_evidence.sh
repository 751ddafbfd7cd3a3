#!/bin/bash
# One gpurun call: tests of the knob-selected code paths, greedy tuning on the box -> megahit_amd/mhx_tuning.conf, then the
# evidence set (tools/evidence_short.sh) under the tuned defaults that will ship.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_round3_knobs.py tests/test_gpu_tuning.py tests/test_gpu_sort_unit_runs.py tests/test_gpu_sdbg.py -m gpu -x -q > $O/e4_tests.log 2>&1; T=$?
echo "tests rc=$T"; grep -E "passed|failed|error" $O/e4_tests.log | tail -3
if [ $T -ne 0 ]; then tail -40 $O/e4_tests.log; exit 1; fi
timeout 120 python tools/ab_options.py "sort_rank_atomic=1" --greedy "s1_gen_blocked s1_digit_hist_preload s1_stream_read_first s1_stream_half" --write-tuning $O/mhx_tuning.conf > $O/e4_ab.jsonl 2> $O/e4_ab.err; echo "ab rc=$?"
python - <<'P'
import json
for l in open("gpurun_out/e4_ab.jsonl"):
    d=json.loads(l); k=d["kernel_ms_per_step"]
    print(d.get("note"), "|", d["ms_per_step"], d["parity_checked"], {x:k[x] for x in k if k[x]>2.0})
P
tail -1 $O/e4_ab.err
[ -f $O/mhx_tuning.conf ] && cp $O/mhx_tuning.conf megahit_amd/mhx_tuning.conf && cat megahit_amd/mhx_tuning.conf
bash tools/evidence_short.sh r03
