#!/bin/bash
# One gpurun call (~2.5 box-minutes): tests of the knob-selected code paths, greedy tuning on the box ->
# megahit_amd/mhx_tuning.conf, then the evidence set (tools/evidence_short.sh) under the tuned defaults that will ship.
#     gpurun --timeout 420 -- 'bash tools/tune_and_evidence.sh r04 "s1_gen_blocked s1_digit_hist_preload"'
# TAG names the files under gpurun_out/ (copy what is to be judged into profiles/ and the tuning file into megahit_amd/).
# KNOBS (optional) are tried on top of the current tuned defaults, in this order; a knob the library does not know changes
# nothing and is not kept.
TAG=${1:-r04}
KNOBS=${2:-"s1_gen_blocked s1_digit_hist_preload"}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TESTS="tests/test_gpu_round3_knobs.py tests/test_gpu_tuning.py tests/test_gpu_sort_unit_runs.py tests/test_gpu_sdbg.py"
timeout 200 python -m pytest $TESTS -m gpu -x -q > $O/${TAG}_tune_tests.log 2>&1; T=$?
echo "tests rc=$T"; grep -E "passed|failed|error" $O/${TAG}_tune_tests.log | tail -3
if [ $T -ne 0 ]; then tail -40 $O/${TAG}_tune_tests.log; exit 1; fi
# start from the committed tuned defaults (every knob of the file, with its value) plus the fixed starting values of multi-valued knobs
BASE=$(python - <<'P'
import bench
d = bench.tuned_defaults()
print(" ".join("%s=%d" % kv for kv in d.items()))
P
)
timeout 150 python tools/ab_options.py "$BASE" --greedy "$KNOBS" --write-tuning $O/mhx_tuning.conf > $O/${TAG}_ab_greedy_tuning.jsonl 2> $O/${TAG}_ab.err; echo "ab rc=$?"
python - "$O/${TAG}_ab_greedy_tuning.jsonl" <<'P'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); k = d["kernel_ms_per_step"]
    print(d.get("note"), "|", d["ms_per_step"], d["parity_checked"], {x: k[x] for x in k if k[x] > 2.0})
P
tail -1 $O/${TAG}_ab.err
[ -f $O/mhx_tuning.conf ] && cp $O/mhx_tuning.conf megahit_amd/mhx_tuning.conf && cat megahit_amd/mhx_tuning.conf
bash tools/evidence_short.sh $TAG
