#!/bin/bash
# One gpurun call: parity of the chained-scan pass with unit-wide runs (both settings of the knob), A/B on one box, PMC bytes.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sort_unit_runs.py -m gpu -x -q > $O/u_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|error" $O/u_tests.log | tail -3
timeout 400 python tools/ab_options.py "sort_unit_runs=0" "sort_unit_runs=1" --rounds 2 > $O/u_ab.jsonl 2> $O/u_ab.err; echo "ab rc=$?"; cut -c1-600 $O/u_ab.jsonl; tail -2 $O/u_ab.err
cd /tmp
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/u_pmc_write -- python $R/tools/ab_options.py "sort_unit_runs=1" --steps 2 --warmup 1 > $O/u_pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/u_pmc_fetch -- python $R/tools/ab_options.py "sort_unit_runs=1" --steps 2 --warmup 1 > $O/u_pmc_fetch.log 2>&1
cd $R
python tools/pmc_to_json.py $O/u_pmc_fetch $O/u_pmc_write > $O/u_pmc_traffic.json 2> $O/u_pmc_to_json.err
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/u_pmc_traffic.json"))
    for k,v in d["kernels"].items():
        if "onesweep" in k or "s1_stream<true" in k: print(k[:70], v)
except Exception as e: print("pmc:", e)
P
find $O/u_pmc_fetch $O/u_pmc_write -type f -delete 2>/dev/null
timeout 200 python tools/ab_options.py "sort_unit_runs=0" "sort_unit_runs=1" --engine count --steps 4 > $O/u_ab_count.jsonl 2> $O/u_ab_count.err; cut -c1-400 $O/u_ab_count.jsonl
timeout 400 python -m pytest tests/test_gpu_hybrid_sort.py tests/test_gpu_sdbg.py -m gpu -x -q > $O/u_tests2.log 2>&1; echo "tests2 rc=$?"; grep -E "passed|failed|error" $O/u_tests2.log | tail -3
