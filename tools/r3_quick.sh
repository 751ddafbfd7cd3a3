#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sdbg.py tests/test_gpu_comm.py tests/test_gpu_fullsize.py tests/test_gpu_multiprocess.py -m gpu -x -q > $O/r3n_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/r3n_tests.log | tail -2
for f in 1 0; do
MHX_S1_DIGIT_HIST_PLAIN=$f timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/r3n_bench_$f.json 2> $O/r3n_bench_$f.err
python - <<P
import json
d=json.load(open("$O/r3n_bench_$f.json")); print("plain=$f",d["ms_per_step"],d["parity_checked"],json.dumps(d["roofline"]["kernel_ms_per_step"]))
P
done
