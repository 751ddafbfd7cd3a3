#!/bin/bash
mkdir -p gpurun_out/s9
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_count.py tests/test_gpu_comm.py tests/test_gpu_passes.py -x -q -m gpu > gpurun_out/s9/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/s9/pytest.log
tail -25 gpurun_out/s9/pytest.log
for v in 1 0; do
MHX_COUNT_SEG=$v timeout 600 python bench.py --engine count --steps 5 --warmup 2 > gpurun_out/s9/bench_count_seg$v.json 2> gpurun_out/s9/bench_count_seg$v.err
python - <<PY
import json
d=json.load(open("gpurun_out/s9/bench_count_seg$v.json"))
print("count_seg=$v", d["ms_per_step"], d["value"], d["parity_checked"], d["roofline"]["kernel_ms_per_step"])
PY
done
