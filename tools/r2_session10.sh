#!/bin/bash
mkdir -p gpurun_out/s10
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sdbg.py tests/test_gpu_comm.py tests/test_gpu_dist.py tests/test_gpu_passes.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/s10/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/s10/pytest.log
tail -12 gpurun_out/s10/pytest.log
for v in 1 0; do
MHX_S1_STREAM=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > gpurun_out/s10/bench_stream$v.json 2> gpurun_out/s10/bench_stream$v.err
python - <<PY
import json
d=json.load(open("gpurun_out/s10/bench_stream$v.json"))
print("stream=$v", d["ms_per_step"], d["value"], d["parity_checked"], d["roofline"]["kernel_ms_per_step"])
PY
done
