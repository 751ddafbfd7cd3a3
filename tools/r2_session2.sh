#!/bin/bash
mkdir -p gpurun_out/s2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sdbg.py -x -q -m gpu -k "segment or compact or matches_oracle" > gpurun_out/s2/pytest_sdbg.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s2/pytest_sdbg.log
tail -5 gpurun_out/s2/pytest_sdbg.log
for v in "8" "4"; do
  MHX_S1_SEG_PER=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s2/bench_seg_per$v.json 2> gpurun_out/s2/bench_seg_per$v.err
  tail -c 1300 gpurun_out/s2/bench_seg_per$v.json
done
