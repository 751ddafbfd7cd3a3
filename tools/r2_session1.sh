#!/bin/bash
# round-2 GPU session 1: parity of the segment group-by, A/B timings, full-size known answers
mkdir -p gpurun_out/s1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sdbg.py tests/test_gpu_dist.py tests/test_gpu_passes.py -x -q -m gpu > gpurun_out/s1/pytest_sdbg.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1/pytest_sdbg.log
tail -5 gpurun_out/s1/pytest_sdbg.log
for v in "8" "4"; do
  MHX_S1_SEG_PER=$v timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s1/bench_seg_per$v.json 2> gpurun_out/s1/bench_seg_per$v.err
  tail -c 1500 gpurun_out/s1/bench_seg_per$v.json
done
MHX_S1_SEG=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s1/bench_classic.json 2> gpurun_out/s1/bench_classic.err
tail -c 1500 gpurun_out/s1/bench_classic.json
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu > gpurun_out/s1/pytest_full.log 2>&1
tail -30 gpurun_out/s1/pytest_full.log
