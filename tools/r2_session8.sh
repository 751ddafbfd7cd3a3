#!/bin/bash
mkdir -p gpurun_out/s8
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sdbg_index.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s8/pytest_index.log 2>&1
echo "rc=$?" >> gpurun_out/s8/pytest_index.log
tail -40 gpurun_out/s8/pytest_index.log
