#!/bin/bash
mkdir -p gpurun_out/s5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/s5/pytest_comm.log 2>&1
echo "rc=$?" >> gpurun_out/s5/pytest_comm.log
tail -40 gpurun_out/s5/pytest_comm.log
