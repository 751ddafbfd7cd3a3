#!/bin/bash
mkdir -p gpurun_out/s4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_consume.py -x -q -m gpu > gpurun_out/s4/pytest_consume.log 2>&1
echo "rc=$?" >> gpurun_out/s4/pytest_consume.log
tail -30 gpurun_out/s4/pytest_consume.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_consume.py > gpurun_out/s4/pytest_all.log 2>&1
echo "rc=$?" >> gpurun_out/s4/pytest_all.log
tail -15 gpurun_out/s4/pytest_all.log
