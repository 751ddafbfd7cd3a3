#!/bin/bash
mkdir -p gpurun_out/s14
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sdbg_index.py tests/test_gpu_cli.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/s14/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/s14/pytest.log
tail -30 gpurun_out/s14/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/s14/pytest_full.log 2>&1
echo "rc=$?" >> gpurun_out/s14/pytest_full.log
tail -15 gpurun_out/s14/pytest_full.log
