#!/bin/bash
mkdir -p gpurun_out/s11
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_buildlib.py tests/test_gpu_consume.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/s11/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/s11/pytest.log
tail -30 gpurun_out/s11/pytest.log
timeout 600 python tools/buildlib_bench.py --reads 4e6 > gpurun_out/s11/buildlib_bench.json 2> gpurun_out/s11/buildlib_bench.err
cat gpurun_out/s11/buildlib_bench.json; tail -3 gpurun_out/s11/buildlib_bench.err
