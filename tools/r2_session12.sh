#!/bin/bash
mkdir -p gpurun_out/s12
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/scale_check.py 10e6 2 > gpurun_out/s12/scale_20M_seg.json 2> gpurun_out/s12/scale_20M_seg.err
cat gpurun_out/s12/scale_20M_seg.json; grep "S1 done\|S2 done" gpurun_out/s12/scale_20M_seg.err | cut -c1-400
MHX_S1_STREAM_MAX=100000 timeout 900 python tools/scale_check.py 10e6 2 > gpurun_out/s12/scale_20M_stream.json 2> gpurun_out/s12/scale_20M_stream.err
cat gpurun_out/s12/scale_20M_stream.json; grep "S1 done" gpurun_out/s12/scale_20M_stream.err | cut -c1-400
timeout 900 python tools/scale_check.py 10e6 4 > gpurun_out/s12/scale_40M.json 2> gpurun_out/s12/scale_40M.err
cat gpurun_out/s12/scale_40M.json; grep "S1 done" gpurun_out/s12/scale_40M.err | cut -c1-400
