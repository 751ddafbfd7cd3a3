#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${1:-r3i}
(timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/${T}_gputests.log 2>&1; echo "rc=$?" >> $O/${T}_gputests.log)
tail -12 $O/${T}_gputests.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
