#!/bin/bash
# two SQ counter passes over a short bench run -> gpurun_out/<tag>_pmc_sq.json     gpurun -- 'bash tools/pmc_sq.sh r06'
TAG=${1:-r06}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/${TAG}_sq1 -- $BENCH > $O/${TAG}_sq1.log 2>&1; echo "sq1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/${TAG}_sq2 -- $BENCH > $O/${TAG}_sq2.log 2>&1; echo "sq2 rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $O/${TAG}_sq3 -- $BENCH > $O/${TAG}_sq3.log 2>&1; echo "sq3 rc=$?"
cd $R
python tools/pmc_sq.py $O/${TAG}_sq1 $O/${TAG}_sq2 $O/${TAG}_sq3 > $O/${TAG}_pmc_sq.json 2> $O/${TAG}_pmc_sq.err
find $O/${TAG}_sq1 $O/${TAG}_sq2 $O/${TAG}_sq3 -type f -delete 2>/dev/null
tail -3 $O/${TAG}_sq1.log $O/${TAG}_sq2.log $O/${TAG}_sq3.log | cut -c1-300
head -c 6000 $O/${TAG}_pmc_sq.json
