#!/bin/bash
# One gpurun call: GPU tests, the bench line, rocprofv3 kernel stats and the two PMC passes of the same command.
#   gpurun --timeout 1500 -- 'bash tools/gpu_evidence.sh TAG [skip_tests]'
# Results land in gpurun_out/TAG_*; copy what is to be judged into profiles/.
TAG=${1:-r01}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$2" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1
  tail -3 $O/${TAG}_pytest.log
fi
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cat $O/${TAG}_bench.json
timeout 600 python bench.py --engine count --steps 3 > $O/${TAG}_bench_count.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --engine seq2sdbg --steps 3 > $O/${TAG}_bench_seq2sdbg.json 2>> $O/${TAG}_bench.err
cat $O/${TAG}_bench_count.json $O/${TAG}_bench_seq2sdbg.json
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- $BENCH > $O/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $BENCH > $O/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $BENCH > $O/${TAG}_pmc_write.log 2>&1
cd $R
python tools/pmc_to_json.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write > $O/${TAG}_pmc_traffic.json 2> $O/${TAG}_pmc_to_json.err
# keep only the summaries (the raw traces are large)
find $O/${TAG}_trace -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
find $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -type f ! -name '*stats.csv' -delete 2>/dev/null
ls -la $O | tail -20
