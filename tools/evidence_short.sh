#!/bin/bash
# Trimmed evidence set for a short GPU budget (one gpurun call): rocprofv3 kernel stats, the two PMC passes of the same command,
# the bench line (default flags: CPU baseline sample + end-to-end block) quoting the PMC traffic of this very build, the
# multi-GPU code path with one rank.     gpurun --timeout 520 -- 'bash tools/evidence_short.sh r03'
TAG=${1:-r03}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e"
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- $BENCH > $O/${TAG}_trace.log 2>&1; echo "trace rc=$?"
timeout 70 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $BENCH > $O/${TAG}_pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 70 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $BENCH > $O/${TAG}_pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python tools/pmc_to_json.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write > $O/${TAG}_pmc_traffic.json 2> $O/${TAG}_pmc_to_json.err && cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
find $O/${TAG}_trace -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
find $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -type f ! -name '*stats.csv' -delete 2>/dev/null
grep -h '^{' $O/${TAG}_trace.log | tail -1 | cut -c1-300
timeout 220 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
cut -c1-1500 $O/${TAG}_bench.json
timeout 45 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/${TAG}_bench_force_dist.json 2> $O/${TAG}_bench_force_dist.err; echo "dist rc=$?"
cut -c1-400 $O/${TAG}_bench_force_dist.json
