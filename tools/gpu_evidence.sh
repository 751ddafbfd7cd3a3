#!/bin/bash
# One gpurun call: the bench lines, rocprofv3 kernel stats and the two PMC passes of the same command, the reference's
# CPU path at full size on this host.   gpurun --timeout 2400 -- 'bash tools/gpu_evidence.sh r02 [skip_cpu]'
# Results land in gpurun_out/TAG_*; copy what is to be judged into profiles/.
TAG=${1:-r02}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$2" ]; then
  # the reference at full size, in the background on host cores while the GPU-side evidence is collected
  (python tools/cpu_fullsize.py --threads 8,32 > $O/${TAG}_cpu_fullsize.json 2> $O/${TAG}_cpu_fullsize.err) &
  CPU_PID=$!
fi
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- $BENCH > $O/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $BENCH > $O/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $BENCH > $O/${TAG}_pmc_write.log 2>&1
cd $R
python tools/pmc_to_json.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write > $O/${TAG}_pmc_traffic.json 2> $O/${TAG}_pmc_to_json.err
cp $O/${TAG}_pmc_traffic.json profiles/r02_pmc_traffic.json   # so that the bench line below carries the traffic of this very tree
find $O/${TAG}_trace -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
find $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -type f ! -name '*stats.csv' -delete 2>/dev/null
[ -n "$CPU_PID" ] && wait $CPU_PID && cp $O/${TAG}_cpu_fullsize.json profiles/r02_cpu_fullsize.json
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cat $O/${TAG}_bench.json
timeout 600 python bench.py --engine count --steps 3 --no-e2e > $O/${TAG}_bench_count.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --engine seq2sdbg --steps 3 --no-e2e > $O/${TAG}_bench_seq2sdbg.json 2>> $O/${TAG}_bench.err
timeout 600 python bench.py --force-dist --steps 3 --no-cpu-baseline --no-e2e > $O/${TAG}_bench_force_dist.json 2>> $O/${TAG}_bench.err
cat $O/${TAG}_bench_count.json $O/${TAG}_bench_seq2sdbg.json
# the three routes to the first graph through the CLI (files in -> files out), stage 1 with the reference-exact tie order,
# buildlib, and the copy-invariance check at 20 M / 40 M reads on one GPU
timeout 600 python tools/e2e_routes.py > $O/${TAG}_e2e_routes.json 2> $O/${TAG}_e2e_routes.err
timeout 300 python tools/mercy_prof.py 10e6 > $O/${TAG}_mercy_stage1.json 2> $O/${TAG}_mercy_stage1.err
timeout 300 python tools/buildlib_bench.py > $O/${TAG}_buildlib.json 2> $O/${TAG}_buildlib.err
timeout 600 python tools/scale_check.py 10e6 2 > $O/${TAG}_scale_20M.json 2> $O/${TAG}_scale_20M.err
timeout 900 python tools/scale_check.py 10e6 4 > $O/${TAG}_scale_40M.json 2> $O/${TAG}_scale_40M.err
tail -c 600 $O/${TAG}_scale_40M.json
ls -la $O | tail -20
