#!/bin/bash
# One gpurun call: the bench lines, rocprofv3 kernel stats and the two PMC passes of the same command, the reference's
# CPU path at full size on this host, the other sub-programs, routes and rows.
#     gpurun --timeout 3000 -- 'bash tools/gpu_evidence.sh r04 [skip_cpu]'
# Results land in gpurun_out/TAG_*; what is to be judged is copied into profiles/ by the caller (the PMC file right here,
# so that the bench line of this very call can quote it).
TAG=${1:-r04}
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
cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # so that the bench line below carries the traffic of this very build
find $O/${TAG}_trace -name '*kernel_stats.csv' -exec cp {} $O/${TAG}_kernel_stats.csv \;
find $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -type f ! -name '*stats.csv' -delete 2>/dev/null
timeout 300 python bench.py --engine count --steps 3 --no-e2e > $O/${TAG}_bench_count.json 2> $O/${TAG}_bench_count.err
timeout 300 python bench.py --engine seq2sdbg --steps 3 --no-e2e > $O/${TAG}_bench_seq2sdbg.json 2> $O/${TAG}_bench_seq2sdbg.err
timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/${TAG}_bench_force_dist.json 2> $O/${TAG}_bench_force_dist.err
cut -c1-300 $O/${TAG}_bench_count.json $O/${TAG}_bench_seq2sdbg.json $O/${TAG}_bench_force_dist.json
timeout 300 python tools/mercy_prof.py 10e6 > $O/${TAG}_mercy_stage1.json 2> $O/${TAG}_mercy_stage1.err
timeout 300 python tools/buildlib_bench.py > $O/${TAG}_buildlib.json 2> $O/${TAG}_buildlib.err
timeout 600 python tools/next_rows_bench.py > $O/${TAG}_next_rows.json 2> $O/${TAG}_next_rows.err
timeout 600 python tools/config_bench.py klist > $O/${TAG}_bench_klist.json 2> $O/${TAG}_bench_klist.err
tail -6 $O/${TAG}_bench_klist.err
timeout 1200 python tools/config_bench.py meta > $O/${TAG}_bench_meta.json 2> $O/${TAG}_bench_meta.err
tail -3 $O/${TAG}_bench_meta.err
[ -n "$CPU_PID" ] && wait $CPU_PID && cp $O/${TAG}_cpu_fullsize.json profiles/${TAG}_cpu_fullsize.json
# the headline line last, on a device that the runs above have left (its e2e part starts a resident server first)
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cat $O/${TAG}_bench.json | cut -c1-2500
timeout 600 python tools/e2e_routes.py > $O/${TAG}_e2e_routes.json 2> $O/${TAG}_e2e_routes.err
# the north-star size (BASELINE configs[2]): 100 M reads on one GPU (memory plan), as eight ranks on this device, and one GPU's
# share of the 8-GPU job eighth by eighth — digests against tests/golden/fullsize_100M.json
mkdir -p /tmp/c2lib
timeout 900 python tools/config_bench.py configs2 /tmp/c2lib > $O/${TAG}_bench_configs2.json 2> $O/${TAG}_bench_configs2.err; tail -3 $O/${TAG}_bench_configs2.err
timeout 600 python tools/config_bench.py owner8 /tmp/c2lib > $O/${TAG}_bench_owner8.json 2> $O/${TAG}_bench_owner8.err; tail -2 $O/${TAG}_bench_owner8.err
rm -rf /tmp/c2lib
# what LDS operations, the insert forms and the read patterns of the bucket streaming cost on this device (tools/micro)
for m in lds_probe insert_probe read_probe; do [ -x tools/micro/$m ] && timeout 60 ./tools/micro/$m > $O/${TAG}_micro_$m.txt 2>&1; done
ls -la $O | grep ${TAG}_ | tail -30
