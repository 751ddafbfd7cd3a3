#!/bin/bash
# Round 6: ONE gpurun call collects every profiles/r06_* file on ONE build, stamps each with the ids of the kernel sources and of the
# library (tools/stamp_build.py) and checks that they agree.
#     gpurun --timeout 3300 -- 'bash tools/gpu_evidence_r06.sh [skip_big]'
# Results land in gpurun_out/r06_*; the caller copies them into profiles/ (the PMC file right here, so that the bench line of this
# very call can quote the traffic of this very build).
TAG=r06
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e"
# twice: the default (stage 1 on super-k-mer records) and, as *_prefix_*, the prefix plan (MHX_S1_SKM=0: what the analysis of DESIGN 4l-4n is about)
for MODE in skm prefix; do
  if [ $MODE = prefix ]; then export MHX_S1_SKM=0; T=${TAG}_prefix; else unset MHX_S1_SKM; T=${TAG}; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_trace -- $BENCH > $O/${T}_trace.log 2>&1; echo "$MODE trace rc=$?"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}_pmc_fetch -- $BENCH > $O/${T}_pmc_fetch.log 2>&1; echo "$MODE fetch rc=$?"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}_pmc_write -- $BENCH > $O/${T}_pmc_write.log 2>&1; echo "$MODE write rc=$?"
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/${T}_sq1 -- $BENCH > $O/${T}_sq1.log 2>&1; echo "$MODE sq1 rc=$?"
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/${T}_sq2 -- $BENCH > $O/${T}_sq2.log 2>&1; echo "$MODE sq2 rc=$?"
  ( cd $R; python tools/pmc_sq.py $O/${T}_sq1 $O/${T}_sq2 > $O/${T}_pmc_sq.json 2> $O/${T}_pmc_sq.err; python tools/pmc_to_json.py $O/${T}_pmc_fetch $O/${T}_pmc_write > $O/${T}_pmc_traffic.json 2> $O/${T}_pmc_to_json.err )
  find $O/${T}_sq1 $O/${T}_sq2 -type f -delete 2>/dev/null
  find $O/${T}_trace -name '*kernel_stats.csv' -exec cp {} $O/${T}_kernel_stats.csv \;
  find $O/${T}_trace $O/${T}_pmc_fetch $O/${T}_pmc_write -type f ! -name '*stats.csv' -delete 2>/dev/null
done
unset MHX_S1_SKM
cd $R
cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
# the other sub-programs and the multi-GPU code path with one rank
timeout 300 python bench.py --engine count --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/${TAG}_bench_count.json 2> $O/${TAG}_bench_count.err; echo "count rc=$?"
timeout 300 python bench.py --engine seq2sdbg --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/${TAG}_bench_seq2sdbg.json 2> $O/${TAG}_bench_seq2sdbg.err; echo "seq2sdbg rc=$?"
timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/${TAG}_bench_force_dist.json 2> $O/${TAG}_bench_force_dist.err; echo "dist rc=$?"
MHX_DIST_SKM=0 timeout 300 python bench.py --force-dist --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/${TAG}_bench_force_dist_prefix.json 2> $O/${TAG}_bench_force_dist_prefix.err; echo "dist prefix rc=$?"
cut -c1-240 $O/${TAG}_bench_count.json $O/${TAG}_bench_seq2sdbg.json $O/${TAG}_bench_force_dist.json $O/${TAG}_bench_force_dist_prefix.json
# low-complexity reads on every path, libraries of several read lengths, the A/Bs of the round in one process each
timeout 600 python tools/lowcomplexity_probe.py > $O/${TAG}_lowcomplexity.json 2> $O/${TAG}_lowcomplexity.err; echo "lowcomplexity rc=$?"
timeout 900 python tools/lowcomplexity_paths_probe.py > $O/${TAG}_lowcomplexity_paths.json 2> $O/${TAG}_lowcomplexity_paths.err; echo "lowcomplexity paths rc=$?"
timeout 600 python tools/varlen_bench.py > $O/${TAG}_varlen.json 2> $O/${TAG}_varlen.err; echo "varlen rc=$?"
timeout 300 python tools/ab_options.py "s1_skm=0" "s1_skm=1" "s1_skm=1 s1_skm_deal=0" --rounds 2 > $O/${TAG}_ab_skm.jsonl 2> $O/${TAG}_ab_skm.err; echo "ab skm rc=$?"
MHX_S1_SKM=0 timeout 600 python tools/lowcomplexity_probe.py > $O/${TAG}_lowcomplexity_prefix.json 2> $O/${TAG}_lowcomplexity_prefix.err; echo "lowcomplexity prefix rc=$?"
timeout 300 python tools/ab_options.py "s1_skm=0 sort_loaded_ut2=0" "s1_skm=0 sort_loaded_ut2=1" --rounds 2 > $O/${TAG}_ab_loaded_ut2.jsonl 2> $O/${TAG}_ab_loaded_ut2.err; echo "ab ut2 rc=$?"
timeout 300 python tools/ab_options.py "count_skm=0 count_stream=0" "count_skm=0 count_stream=1" "count_skm=1 count_stream=1" --engine count --rounds 2 > $O/${TAG}_ab_count_stream.jsonl 2> $O/${TAG}_ab_count_stream.err; echo "ab count rc=$?"
timeout 120 python tools/probe_sort_widths.py 5e8 > $O/${TAG}_sort_widths.json 2> $O/${TAG}_sort_widths.err; echo "sort widths rc=$?"
timeout 300 tools/micro/alloc_probe 200 > $O/${TAG}_alloc_probe.jsonl 2>&1; echo "alloc probe rc=$?"
timeout 300 python tools/mercy_prof.py 10e6 > $O/${TAG}_mercy_stage1.json 2> $O/${TAG}_mercy_stage1.err
timeout 300 python tools/buildlib_bench.py > $O/${TAG}_buildlib.json 2> $O/${TAG}_buildlib.err
timeout 600 python tools/next_rows_bench.py > $O/${TAG}_next_rows.json 2> $O/${TAG}_next_rows.err
timeout 600 python tools/e2e_routes.py > $O/${TAG}_e2e_routes.json 2> $O/${TAG}_e2e_routes.err
if [ -z "$1" ]; then
  timeout 600 python tools/config_bench.py klist > $O/${TAG}_bench_klist.json 2> $O/${TAG}_bench_klist.err; tail -3 $O/${TAG}_bench_klist.err
  timeout 1200 python tools/config_bench.py meta > $O/${TAG}_bench_meta.json 2> $O/${TAG}_bench_meta.err; tail -2 $O/${TAG}_bench_meta.err
  # the north-star size (BASELINE configs[2]): 100 M reads on one GPU (memory plan), as eight ranks on this device, one GPU's share of
  # the 8-GPU job eighth by eighth — digests against tests/golden/fullsize_100M.json
  mkdir -p /tmp/c2lib
  timeout 900 python tools/config_bench.py configs2 /tmp/c2lib > $O/${TAG}_bench_configs2.json 2> $O/${TAG}_bench_configs2.err; tail -3 $O/${TAG}_bench_configs2.err
  timeout 600 python tools/config_bench.py owner8 /tmp/c2lib > $O/${TAG}_bench_owner8.json 2> $O/${TAG}_bench_owner8.err; tail -2 $O/${TAG}_bench_owner8.err
  rm -rf /tmp/c2lib
fi
# the headline line last (default flags: CPU baseline sample, the reference on the whole workload alone at the end, end to end)
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
cut -c1-2000 $O/${TAG}_bench.json
# every JSON of the set carries the ids of this build; they all have to agree
python tools/stamp_build.py stamp $O/${TAG}_*.json $O/${TAG}_*.jsonl > /dev/null
python tools/stamp_build.py check $O/${TAG}_*.json $O/${TAG}_*.jsonl
ls -la $O | grep ${TAG}_ | tail -40
