#!/bin/bash
mkdir -p gpurun_out/s3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 1 2 4 7; do
  MHX_S1_SEG_DBG=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s3/bench_dbg$v.json 2> gpurun_out/s3/bench_dbg$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/s3/bench_dbg$v.json"))
print("dbg=$v", d["ms_per_step"], {k:v for k,v in d["roofline"]["kernel_ms_per_step"].items() if k in ("s1_groups","radix_scatter_12B","s1_extract")})
PY
done
