#!/bin/bash
# A/B on ONE box: bench.py under several environment settings (box-to-box variation is larger than most effects).
#   gpurun -- 'bash tools/ab_bench.sh "MHX_SORT=classic" "MHX_SORT_UT=2" ""'
for cfg in "$@"; do
  echo "== ${cfg:-default}"
  env $cfg python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: v for k, v in d['roofline']['kernel_ms_per_step'].items() if v > 1})"
done
