#!/bin/bash
# tuning matrix (gpurun): library variants x persistent grid sizes, one frame in flight (kernel times undisturbed)
R=$GRAFT_REPO_ROOT; cd $R
for lib in "$@"; do
  for g in 4096 5120 6144 8192 16200; do
    SWFR_TILES_GRID=$g SWFR_FRAMES_IN_FLIGHT=1 python3 tools/bench_with_lib.py $lib --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', $g, l['value'], l['kernel_ms_per_frame'])"
  done
done
