#!/bin/bash
# A/B of library variants in one call (same box): default frames in flight, three repetitions each, interleaved
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  for lib in "$@"; do
    python3 tools/bench_with_lib.py $lib --steps 300 --warmup 30 --no-cpu-baseline --no-full-path 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', l['value'], l['kernel_ms_per_frame'], l['roofline']['one_frame_in_flight'])"
  done
done
