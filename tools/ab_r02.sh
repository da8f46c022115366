#!/bin/bash
# A/B of library variants in one call (same box): default frames in flight, three repetitions each, interleaved; S1 and S2
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  for lib in "$@"; do
    for wl in s1 s2; do
    python3 tools/bench_with_lib.py $lib --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-full-path 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$wl', l['value'], l['kernel_ms_per_frame'], l['roofline']['one_frame_in_flight'])"
    done
  done
done
