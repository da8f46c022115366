#!/bin/bash
# frames in flight x k2_tiles occupancy variants, S1 pipelined throughput
cd $GRAFT_REPO_ROOT
B="--steps 300 --warmup 30 --no-cpu-baseline --no-full-path --no-verify --no-batched"
for fl in 2 3 4; do
  for v in base w4 w5 w7 w8; do
    lib=build/$v/libswfr.so; [ "$v" = base ] && lib=swf_renderer_amd/libswfr.so
    [ -f $lib ] || continue
    SWFR_FRAMES_IN_FLIGHT=$fl python3 tools/bench_with_lib.py $lib $B 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in_flight $fl', '$v', l['value'], l['ms_per_step'])"
  done
done
