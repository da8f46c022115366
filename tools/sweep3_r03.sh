#!/bin/bash
cd $GRAFT_REPO_ROOT
B="--steps 300 --warmup 30 --no-cpu-baseline --no-full-path --no-verify --no-batched"
for rep in 1 2; do for fl in 3 4; do for v in "$@"; do
  lib=build/$v/libswfr.so; [ "$v" = base ] && lib=swf_renderer_amd/libswfr.so
  SWFR_FRAMES_IN_FLIGHT=$fl python3 tools/bench_with_lib.py $lib $B 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in_flight $fl', '$v', l['value'], l['ms_per_step'], l['kernel_ms_per_frame']['k2_rows'], l['kernel_ms_per_frame']['k2_tiles'])"
done; done; done
