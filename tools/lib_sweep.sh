#!/bin/bash
# S1 / S2 bench lines for alternative builds of the library (build/<name>/libswfr.so from tools/build_variant.sh).   usage: bash tools/lib_sweep.sh <reps> name...
R=$GRAFT_REPO_ROOT; cd $R
reps=$1; shift
for rep in $(seq 1 $reps); do for n in "$@"; do for wl in s1 s2; do
  lib=swf_renderer_amd/libswfr.so; [ "$n" != "head" ] && lib=build/$n/libswfr.so
  timeout -k 10 300 python tools/bench_with_lib.py $lib --workload $wl --steps 300 --warmup 20 --no-cpu-baseline --no-batched --no-full-path 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib', '$n', '$wl', 'ms_per_step', d['ms_per_step'], d.get('kernel_ms_per_frame',{}).get('k2_rows'))"
done; done; done
