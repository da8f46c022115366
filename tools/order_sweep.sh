#!/bin/bash
# launch order of k2_tiles (SWFR_STRIP_ORDER=0 row-major, 1 heaviest first): S1 / S2 ms_per_step, repeated.   usage: bash tools/order_sweep.sh <reps>
R=$GRAFT_REPO_ROOT; cd $R
for rep in $(seq 1 ${1:-2}); do for o in 1 0; do for wl in s1 s2; do
  SWFR_STRIP_ORDER=$o timeout -k 10 300 python bench.py --workload $wl --steps 300 --warmup 20 --no-cpu-baseline --no-batched 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order', $o, '$wl', 'ms_per_step', d['ms_per_step'], 'one in flight', d.get('kernel_ms_per_frame'))" | cut -c1-200
done; done; done
