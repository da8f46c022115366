#!/bin/bash
# S2 only: ms_per_step per SWFR_TILES_GRID, several repetitions.   usage: bash tools/grid_sweep_s2.sh <reps> grids...
R=$GRAFT_REPO_ROOT; cd $R
reps=$1; shift
for rep in $(seq 1 $reps); do for g in "$@"; do
  SWFR_TILES_GRID=$g timeout -k 10 300 python bench.py --workload s2 --steps 200 --warmup 20 --no-cpu-baseline --no-batched 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid', $g, 's2 ms_per_step', d['ms_per_step'])"
done; done
