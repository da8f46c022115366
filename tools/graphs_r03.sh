#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 1 0; do for k in "20 5" "20 5" "300 30"; do set -- $k
SWFR_GRAPHS=$g python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-full-path --no-batched 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graphs $g', 'K $1', l['value'], l['ms_per_step'], l['verified'])"
done; done
