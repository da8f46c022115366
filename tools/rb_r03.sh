#!/bin/bash
cd $GRAFT_REPO_ROOT
for rb in 0 2 4; do for k in "20 5" "20 5" "300 30"; do set -- $k
SWFR_RESIDENT_BATCH=$rb python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-full-path --no-batched 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident_batch $rb', 'K $1', l['value'], l['ms_per_step'], l['verified'])"
done; done
