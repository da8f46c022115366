#!/bin/bash
cd $GRAFT_REPO_ROOT
for fl in 3 4; do for nb in "" "--no-batched"; do for rep in 1 2; do
SWFR_FRAMES_IN_FLIGHT=$fl python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-full-path --no-verify $nb 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in_flight $fl', '$nb', l['value'], l['ms_per_step'])"
done; done; done
