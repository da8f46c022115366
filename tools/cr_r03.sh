cd $GRAFT_REPO_ROOT
for cr in 64 32 16; do
SWFR_CHUNK_ROWS=$cr python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-full-path --no-verify 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk_rows $cr', l['value'], l['kernel_ms_per_frame'], l['roofline']['one_frame_in_flight'])"
done
