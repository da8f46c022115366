cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in base bwait; do
  lib=build/$v/libswfr.so; [ "$v" = base ] && lib=swf_renderer_amd/libswfr.so
  python3 tools/bench_with_lib.py $lib --steps 20 --warmup 5 --no-cpu-baseline --no-full-path --no-verify --no-batched 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', l['value'], l['ms_per_step'])"
done; done
