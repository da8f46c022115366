#!/bin/bash
# round-2 quick GPU check: parity tests, then S1 bench under a few knobs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -15 $O/pytest.log
[ $rc -ne 0 ] && [ -z "$KEEP_GOING" ] && exit 1
run() { name=$1; shift; env "$@" python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python3 - "$name" "$O/$name.json" <<'P'
import json,sys
try:
    l=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], l["value"], l["frames_per_sec"], l["kernel_ms_per_frame"], l["roofline"].get("one_frame_in_flight"))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace(".json",".err")).read()[-400:])
P
}
for v in "$@"; do
  name=$(echo "$v" | tr ' =' '__')
  run "$name" $v
done
