#!/bin/bash
# round-2 exploration: baseline numbers of the round-1 kernels under their knobs (gpurun)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/explore
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -5 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
run() { name=$1; shift; env "$@" python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python3 - "$name" "$O/$name.json" <<'P'
import json,sys
try:
    l=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], l["value"], l["frames_per_sec"], l["kernel_ms_per_frame"], l["roofline"].get("one_frame_in_flight"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
run base A=1
run fif1 SWFR_FRAMES_IN_FLIGHT=1
run fif3 SWFR_FRAMES_IN_FLIGHT=3
run fif4 SWFR_FRAMES_IN_FLIGHT=4
run rs8 SWFR_CHUNK_ROWS=8
run rs8_fif1 SWFR_CHUNK_ROWS=8 SWFR_FRAMES_IN_FLIGHT=1
run c16 SWFR_CHUNK_ROWS=16
run c32 SWFR_CHUNK_ROWS=32
run nofuse SWFR_FUSED_CLASS=0 SWFR_FRAMES_IN_FLIGHT=1
run fuse1 SWFR_FUSED_CLASS=1 SWFR_FRAMES_IN_FLIGHT=1
run noorder SWFR_STRIP_ORDER=0 SWFR_FRAMES_IN_FLIGHT=1
run dbg1 SWFR_TILES_DEBUG=1 SWFR_FRAMES_IN_FLIGHT=1
run dbg2 SWFR_TILES_DEBUG=2 SWFR_FRAMES_IN_FLIGHT=1
run dbg3 SWFR_TILES_DEBUG=3 SWFR_FRAMES_IN_FLIGHT=1
run dbg4 SWFR_TILES_DEBUG=4 SWFR_FRAMES_IN_FLIGHT=1
run dbg11 SWFR_TILES_DEBUG=11 SWFR_FRAMES_IN_FLIGHT=1
run dbg12 SWFR_TILES_DEBUG=12 SWFR_FRAMES_IN_FLIGHT=1
run dbg13 SWFR_TILES_DEBUG=13 SWFR_FRAMES_IN_FLIGHT=1
SWFR_TILES_DEBUG=9 SWFR_FRAMES_IN_FLIGHT=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep swfr | tail -1
python3 bench.py --workload s2 --steps 60 --warmup 10 --no-cpu-baseline | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s2', l['value'], l['frames_per_sec'], l['kernel_ms_per_frame'])"
python3 tools/config_bench.py > $O/config_bench.txt 2>&1; tail -12 $O/config_bench.txt
