#!/bin/bash
# rocprofv3 evidence for the S1 bench on the GPU box (gpurun): kernel trace + stats (default frames in flight, and one frame in
# flight), FETCH_SIZE and WRITE_SIZE in their own --pmc passes, SQ counters in a third; the bench line itself.
# usage: bash tools/profile_r02.sh <tag> [bench args]   -> gpurun_out/<tag>_{trace,trace1,fetch,write,sq}/ + gpurun_out/<tag>_bench.json
TAG=${1:-prof}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 300 --warmup 30 "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || { tail -5 $R/gpurun_out/${TAG}_bench.err; exit 1; }
B="--no-cpu-baseline --no-full-path --no-verify --no-batched"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 $B "$@" > /dev/null 2>&1
# (the counter passes: one frame in flight and one frame per launch, so that "per launch" is per frame and a k2_tiles wavefront is one
#  strip -- with frames overlapping the default is two frames per launch and two strips per wavefront)
export SWFR_RESIDENT_BATCH=1
export SWFR_FRAMES_IN_FLIGHT=1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 $B "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 $B "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/${TAG}_sq --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 $B "$@" > /dev/null 2>&1
unset SWFR_RESIDENT_BATCH
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace1 --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 $B "$@" > /dev/null 2>&1
unset SWFR_FRAMES_IN_FLIGHT
tail -c 1500 $R/gpurun_out/${TAG}_bench.json
