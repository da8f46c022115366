#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration on the GPU box (run through gpurun):
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes, then SQ counters.
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{trace,fetch,write,sq}/ + gpurun_out/<tag>_bench.json
set -e
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 300 --warmup 30 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/${TAG}_sq --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
# the same trace with one frame in flight (kernel durations without overlap from the neighbouring frame)
export SWFR_FRAMES_IN_FLIGHT=1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace1 --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
unset SWFR_FRAMES_IN_FLIGHT
python3 $R/bench.py --steps 300 --warmup 30 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
tail -1 $R/gpurun_out/${TAG}_bench.json
