#!/bin/bash
# rocprofv3 evidence for the shaded tile kernel (config 4, one frame in flight): kernel stats, FETCH/WRITE PMC, SQ counters
TAG=${1:-r02shaded}; W=${2:-large}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SWFR_FRAMES_IN_FLIGHT=1
python3 $R/tools/shaded_bench.py $W 100 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || { tail -5 $R/gpurun_out/${TAG}_bench.err; exit 1; }
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_trace1 --output-format csv -- python3 $R/tools/shaded_bench.py $W 60 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch --output-format csv -- python3 $R/tools/shaded_bench.py $W 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write --output-format csv -- python3 $R/tools/shaded_bench.py $W 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/${TAG}_sq --output-format csv -- python3 $R/tools/shaded_bench.py $W 5 > /dev/null 2>&1
cat $R/gpurun_out/${TAG}_bench.json
