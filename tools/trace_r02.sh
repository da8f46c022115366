#!/bin/bash
# kernel trace of the S1 bench, one frame in flight (gpurun): per-kernel average durations
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_now
SWFR_FRAMES_IN_FLIGHT=${FIF:-1} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_now --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<P
import csv, glob
f = glob.glob('$R/gpurun_out/trace_now/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('void ', '')
    if 'swfr' in n: print('%-40s calls %5s avg %8.2f us  min %8.2f max %8.2f' % (n, r['Calls'], float(r['AverageNs'])/1000, float(r['MinNs'])/1000, float(r['MaxNs'])/1000))
P
