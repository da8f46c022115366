#!/bin/bash
# A short GPU check of a kernel change: parity tests, S1 / S2 bench lines, the k2_bin workgroup timelines.   usage: bash tools/quick_r04.sh <tag>
TAG=${1:-q}
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.txt 2>&1; tail -2 gpurun_out/${TAG}_tests.txt
timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2>/dev/null; cut -c1-160 gpurun_out/${TAG}_bench.json
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench20.json 2>/dev/null; cut -c1-160 gpurun_out/${TAG}_bench20.json
timeout -k 10 300 python bench.py --workload s2 --steps 100 --warmup 10 --no-cpu-baseline --no-batched > gpurun_out/${TAG}_bench_s2.json 2>/dev/null; cut -c1-160 gpurun_out/${TAG}_bench_s2.json
bash tools/build_variant.sh trace -DSWFR_TRACE > /dev/null 2>&1
for w in s1 s2; do TRACE_BUILD=trace timeout -k 10 120 python tools/trace_wg.py $w 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_wg_timeline_$w.txt; sed -n 1,3p gpurun_out/${TAG}_wg_timeline_$w.txt | cut -c1-260; done
