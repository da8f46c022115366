#!/bin/bash
# The round's evidence in one gpurun call: S1 profile (trace, PMC, SQ), shaded-kernel profiles, configs 2-4, workgroup timelines,
# per-block timings for the multi-GPU note, soak with row-kernel statistics, the N>1 bench path rehearsed over gloo.
# usage: bash tools/final_r02.sh <tag>
TAG=${1:-r02e}
R=$GRAFT_REPO_ROOT; cd $R
bash tools/profile_r02.sh $TAG > /dev/null 2>&1; tail -c 400 gpurun_out/${TAG}_bench.json; echo
bash tools/profile_shaded.sh ${TAG}_large large > /dev/null 2>&1
bash tools/profile_shaded.sh ${TAG}_magnified magnified > /dev/null 2>&1
cd $R
timeout -k 10 300 python tools/config_bench.py > gpurun_out/${TAG}_config_bench.txt 2>&1; tail -3 gpurun_out/${TAG}_config_bench.txt
for w in s1 s2; do TRACE_BUILD=trace timeout -k 10 120 python tools/trace_wg.py $w 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_wg_timeline_$w.txt; done
SWFR_FRAMES_IN_FLIGHT=1 timeout -k 10 300 python tools/pipeline_timing.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_blocks_timing.json
timeout -k 10 300 python bench.py --workload s2 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench_s2.json 2>/dev/null; cut -c1-330 gpurun_out/${TAG}_bench_s2.json
timeout -k 10 300 python tools/soak.py gpu 150 777 > gpurun_out/${TAG}_soak.txt 2>&1
SOAK_LONG=1 timeout -k 10 300 python tools/soak.py gpu 60 778 >> gpurun_out/${TAG}_soak.txt 2>&1
SOAK_BIG=1 timeout -k 10 300 python tools/soak.py gpu 40 779 >> gpurun_out/${TAG}_soak.txt 2>&1
grep -v "^\[" gpurun_out/${TAG}_soak.txt | tail -8
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 40 --warmup 5 > gpurun_out/${TAG}_bench_gloo2.log 2>&1; tail -1 gpurun_out/${TAG}_bench_gloo2.log | cut -c1-600
