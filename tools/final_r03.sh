#!/bin/bash
# The round's evidence in one gpurun call: S1 profile (trace, PMC, SQ), shaded-kernel profiles, configs 2-4, workgroup timelines,
# per-block timings for the multi-GPU note, the N>1 bench path rehearsed over gloo (self-launched), the driver's command.
# usage: bash tools/final_r03.sh <tag>
TAG=${1:-r03q}
R=$GRAFT_REPO_ROOT; cd $R
bash tools/profile_r02.sh $TAG > /dev/null 2>&1; tail -c 300 gpurun_out/${TAG}_bench.json; echo
cd $R
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_command.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_driver_command.json
bash tools/profile_shaded.sh ${TAG}_large large > /dev/null 2>&1
bash tools/profile_shaded.sh ${TAG}_magnified magnified > /dev/null 2>&1
cd $R
timeout -k 10 300 python tools/config_bench.py > gpurun_out/${TAG}_config_bench.txt 2>&1; tail -2 gpurun_out/${TAG}_config_bench.txt | cut -c1-300
bash tools/build_variant.sh trace -DSWFR_TRACE > /dev/null 2>&1
for w in s1 s2; do TRACE_BUILD=trace timeout -k 10 120 python tools/trace_wg.py $w 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_wg_timeline_$w.txt; done
SWFR_FRAMES_IN_FLIGHT=1 timeout -k 10 300 python tools/pipeline_timing.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_blocks_timing.json
timeout -k 10 300 python bench.py --workload s2 --steps 100 --warmup 10 --no-cpu-baseline --no-batched > gpurun_out/${TAG}_bench_s2.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_s2.json
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_gloo_2ranks_one_gpu.json 2> gpurun_out/${TAG}_bench_gloo2.err; cut -c1-200 gpurun_out/${TAG}_bench_gloo_2ranks_one_gpu.json
bash tools/build_variant.sh stats -DSWFR_TSTATS > /dev/null 2>&1
timeout -k 10 120 python tools/tile_stats.py s1 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_tile_stats_s1.txt
# the saturated (8 frames per launch) S1 and the S0 store-roof probe under rocprofv3: kernel stats, FETCH / WRITE in their own passes
cd /tmp && export TMPDIR=/tmp
for w in s1 s0; do
  python3 $R/tools/batched_bench.py $w > $R/gpurun_out/${TAG}_batched_${w}.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_batched_${w}_trace --output-format csv -- python3 $R/tools/batched_bench.py $w 6 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_batched_${w}_fetch --output-format csv -- python3 $R/tools/batched_bench.py $w 2 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_batched_${w}_write --output-format csv -- python3 $R/tools/batched_bench.py $w 2 > /dev/null 2>&1
  cat $R/gpurun_out/${TAG}_batched_${w}.json
done
cd $R
