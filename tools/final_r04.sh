#!/bin/bash
# The round's evidence in one gpurun call, taken from the build that is in the tree: S1 profile (kernel trace in both modes, FETCH /
# WRITE PMC passes, SQ counters), the driver's command, S2, shaded-kernel profiles, configs 2-4, workgroup timelines, per-block
# timings, the saturated S1 / S0 runs under rocprofv3, the N>1 bench path over gloo in both assemblies, the store-shape microbenchmark.
# usage: bash tools/final_r04.sh <tag>       then, in the container:  python tools/summarize_r04.py <tag>
TAG=${1:-r04z}
R=$GRAFT_REPO_ROOT; cd $R
git rev-parse --short HEAD > gpurun_out/${TAG}_commit.txt 2>/dev/null || true
sha256sum swf_renderer_amd/libswfr.so | cut -c1-16 > gpurun_out/${TAG}_lib_sha16.txt
bash tools/profile_r02.sh $TAG > /dev/null 2>&1; tail -c 200 gpurun_out/${TAG}_bench.json; echo
cd $R
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_command.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_driver_command.json
bash tools/sq_detail_r04.sh ${TAG}_sqd > gpurun_out/${TAG}_sq_detail.txt 2>&1
cd $R
bash tools/profile_shaded.sh ${TAG}_large large > /dev/null 2>&1
bash tools/profile_shaded.sh ${TAG}_magnified magnified > /dev/null 2>&1
cd $R
timeout -k 10 300 python tools/config_bench.py > gpurun_out/${TAG}_config_bench.txt 2>&1; tail -2 gpurun_out/${TAG}_config_bench.txt | cut -c1-300
bash tools/build_variant.sh trace -DSWFR_TRACE > /dev/null 2>&1
for w in s1 s2; do TRACE_BUILD=trace timeout -k 10 120 python tools/trace_wg.py $w 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_wg_timeline_$w.txt; done
for w in config3 config2h; do TRACE_BUILD=trace SWFR_CHUNK_ROWS=16 timeout -k 10 120 python tools/trace_wg.py $w 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_wg_timeline_$w.txt; done
timeout -k 10 300 python tools/short_run_probe.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_short_run.txt
SWFR_FRAMES_IN_FLIGHT=1 timeout -k 10 300 python tools/pipeline_timing.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_blocks_timing.json
timeout -k 10 300 python bench.py --workload s2 --steps 100 --warmup 10 --no-cpu-baseline --no-batched > gpurun_out/${TAG}_bench_s2.json 2>/dev/null; cut -c1-200 gpurun_out/${TAG}_bench_s2.json
for a in rotate root; do
  timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --assembly $a --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_gloo_2ranks_one_gpu_$a.json 2> gpurun_out/${TAG}_bench_gloo2_$a.err
  grep -h '^{"metric"' gpurun_out/${TAG}_bench_gloo_2ranks_one_gpu_$a.json | cut -c1-200
done
bash tools/build_variant.sh stats -DSWFR_TSTATS > /dev/null 2>&1
timeout -k 10 120 python tools/tile_stats.py s1 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_tile_stats_s1.txt
./build/store_shape > gpurun_out/${TAG}_store_shape.txt 2>&1; cat gpurun_out/${TAG}_store_shape.txt
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
