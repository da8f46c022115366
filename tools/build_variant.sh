#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>  ->  build/<name>/libswfr.so   (tuning experiments)
name=$1; shift
mkdir -p /root/repo/build/$name
cd /root/repo/swf_renderer_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-function -DSWFR_BUILD "$@" raster2.hip renderer.cpp geometry.cpp shape_decoder.cpp frame_builder.cpp bitmap_decode.cpp -o /root/repo/build/$name/libswfr.so 2>&1 | grep -E "error" 
echo built $name
