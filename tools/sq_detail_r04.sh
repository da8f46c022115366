#!/bin/bash
# SQ counter detail of the S1 bench kernels (one frame in flight: each kernel alone on the GPU), two --pmc passes
# usage: bash tools/sq_detail_r04.sh <tag> [lib]
TAG=${1:-r04sq}; LIB=${2:-swf_renderer_amd/libswfr.so}
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; export SWFR_FRAMES_IN_FLIGHT=1
B="--steps 4 --warmup 2 --no-cpu-baseline --no-full-path --no-verify --no-batched"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $R/gpurun_out/${TAG}_a --output-format csv -- python3 $R/tools/bench_with_lib.py $LIB $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $R/gpurun_out/${TAG}_b --output-format csv -- python3 $R/tools/bench_with_lib.py $LIB $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY -d $R/gpurun_out/${TAG}_c --output-format csv -- python3 $R/tools/bench_with_lib.py $LIB $B > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
out=collections.defaultdict(dict)
for sub in "abc":
    fs=sorted(glob.glob('$R/gpurun_out/${TAG}_%s/*/*counter_collection.csv'%sub))
    if not fs: continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in agg.items():
        if k.startswith('swfr::k2_tiles') or k.startswith('swfr::k2_rows_b') or k.startswith('swfr::k2_bin'):
            for c,x in d.items(): out[k][c]=sum(x)/len(x)
for k,m in out.items():
    w=max(m.get('SQ_WAVES',1),1)
    print(k, 'waves %d'%w)
    for c in sorted(m):
        if c!='SQ_WAVES': print('   %-24s %14.0f   per wave %10.1f'%(c,m[c],m[c]/w))
PY
