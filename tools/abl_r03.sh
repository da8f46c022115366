#!/bin/bash
# ablation of k2_tiles (timing-only builds: wrong pixels): per variant the bench line's kernel times and the SQ instruction counts
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --no-full-path --no-verify --no-batched"
for v in "$@"; do
  lib=build/$v/libswfr.so; [ "$v" = base ] && lib=swf_renderer_amd/libswfr.so
  python3 $R/tools/bench_with_lib.py $lib --steps 200 --warmup 20 $B 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', l['value'], l['kernel_ms_per_frame'], l['roofline']['one_frame_in_flight'])"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/abl_$v --output-format csv -- python3 $R/tools/bench_with_lib.py $lib --steps 4 --warmup 2 $B > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob('$R/gpurun_out/abl_$v/*/*counter_collection.csv'))[-1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'swfr::k2_tiles' in k or 'swfr::k2_rows_b' in k or 'swfr::k2_bin' in k:
        m={c:sum(x)/len(x) for c,x in d.items()}
        w=max(m.get('SQ_WAVES',1),1)
        print('   $v', k[-18:], 'waves %d valu/launch %.3fM salu %.3fM lds %.3fM | per wave valu %.0f salu %.0f lds %.0f' % (w, m['SQ_INSTS_VALU']/1e6, m['SQ_INSTS_SALU']/1e6, m['SQ_INSTS_LDS']/1e6, m['SQ_INSTS_VALU']/w, m['SQ_INSTS_SALU']/w, m['SQ_INSTS_LDS']/w))
PY
done
