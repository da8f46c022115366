#!/bin/bash
# k2_tiles launch shape sweep: S1 / S0 / S2 bench lines per SWFR_TILES_GRID (0 = one wavefront per strip).   usage: bash tools/grid_sweep.sh <tag> [grids...]
TAG=${1:-g}; shift
R=$GRAFT_REPO_ROOT; cd $R
for g in ${@:-0 4096 5120 8192}; do
  for wl in s1 s2; do
    SWFR_TILES_GRID=$g timeout -k 10 300 python bench.py --workload $wl --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_grid${g}_$wl.json 2>/dev/null
    python3 - "$g" "$wl" gpurun_out/${TAG}_grid${g}_$wl.json <<'PY'
import json, sys
g, wl, f = sys.argv[1:4]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print("grid", g, wl, "ms_per_step", d["ms_per_step"], "kernels", d.get("kernel_ms_per_frame", {}).get("k2_tiles"), d.get("kernel_ms_per_frame", {}).get("k2_rows"), "batched", (r.get("batched") or {}).get("ms_per_frame"), "s0 tiles", (r.get("s0") or {}).get("k2_tiles_ms"), "s0 batched", ((r.get("s0") or {}).get("batched") or {}).get("ms_per_frame"))
except Exception as e:
    print("grid", g, wl, "failed", e)
PY
  done
done
