"""Work counts of k2_tiles per strip from the -DSWFR_TSTATS build (build/stats/libswfr.so): band entries of the tile-row, entries
above the last opaque cover, entries walked, partial (path, strip) pairs, cells fetched for them.  One frame in flight, scene S1 (argv[1] = s2 for S2).   gpurun: bash tools/build_variant.sh stats -DSWFR_TSTATS && python tools/tile_stats.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWFR_FRAMES_IN_FLIGHT"] = "1"
import numpy as np
from swf_renderer_amd import api
LIB = os.path.join(ROOT, "build", "stats", "libswfr.so")
api.library_path = lambda: LIB
import swf_renderer_amd as S
from swf_renderer_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "s1"
cfg = synth.S1 if which == "s1" else synth.S2
W, H = cfg["width"], cfg["height"]
pts, cols = synth.scene(**cfg)
host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY)
scene = host.build_frame(api.stars_to_stage(pts, cols)); host.close()
r = S.Renderer(W, H)
r.upload_edges(*scene)
r.render_resident(3)
r.render_resident(1)
L = ctypes.CDLL(LIB)
buf = np.zeros((3, 32768, 8), dtype=np.uint32)
L.swfr_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.swfr_debug_trace(buf.ctypes.data, buf.nbytes) > 0
b = buf[2][buf[2][:, 7] != 0].astype(np.int64)
names = ["class bytes scanned (64 per round)", "non-empty entries above the last opaque cover", "entries walked", "partial (path, strip) pairs",
         "cells of the pairs' first row (lane 0)"]
print("%s: %d strips recorded" % (which, len(b)))
for i, n in enumerate(names):
    c = b[:, i]
    print("  %-40s total %9d  mean %7.2f  p50 %5d  p90 %5d  p99 %5d  max %6d  strips with none %6d" % (n, c.sum(), c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max(), int((c == 0).sum())))
pairs = b[:, 3]
print("  cells per (pair, row) %.1f" % (b[:, 4].sum() / max(pairs.sum(), 1)))
print("  strips without a partial pair: %d of %d; of those with no entry walked at all: %d" % (int((pairs == 0).sum()), len(b), int(((pairs == 0) & (b[:, 2] == 0)).sum())))
