"""Per-strip cost distribution of k_tiles on S1 (diagnostic; SWFR_TILES_DEBUG=8 makes every wave write its
s_memtime delta (100 MHz ticks), partial-path count and record count into the first three pixels of its strip).
usage (GPU box): python tools/strip_times.py [strip_h]"""
import os, sys
os.environ["SWFR_TILES_DEBUG"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from swf_renderer_amd import synth
from swf_renderer_amd.api import Renderer, polygons_to_scene

strip_h = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = 3840, 2160
pts, colors = synth.scene(**synth.S1)
r = Renderer(W, H)
edges, paths, styles = polygons_to_scene(synth.twips_to_fixed(pts), colors, W, H)
r.upload_edges(edges, paths, styles)
r.render_resident(3)
img = r.read_image(premultiplied=True).view(np.uint32).reshape(H, W)
def unswap(p):
    return (p & 0xff00ff00) | ((p >> 16) & 0xff) | ((p & 0xff) << 16)
t = unswap(img[0::strip_h, 0::64]).astype(np.int64)
pairs = unswap(img[0::strip_h, 1::64]).astype(np.int64)
recs = unswap(img[0::strip_h, 2::64]).astype(np.int64)
us = t / 100.0
print("strips", us.size, "mean us %.2f" % us.mean(), "p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(us, [50, 90, 99, 100])))
print("sum of strip times / 4096 slots: %.1f us" % (us.sum() / 4096))
for lo, hi in [(0, 1), (1, 2), (2, 3), (3, 5), (5, 9), (9, 99)]:
    m = (pairs >= lo) & (pairs < hi)
    if m.any():
        print("partial paths %d..%d: %6d strips, mean %.2f us, mean records %.1f" % (lo, hi - 1, m.sum(), us[m].mean(), recs[m].mean()))
