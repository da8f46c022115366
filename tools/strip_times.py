"""Per-strip cost distribution of k_tiles on S1 (diagnostic; SWFR_TILES_DEBUG=8 makes every wave write its
s_memtime delta (100 MHz ticks), partial-path count and record count into the first three pixels of its strip).
The wall-clock timeline and the per-phase clocks need a library built with SWFR_PHASES=1 (python -m swf_renderer_amd.build).
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
wall = unswap(img[0::strip_h, 3::64]).astype(np.int64).ravel()
print("s_memtime ticks per 100 MHz s_memrealtime tick: %.2f" % (t.sum() / max(wall.sum(), 1)))
w0 = unswap(img[0::strip_h, 4::64]).astype(np.int64).ravel() | (unswap(img[0::strip_h, 5::64]).astype(np.int64).ravel() << 32)
print("wall-clock span of the kernel: %.1f us (first start to last end)" % (((w0 + wall).max() - w0.min()) / 100.0))
t0 = (w0 - w0.min()).astype(np.float64); t1 = t0 + wall
span = t1.max()
print("average concurrency %.0f waves (%.2f per SIMD)" % ((t1 - t0).sum() / span, (t1 - t0).sum() / span / 1024))
edges_t = np.linspace(0, span, 21)
for a, b in zip(edges_t[:-1], edges_t[1:]):
    mid = (a + b) / 2
    print("  t=%3.0f%%: %5d waves resident, %5d started in slice" % (100 * mid / span, ((t0 <= mid) & (t1 > mid)).sum(), ((t0 >= a) & (t0 < b)).sum()))
us = t / 2280.0
print("strips", us.size, "mean us %.2f" % us.mean(), "p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(us, [50, 90, 99, 100])))
print("sum of strip times / 4096 slots: %.1f us" % (us.sum() / 4096))
for lo, hi in [(0, 1), (1, 2), (2, 3), (3, 5), (5, 9), (9, 99)]:
    m = (pairs >= lo) & (pairs < hi)
    if m.any():
        print("partial paths %d..%d: %6d strips, mean %.2f us, mean records %.1f" % (lo, hi - 1, m.sum(), us[m].mean(), recs[m].mean()))

names = ["bin+walk", "batch headers", "stage records", "accumulate", "scan+blend"]
ph = [unswap(img[0::strip_h, 8 + k::64]).astype(np.int64) / 2280.0 for k in range(5)]
print("phase means (us):", ", ".join("%s %.2f" % (n, v.mean()) for n, v in zip(names, ph)), "| total %.2f" % us.mean())
heavy = pairs >= 5
print("strips with >= 5 partial paths:", ", ".join("%s %.2f" % (n, v[heavy].mean()) for n, v in zip(names, ph)), "| total %.2f, paths %.1f" % (us[heavy].mean(), pairs[heavy].mean()))
