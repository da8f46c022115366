import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWFR_TILES_DEBUG"] = "8"
import swf_renderer_amd as S
from swf_renderer_amd import api, synth
pts, cols = synth.scene(**synth.S1); W, H = 3840, 2160
edges, paths, styles = api.polygons_to_scene(synth.twips_to_fixed(pts), cols, W, H)
r = S.Renderer(W, H); r.upload_edges(edges, paths, styles); r.render_resident(3)
img = r.read_image(True)
def word(k):
    t = img[::16, k::64].copy().view(np.uint32).reshape(135, 60)   # k-th pixel of each tile (RGBA bytes -> swizzled u32)
    # undo the ARGB->RGBA byte swizzle: value was stored via rgba = (p & 0xff00ff00) | ((p>>16)&0xff) | ((p&0xff)<<16)
    return ((t & 0xff00ff00) | ((t >> 16) & 0xff) | ((t & 0xff) << 16)).astype(np.float64)
p, pairs, recs = word(0), word(1), word(2)
A = np.stack([np.ones(p.size), pairs.ravel(), recs.ravel()], 1)
coef, *_ = np.linalg.lstsq(A, p.ravel(), rcond=None)
print("fit clocks = %.0f + %.0f * pairs + %.1f * records;  pairs/tile mean %.2f max %d, records/tile mean %.1f max %d" % (coef[0], coef[1], coef[2], pairs.mean(), pairs.max(), recs.mean(), recs.max()))
print("tiles", p.size, "clocks: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f  sum/4096 %.0f" % (p.mean(), np.median(p), np.percentile(p, 90), np.percentile(p, 99), p.max(), p.sum() / 4096))
print(r.timing())
hist, edges_ = np.histogram(p, bins=12); print(hist, edges_.astype(int))
