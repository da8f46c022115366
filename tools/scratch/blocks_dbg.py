import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import swf_renderer_amd as S
from swf_renderer_amd import distributed as D
from helpers import oracle_render, diff_stats
import scenarios
sc = scenarios.scenarios()["stroke_curves"]
w, h = sc["width"], sc["height"]
want = oracle_render(sc)
print("size", w, h)
for world in (3, 7):
    n = D.block_rows(h, world) * 16
    for mode in ("render", "edges"):
        for rank in range(world):
            rb = S.Renderer(w, h, band_index=rank, band_count=world, contiguous_bands=True)
            if mode == "render": rb.render(sc["stage"])
            else: rb.render_edges(*rb.build_frame(sc["stage"]))
            img = rb.read_image(premultiplied=True)
            slab = rb.band_slab()
            y0 = rank * n; y1 = min(h, y0 + n)
            a = img[y0:max(y1,y0)] != want[y0:max(y1,y0)]
            b = slab[:max(y1 - y0, 0)] != want[y0:max(y1, y0)]
            print(world, mode, rank, "image rows bad", np.nonzero(a.any(axis=(1, 2)))[0][:20] + y0, "slab rows bad", np.nonzero(b.any(axis=(1, 2)))[0][:20] + y0, "tail nonzero", bool(slab[y1 - y0:].any()))
            if a.any():
                ys, xs = np.nonzero(a.any(axis=2)); print("   x range", xs.min(), xs.max(), "count", len(xs))
            rb.close()
