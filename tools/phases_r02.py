"""k2_rows phase clocks of S1 from the -DSWFR_PHASES build (build/phases/libswfr.so), gpurun."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWFR_PRINT_PHASES"] = "1"
os.environ["SWFR_FRAMES_IN_FLIGHT"] = "1"
import torch
assert torch.cuda.is_available()
from swf_renderer_amd import api
api.library_path = lambda: os.path.join(ROOT, "build", os.environ.get("PHASES_BUILD", "phases"), "libswfr.so")
import swf_renderer_amd as S
from swf_renderer_amd import synth
import numpy as np
which = sys.argv[1] if len(sys.argv) > 1 else "s1"
if which in ("s1", "s2"):
    cfg = synth.S1 if which == "s1" else synth.S2
    pts, cols = synth.scene(**cfg)
    fx = synth.twips_to_fixed(pts)
    edges, paths, styles = api.polygons_to_scene(fx, cols, cfg["width"], cfg["height"])
    r = S.Renderer(cfg["width"], cfg["height"])
else:                                                   # a scenario of tests/scenarios.py, e.g. config2_homestuck-beta-1
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenarios
    sc = scenarios.scenarios()[which]
    r = S.Renderer(sc["width"], sc["height"])
    edges, paths, styles = r.build_frame(sc["stage"])
r.upload_edges(edges, paths, styles)
r.render_resident(20)
r.render_resident(50)
print(r.timing())
r.close()
