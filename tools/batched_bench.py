"""The saturated-GPU measurements of bench.py on their own (for rocprofv3): the resident S1 scene, or S0 (one full-frame opaque
rectangle), as 8 frames per kernel launch.   usage (GPU box): python tools/batched_bench.py s1|s0 [launches]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWFR_FRAMES_IN_FLIGHT"] = "1"
import numpy as np
import swf_renderer_amd as S
from swf_renderer_amd import api, synth
which = sys.argv[1] if len(sys.argv) > 1 else "s1"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = synth.S1
W, H = cfg["width"], cfg["height"]
if which == "s1":
    pts, cols = synth.scene(**cfg)
    host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY)
    scene = host.build_frame(api.stars_to_stage(pts, cols)); host.close()
else:
    rect = np.array([[[0, 0], [W * 256, 0], [W * 256, H * 256], [0, H * 256]]], dtype=np.int32)
    scene = api.polygons_to_scene(rect, np.array([[30, 60, 90, 255]], dtype=np.uint8), W, H)
r = S.Renderer(W, H)
r.upload_edges(*scene)
r.render_resident(4)
ms = r.render_resident_batched(8, launches)
algo = 4 * W * H + 16 * len(scene[0]) + 16 * len(scene[1])
per = ms / (8 * launches)
print(json.dumps({"scene": which, "frames_per_launch": 8, "launches": launches, "ms_per_frame": round(per, 4), "algorithmic_bytes": algo,
                  "achieved_GBps": round(algo / (per * 1e-3) / 1e9, 1), "frac_of_8TBps": round(algo / (per * 1e-3) / 1e9 / 8000, 4)}))
r.close()
