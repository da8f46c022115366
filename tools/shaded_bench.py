"""The bitmap-fill workloads of BASELINE.json config 4 as a resident-scene loop (for rocprofv3): `magnified` = the reference's
textured fixture scaled to 3840x2160 (139x208 texture), `large` = a 4096x4096 texture sampled about 1:1 (HBM-bound sampling).
usage (GPU box): python tools/shaded_bench.py magnified|large [frames]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
assert torch.cuda.is_available()
import swf_renderer_amd as S
import scenarios
from helpers import fixture, large_texture_scene

which = sys.argv[1] if len(sys.argv) > 1 else "magnified"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
if which == "large":
    sc = large_texture_scene()
    stage, bitmaps, texels = sc["stage"], sc["bitmaps"], 4096 * int(2160 * 4096 / 3840) * 4
else:
    tag4 = fixture("homestuck-beta-4")
    b = tag4["bounds"]
    sx, sy = 3840 * 20 / (b["x_max"] - b["x_min"]), 2160 * 20 / (b["y_max"] - b["y_min"])
    stage = {"children": [{"type": "shape", "definition": tag4, "matrix": scenarios._m(sx, sy, -b["x_min"] * sx, -b["y_min"] * sy)}]}
    bitmaps, texels = [fixture("homestuck-beta-3.bitmap")], 139 * 208 * 4
r = S.Renderer(3840, 2160)
for bm in bitmaps:
    r.add_bitmap(bm)
edges, paths, styles = r.build_frame(stage)
r.upload_edges(edges, paths, styles)
r.render_resident(10)
r.render_resident(frames)
tm = r.timing()
n = max(tm["timed_frames"], 1)
algo = 4 * 3840 * 2160 + texels
tiles_us = tm["tiles_ms"] * 1e3 / n
print(json.dumps({"workload": "config 4 (%s texture) @ 3840x2160" % which, "frames_in_flight": int(os.environ.get("SWFR_FRAMES_IN_FLIGHT", "3")),
                  "frames_per_sec": round(tm["frames"] / (tm["total_ms"] * 1e-3), 1), "k2_bin_us": round(tm["setup_ms"] * 1e3 / n, 1),
                  "k2_rows_us": round(tm["rows_ms"] * 1e3 / n, 1), "k2_tiles_us": round(tiles_us, 1), "algorithmic_bytes": algo,
                  "achieved_GBps": round(algo / tiles_us / 1e3, 1), "frac_of_8TBps": round(algo / tiles_us / 1e3 / 8000, 4)}))
r.close()
