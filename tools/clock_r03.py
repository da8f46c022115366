"""Does a short resident call depend on what the GPU did just before it?  20 frames after an idle gap against 20 frames right behind
200 frames (clock ramp)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWFR_EVENT_STRIDE"] = "1000000"
import torch
import swf_renderer_amd as S
from swf_renderer_amd import api, synth
cfg = synth.S1
W, H = cfg["width"], cfg["height"]
pts, cols = synth.scene(**cfg)
host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY)
scene = host.build_frame(api.stars_to_stage(pts, cols)); host.close()
r = S.Renderer(W, H)
r.upload_edges(*scene)
r.render_resident(5)
def timed(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.render_resident(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6, r.timing()["total_ms"] * 1e3
for gap_ms in (0, 1, 10, 50, 200):
    for pre in (0, 5, 200):
        res = []
        for rep in range(3):
            time.sleep(gap_ms / 1e3)
            if pre:
                r.render_resident(pre)
            res.append(timed(20))
        print("idle %3d ms, then %3d untimed frames, then 20 timed: wall %s us, events %s us" % (gap_ms, pre, " ".join("%.0f" % a for a, _ in res), " ".join("%.0f" % b for _, b in res)))
r.close()
