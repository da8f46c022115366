"""Where the time of a short resident call goes (the driver's command times 20 frames): wall clock of the call against the HIP-event
time between its first and last kernel, for several frame counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWFR_EVENT_STRIDE"] = "1000000"
import numpy as np, torch
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
for k in (1, 2, 4, 8, 20, 20, 40, 100, 300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.render_resident(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e6
    tm = r.timing()
    print("frames %4d  wall %8.1f us (%.1f per frame)   events first->last kernel %8.1f us (%.1f per frame)" % (k, dt, dt / k, tm["total_ms"] * 1e3, tm["total_ms"] * 1e3 / k))
r.close()
