"""What a short resident run costs beyond its frames: T(K) = time of render_resident(K) (which returns when the K frames are done) for
several K on scene S1, the bench's settings (no per-kernel events).  A straight line a + b K: b is the steady-state frame time, a the
price of filling and draining the four-frame pipeline plus the final synchronisation.     usage (GPU box): python tools/short_run_probe.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SWFR_EVENT_STRIDE", "1000000")
import numpy as np
import torch
assert torch.cuda.is_available()
import swf_renderer_amd as S
from swf_renderer_amd import api, synth
cfg = synth.S1
W, H = cfg["width"], cfg["height"]
pts, cols = synth.scene(**cfg)
host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY)
scene = host.build_frame(api.stars_to_stage(pts, cols)); host.close()
r = S.Renderer(W, H)
r.upload_edges(*scene)
r.render_resident(50)
out = {}
for K in (1, 2, 3, 4, 6, 8, 12, 20, 40, 100, 300):
    ts = []
    for rep in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.render_resident(K)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e6
    out[K] = {"us_p50": round(float(np.percentile(ts, 50)), 1), "us_min": round(float(ts.min()), 1), "per_frame_p50": round(float(np.percentile(ts, 50)) / K, 2)}
    print(K, out[K], flush=True)
# the same after the GPU has idled (a new handle's allocations, a host-side pause): clocks and queues start cold
for idle_ms in (1, 5, 50):
    ts = []
    for rep in range(12):
        torch.cuda.synchronize(); time.sleep(idle_ms * 1e-3)
        r.render_resident(5); torch.cuda.synchronize()          # the driver's warm-up
        t0 = time.perf_counter()
        r.render_resident(20); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[2:]) * 1e6
    print("20 frames after %d ms idle + 5 warm-up frames: p50 %.1f us (%.2f per frame), min %.1f" % (idle_ms, np.percentile(ts, 50), np.percentile(ts, 50) / 20, ts.min()), flush=True)
ks = np.array([20, 40, 100, 300], float); tv = np.array([out[int(k)]["us_p50"] for k in ks])
b, a = np.polyfit(ks, tv, 1)
print(json.dumps({"fit_over_K_20_to_300": {"fixed_us": round(float(a), 1), "per_frame_us": round(float(b), 2)}}))
r.close()
