"""What torch.cuda.synchronize() (hipDeviceSynchronize) costs on an idle device, by the number of live renderer handles (each owns up
to four non-blocking streams).  The bench contract brackets the timed region with it.    usage (GPU box): python tools/sync_cost_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
assert torch.cuda.is_available()
import swf_renderer_amd as S
from swf_renderer_amd import api, synth
def cost(label):
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-40s torch.cuda.synchronize() idle: p50 %.1f us, p90 %.1f us" % (label, np.percentile(ts, 50) * 1e6, np.percentile(ts, 90) * 1e6), flush=True)
torch.zeros(1, device="cuda"); cost("torch alone")
cfg = synth.S1; W, H = cfg["width"], cfg["height"]
pts, cols = synth.scene(**cfg)
host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY); scene = host.build_frame(api.stars_to_stage(pts, cols)); host.close()
hs = []
for n in range(1, 4):
    r = S.Renderer(W, H); r.upload_edges(*scene); r.render_resident(8); hs.append(r)
    cost("%d handle(s), used" % n)
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); hs[0].render_resident(20); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    ts = np.array(ts[5:]) * 1e6
    print("   render_resident(20) %.1f us, the synchronize() behind it %.1f us" % (np.percentile(ts[:, 0], 50), np.percentile(ts[:, 1], 50)), flush=True)
for r in hs: r.close()
cost("handles closed")
