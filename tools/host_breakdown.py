import sys, time, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, swf_renderer_amd as S, scenarios
from helpers import fixture
m=scenarios._m
tag = fixture("homestuck-beta-29")
b, mb = tag["bounds"], tag["morph_bounds"]
x0, x1 = min(b["x_min"], mb["x_min"]), max(b["x_max"], mb["x_max"])
y0, y1 = min(b["y_min"], mb["y_min"]), max(b["y_max"], mb["y_max"])
sx, sy = 1920 * 20 / (x1 - x0), 1080 * 20 / (y1 - y0)
stages = [{"children": [{"type": "morph-shape", "definition": tag, "ratio": k / 255, "matrix": m(sx, sy, -x0 * sx, -y0 * sy)}]} for k in range(256)]
r=S.Renderer(1920,1080)
for st in stages[:3]: r.render(st)
t=time.perf_counter()
for st in stages: r.render(st)
print('render(stage) total %.1f us/frame'%((time.perf_counter()-t)/256*1e6))
# python marshalling only
t=time.perf_counter()
for st in stages: c=r._marshal_stage(st) if hasattr(r,'_marshal_stage') else None
print('marshal %.1f us/frame'%((time.perf_counter()-t)/256*1e6))
t=time.perf_counter()
for st in stages: e,p,s=r.build_frame(st)
print('build_frame (incl marshal + numpy copies) %.1f us/frame'%((time.perf_counter()-t)/256*1e6))
t=time.perf_counter()
for _ in range(256): r.upload_edges(e,p,s)
print('upload_edges %.1f us/frame'%((time.perf_counter()-t)/256*1e6))
t=time.perf_counter()
for _ in range(256): r.render_resident(1)
print('render_resident(1) %.1f us/frame'%((time.perf_counter()-t)/256*1e6))
