"""Emulator check of swfr_render_batch's frames-per-launch path: the 'device' destination is host memory here."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
os.environ.setdefault("SWFR_BATCH_FRAMES", "3")
import run as emurun
emurun.use_emulator()
import numpy as np
import swf_renderer_amd as S
from helpers import fixture, oracle_render, diff_stats
from oracle import canvas_replay as cr
import scenarios
tag = fixture("homestuck-beta-29")
ratios = [k / 7 for k in range(8)]
stages = [cr.stage_for_morph_shape(tag, q)[2] for q in ratios]
w, h, _ = cr.stage_for_morph_shape(tag, 0.0)
r = S.Renderer(w, h)
out = np.zeros((len(stages), h, w, 4), dtype=np.uint8)
r.render_batch(stages, out.ctypes.data, h * w * 4)
bad = 0
for i, st in enumerate(stages):
    d = diff_stats(out[i], oracle_render(dict(width=w, height=h, stage=st)))
    print(i, d); bad += d != (0, 0)
SC = scenarios.scenarios()
for name in ("stroke_curves", "fixture_homestuck-beta-1", "translucent_stack"):
    sc = SC[name]
    r2 = S.Renderer(sc["width"], sc["height"])
    o2 = np.zeros((4, sc["height"], sc["width"], 4), dtype=np.uint8)
    r2.render_batch([sc["stage"]] * 4, o2.ctypes.data, sc["height"] * sc["width"] * 4)
    for i in range(4):
        d = diff_stats(o2[i], oracle_render(sc)); print(name, i, d); bad += d != (0, 0)
    r2.close()
r.close()
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
