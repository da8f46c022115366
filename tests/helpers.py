import json
import os

import numpy as np

from oracle import canvas_replay as cr, oracle_backend as ob

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
FIX = os.path.join(GOLD, "fixtures")


def fixture(name):
    with open(os.path.join(FIX, name + ".ast.json")) as f:
        return json.load(f)


def fixture_text(name):
    with open(os.path.join(FIX, name)) as f:
        return f.read()


def golden(name, key):
    return np.load(os.path.join(GOLD, name + ".npz"))[key]


def oracle_render(sc):
    """premultiplied RGBA of a tests/scenarios.py scenario through the CPU restatement"""
    be = ob.OracleBackend(sc["width"], sc["height"])
    if sc.get("even_odd"):
        be.set_fill_rule(True)
    rp = cr.CanvasReplay(be, linear_extension=True)
    for b in sc.get("bitmaps", []):
        rp.add_bitmap(b)
    rp.render(sc["stage"])
    out = be.premultiplied_rgba()
    unsupported = be.unsupported
    be.close()
    assert not unsupported, "scenario uses stroker features outside the restated subset"
    return out


def product_render(sc, **kw):
    """premultiplied RGBA of a scenario through libswfr.so (HIP path)"""
    import swf_renderer_amd as S
    r = S.Renderer(sc["width"], sc["height"], even_odd=bool(sc.get("even_odd")), **kw)
    try:
        for b in sc.get("bitmaps", []):
            r.add_bitmap(b)
        r.render(sc["stage"])
        return r.read_image(premultiplied=True)
    finally:
        r.close()


def diff_stats(a, b):
    d = (a != b).any(-1)
    mx = int(np.abs(a.astype(int) - b.astype(int)).max()) if d.any() else 0
    return int(d.sum()), mx
