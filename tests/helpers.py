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


def product_render(sc, stats=None, **kw):
    """premultiplied RGBA of a scenario through libswfr.so (HIP path); `stats` (a dict) collects the handle's swfr_stats"""
    import swf_renderer_amd as S
    r = S.Renderer(sc["width"], sc["height"], even_odd=bool(sc.get("even_odd")), **kw)
    try:
        for b in sc.get("bitmaps", []):
            r.add_bitmap(b)
        try:
            r.render(sc["stage"])
        finally:
            if stats is not None:
                for k, v in r.stats().items():
                    stats[k] = stats.get(k, 0) + v
        return r.read_image(premultiplied=True)
    finally:
        r.close()


def diff_stats(a, b):
    d = (a != b).any(-1)
    mx = int(np.abs(a.astype(int) - b.astype(int)).max()) if d.any() else 0
    return int(d.sum()), mx


# ---- the reference's own acceptance metric (ts/src/test/node-canvas-renderer.spec.ts:54-65: pixelmatch, threshold 0.05,
#      anti-aliased pixels skipped, at most 1e-4 of the pixels may differ) restated for reporting
def _pm_delta(p1, p2, y_only=False):
    r1, g1, b1, a1 = (float(v) for v in p1)
    r2, g2, b2, a2 = (float(v) for v in p2)
    if (r1, g1, b1, a1) == (r2, g2, b2, a2):
        return 0.0
    if a1 < 255:
        a1 /= 255; r1, g1, b1 = 255 + (r1 - 255) * a1, 255 + (g1 - 255) * a1, 255 + (b1 - 255) * a1
    if a2 < 255:
        a2 /= 255; r2, g2, b2 = 255 + (r2 - 255) * a2, 255 + (g2 - 255) * a2, 255 + (b2 - 255) * a2
    y = (r1 * 0.29889531 + g1 * 0.58662247 + b1 * 0.11448223) - (r2 * 0.29889531 + g2 * 0.58662247 + b2 * 0.11448223)
    if y_only:
        return y
    i = (r1 * 0.59597799 - g1 * 0.27417610 - b1 * 0.32180189) - (r2 * 0.59597799 - g2 * 0.27417610 - b2 * 0.32180189)
    q = (r1 * 0.21147017 - g1 * 0.52261711 + b1 * 0.31113672) - (r2 * 0.21147017 - g2 * 0.52261711 + b2 * 0.31113672)
    return 0.5053 * y * y + 0.299 * i * i + 0.1957 * q * q


def _pm_many_siblings(img, x1, y1):
    h, w = img.shape[:2]
    zeroes = 1 if (x1 == 0 or y1 == 0 or x1 == w - 1 or y1 == h - 1) else 0
    for x in range(max(x1 - 1, 0), min(x1 + 1, w - 1) + 1):
        for y in range(max(y1 - 1, 0), min(y1 + 1, h - 1) + 1):
            if (x, y) != (x1, y1) and (img[y, x] == img[y1, x1]).all():
                zeroes += 1
                if zeroes > 2:
                    return True
    return False


def _pm_antialiased(img, x1, y1, img2):
    h, w = img.shape[:2]
    zeroes = 1 if (x1 == 0 or y1 == 0 or x1 == w - 1 or y1 == h - 1) else 0
    mn = mx = 0.0
    mnp = mxp = None
    for x in range(max(x1 - 1, 0), min(x1 + 1, w - 1) + 1):
        for y in range(max(y1 - 1, 0), min(y1 + 1, h - 1) + 1):
            if (x, y) == (x1, y1):
                continue
            d = _pm_delta(img[y1, x1], img[y, x], True)
            if d == 0:
                zeroes += 1
                if zeroes > 2:
                    return False
            elif d < mn:
                mn, mnp = d, (x, y)
            elif d > mx:
                mx, mxp = d, (x, y)
    if mn == 0 or mx == 0:
        return False
    return ((_pm_many_siblings(img, *mnp) and _pm_many_siblings(img2, *mnp)) or
            (_pm_many_siblings(img, *mxp) and _pm_many_siblings(img2, *mxp)))


def pixelmatch_count(a, b, threshold=0.05):
    """Number of mismatched pixels as the reference's spec counts them (straight RGBA8 images of equal shape)."""
    assert a.shape == b.shape
    max_delta = 35215 * threshold * threshold
    ys, xs = np.nonzero((a != b).any(-1))
    n = 0
    for y, x in zip(ys.tolist(), xs.tolist()):
        if _pm_delta(a[y, x], b[y, x]) > max_delta:
            if not (_pm_antialiased(a, x, y, b) or _pm_antialiased(b, x, y, a)):
                n += 1
    return n


# ---- the Rust decoder's golden format (rs/src/decoder/shape_decoder.rs: `format!("{:#?}\n", shape)` of Shape { paths:
#      Vec<StyledPath { path: lyon Path, fill: Option<FillStyle>, line: Option<LineStyle> }> }, rs/src/lib.rs:26-71)
def shape_to_rs_log(decoded: dict, ast: dict) -> str:
    """Pretty Debug text of the Rust `Shape` for a decoded shape (the product's swfr_shape_json output, straight lines and
    solid styles only, as in the three fixtures the Rust test holds); line-style attributes the TS form drops come from the AST."""
    def ind(n):
        return "    " * n

    def solid(color, n):
        c = [int(round(color[k] * 255)) for k in "rgba"]
        return [ind(n) + "Solid(", ind(n + 1) + "Solid {", ind(n + 2) + "color: StraightSRgba8 {",
                ind(n + 3) + "r: %d," % c[0], ind(n + 3) + "g: %d," % c[1], ind(n + 3) + "b: %d," % c[2], ind(n + 3) + "a: %d," % c[3],
                ind(n + 2) + "},", ind(n + 1) + "},", ind(n) + "),"]

    out = ["Shape {", ind(1) + "paths: ["]
    for p in decoded["paths"]:
        pts, verbs = [], []
        for c in p["commands"]:
            if c["type"] == 2:
                pts.append((c["x"], c["y"])); verbs.append("MoveTo")
            elif c["type"] == 0:
                pts.append((c["endX"], c["endY"])); verbs.append("LineTo")
            else:
                raise ValueError("the Rust decoder of the reference handles straight edges only")
        out += [ind(2) + "StyledPath {", ind(3) + "path: Path {", ind(4) + "points: ["]
        out += [ind(5) + "(%.1f,%.1f)," % (x, y) for x, y in pts]
        out += [ind(4) + "],", ind(4) + "verbs: ["]
        out += [ind(5) + v + "," for v in verbs]
        out += [ind(4) + "],", ind(3) + "},"]
        if "fill" in p:
            out += [ind(3) + "fill: Some("] + solid(p["fill"]["color"], 4) + [ind(3) + "),"]
        else:
            out.append(ind(3) + "fill: None,")
        if "line" in p:
            ls = next(l for l in ast["shape"]["initial_styles"]["line"] if l["width"] == p["line"]["width"])
            cap = lambda v: v.capitalize()
            out += [ind(3) + "line: Some(", ind(4) + "LineStyle {", ind(5) + "width: %d," % ls["width"],
                    ind(5) + "start_cap: %s," % cap(ls["start_cap"]), ind(5) + "end_cap: %s," % cap(ls["end_cap"]),
                    ind(5) + "join: %s," % cap(ls["join"]["type"])]
            out += [ind(5) + "%s: %s," % (k, "true" if ls[k] else "false") for k in ("no_h_scale", "no_v_scale", "no_close", "pixel_hinting")]
            out += [ind(5) + "fill: " + solid(p["line"]["fill"]["color"], 5)[0].strip()] + solid(p["line"]["fill"]["color"], 5)[1:]
            out += [ind(4) + "},", ind(3) + "),"]
        else:
            out.append(ind(3) + "line: None,")
        out.append(ind(2) + "},")
    out += [ind(1) + "],", "}"]
    return "\n".join(out) + "\n"


def make_bitmap_tag(bitmap_id, width, height, rng, colors=64):
    """A random DefineBitmap in the only format the reference decodes (image/x-swf-bmp, format 3: zlib-compressed colour table +
    8-bit indices, rows padded to 4; decode-x-swf-bmp.ts:9-41)."""
    import zlib
    table = rng.integers(0, 256, (colors, 3)).astype(np.uint8)
    padded = width + ((4 - (width % 4)) % 4)
    idx = np.zeros((height, padded), np.uint8)
    idx[:, :width] = rng.integers(0, colors, (height, width))
    body = zlib.compress(table.tobytes() + idx.tobytes())
    data = bytes([3, width & 255, width >> 8, height & 255, height >> 8, colors - 1]) + body
    return {"type": "define-bitmap", "id": bitmap_id, "width": width, "height": height, "media_type": "image/x-swf-bmp", "data": data.hex()}


def large_texture_scene(width=3840, height=2160, tex=4096, seed=99):
    """BASELINE.json config 4's HBM-bound variant: a frame-filling rectangle with a tex x tex bitmap fill sampled
    at width / tex pixels per texel (bilinear: every texel of the visible part is fetched about once, nothing stays in a cache)."""
    import scenarios
    rng = np.random.default_rng(seed)
    bmp = make_bitmap_tag(9, tex, tex, rng, colors=256)
    k = width / tex
    fill = {"type": "bitmap", "bitmap_id": 9, "repeating": False, "smoothed": True, "matrix": scenarios._m(20 * k, 20 * k, 0, 0)}
    pts = np.array([(0, 0), (width, 0), (width, height), (0, height)], float)
    return dict(width=width, height=height, bitmaps=[bmp], stage={"children": [{"type": "shape", "definition": scenarios._poly_shape(np.rint(pts * 20), fill)}]})


def rand_bitmap_scene(rng):
    """One random frame of bitmap-filled shapes as the renderer draws them: repeat / no-repeat fills from 20x minified to 30x
    magnified, rotated, reflected, partly off-frame, polygons and (un)aligned rectangles, with translucent solids in between."""
    import scenarios
    W, H = int(rng.integers(40, 200)), int(rng.integers(40, 140))
    bmp = make_bitmap_tag(3, int(rng.integers(1, 48)), int(rng.integers(1, 48)), rng)
    kids = []
    for _ in range(int(rng.integers(1, 4))):
        if rng.integers(0, 4) == 0:
            pts = rng.uniform(0, 1, (4, 2)) * [W, H]
            kids.append({"type": "shape", "definition": scenarios._poly_shape(np.rint(pts * 20), {"type": "solid", "color": scenarios._rgba(30, 200, 90, int(rng.choice([255, 140])))})})
            continue
        lo, hi = [(0.05, 0.74), (0.3, 3.0), (0.76, 8.0), (10, 30)][int(rng.integers(0, 4))]
        k = float(rng.uniform(lo, hi))
        t = float(rng.uniform(-3.2, 3.2)) if rng.integers(0, 3) else 0.0
        c, sn = np.cos(t), np.sin(t)
        flip = -1.0 if rng.integers(0, 5) == 0 else 1.0
        fill = {"type": "bitmap", "bitmap_id": 3, "repeating": bool(rng.integers(0, 2)), "smoothed": True,
                "matrix": scenarios._m(20 * k * c * flip, 20 * k * c, int(rng.integers(-200, W * 12)), int(rng.integers(-200, H * 12)), 20 * k * sn * flip, -20 * k * sn)}
        kind = int(rng.integers(0, 3))
        if kind == 0:
            pts = rng.uniform(-0.2, 1.2, (int(rng.integers(3, 7)), 2)) * [W, H]
        elif kind == 1:
            x0, y0 = rng.uniform(0, W / 2), rng.uniform(0, H / 2)
            x1, y1 = x0 + rng.uniform(3, W), y0 + rng.uniform(3, H)
            pts = np.array([(x0, y0), (x1, y0), (x1, y1), (x0, y1)])
        else:
            x0, y0 = int(rng.integers(0, W // 2)), int(rng.integers(0, H // 2))
            x1, y1 = x0 + int(rng.integers(1, W)), y0 + int(rng.integers(1, H))
            pts = np.array([(x0, y0), (x1, y0), (x1, y1), (x0, y1)], float)
        mat = scenarios._m(float(rng.choice([1, 1, 0.7, 1.6])), float(rng.choice([1, 1, 1.3])), int(rng.integers(-200, 300)), int(rng.integers(-200, 300)))
        kids.append({"type": "shape", "definition": scenarios._poly_shape(np.rint(pts * 20), fill), "matrix": mat})
    return dict(width=W, height=H, bitmaps=[bmp], stage={"children": kids})


def rand_radial_scene(rng):
    """One random frame of shapes with radial / focal gradient fills as the reference draws them (fill matrix maps the +-16384
    gradient box onto the shape), 1-8 stops with translucent colours and duplicate ratios, on a clear frame and over each other.
    The gradient circle is at least 0.6 frame diagonals wide and centred in the frame: samples stay inside pixman's 16.16 range."""
    import scenarios
    W, H = int(rng.integers(40, 200)), int(rng.integers(40, 140))
    kids = []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(1, 9))
        ratios = sorted(int(v) for v in rng.integers(0, 256, n))
        if n > 2 and rng.integers(0, 3) == 0:
            ratios[1] = ratios[0]
        colors = [(ratios[i], (int(rng.integers(0, 256)), int(rng.integers(0, 256)), int(rng.integers(0, 256)), int(rng.choice([255, 255, 128, 0, 37])))) for i in range(n)]
        ky = float(rng.uniform(0.5, 1.0))
        sc = float(rng.uniform(0.6, 3.0)) * 20 / 16384 * float(np.hypot(W, H)) / ky
        t = float(rng.uniform(-3.2, 3.2))
        c, sn = np.cos(t), np.sin(t)
        fill = {"type": "focal-gradient" if rng.integers(0, 2) else "radial-gradient", "gradient": scenarios._grad(colors),
                "matrix": scenarios._m(sc * c, sc * ky * c, int(rng.integers(0, W * 20)), int(rng.integers(0, H * 20)), sc * sn, -sc * ky * sn)}
        if fill["type"] == "focal-gradient":
            fill["focal_point"] = {"epsilons": int(rng.integers(-240, 241))}   # Sfixed8P8: -0.94 .. 0.94
        pts = rng.uniform(-0.1, 1.1, (int(rng.integers(3, 7)), 2)) * [W, H]
        kids.append({"type": "shape", "definition": scenarios._poly_shape(np.rint(pts * 20), fill)})
    return dict(width=W, height=H, stage={"children": kids})


def rand_mixed_scene(rng):
    """solid polygons (both fill rules, translucent), strokes of every style, morph shapes"""
    import scenarios
    from test_host import _rand_path_shape
    W, H = int(rng.integers(40, 260)), int(rng.integers(40, 180))
    kids = []
    for _ in range(int(rng.integers(1, 6))):
        k = int(rng.integers(0, 3))
        if k == 0:
            n = int(rng.integers(3, 10))
            mode = int(rng.integers(0, 4))
            if mode == 0: pts = rng.uniform(0, 1, (n, 2)) * [W, H]
            elif mode == 1: pts = rng.integers(0, 4 * min(W, H), (n, 2)) / 4.0
            elif mode == 2: pts = rng.integers(0, min(W, H), (n, 2)).astype(float)
            else: pts = rng.uniform(-40, 40 + max(W, H), (n, 2))
            col = scenarios._rgba(*[int(v) for v in rng.integers(0, 256, 3)], int(rng.choice([255, 255, 200, 128, 31, 1])))
            kids.append({"type": "shape", "definition": scenarios._poly_shape(np.rint(pts * 20), {"type": "solid", "color": col})})
        else:
            morph = k == 2
            tag = _rand_path_shape(rng, int(rng.choice([1, 2, 5, 20, 45, 90, 200])), morph)
            sx, sy = float(rng.choice([1, 1, 0.6, 1.7, -1])), float(rng.choice([1, 1, 0.8, 1.3]))
            mat = scenarios._m(sx, sy, int(rng.integers(-300, 900)) + (2000 if sx < 0 else 0), int(rng.integers(-300, 500)),
                               float(rng.choice([0, 0, 0.2])), float(rng.choice([0, 0, -0.15])))
            kids.append({"type": "morph-shape", "definition": tag, "ratio": float(rng.uniform(0, 1)), "matrix": mat} if morph else
                        {"type": "shape", "definition": tag, "matrix": mat})
    return dict(width=W, height=H, even_odd=bool(rng.integers(0, 2)), stage={"children": kids})




def rand_big_scene(rng):
    """A large frame (several hundred tiles) with dozens of overlapping shapes of every kind: the 64-row chunks of the fused row
    kernel, long band lists, occlusion culling and the shaded tile kernel all at once."""
    import scenarios
    W, H = int(rng.integers(500, 1300)), int(rng.integers(400, 900))
    bmp = make_bitmap_tag(3, int(rng.integers(8, 64)), int(rng.integers(8, 64)), rng)
    kids = []
    for _ in range(int(rng.integers(15, 50))):
        kind = int(rng.integers(0, 10))
        if kind < 5:
            sub = rand_mixed_scene(rng)
        elif kind < 8:
            sub = rand_bitmap_scene(rng)
        else:
            sub = rand_radial_scene(rng)
        k = sub["stage"]["children"][int(rng.integers(0, len(sub["stage"]["children"])))]
        k = dict(k)
        sc = float(rng.choice([1.0, 2.5, 4.0, 7.0]))
        m0 = k.get("matrix") or scenarios._m()
        tx, ty = int(rng.integers(-200, W * 20 - 200)), int(rng.integers(-200, H * 20 - 200))
        # outer placement: scale the child's own matrix and move it somewhere in the big frame
        k["matrix"] = {"scale_x": int(m0["scale_x"] * sc), "scale_y": int(m0["scale_y"] * sc), "rotate_skew0": int(m0["rotate_skew0"] * sc),
                       "rotate_skew1": int(m0["rotate_skew1"] * sc), "translate_x": int(m0["translate_x"] * sc) + tx, "translate_y": int(m0["translate_y"] * sc) + ty}
        kids.append(k)
    return dict(width=W, height=H, bitmaps=[bmp], stage={"children": kids})


def rand_long_scene(rng):
    """Strokes of 40 to 150 segments (round joins and caps when the shape is a morph shape): outlines of many hundred to a few
    thousand edges in ONE path -- the row kernels' larger routines, and the tie-order reconstruction over long edge lists."""
    import scenarios
    from test_host import _rand_path_shape
    W, H = int(rng.integers(250, 700)), int(rng.integers(200, 500))
    kids = []
    for _ in range(int(rng.integers(1, 4))):
        morph = bool(rng.integers(0, 3))
        tag = _rand_path_shape(rng, int(rng.choice([20, 45, 90, 200])), morph, segments=(40, 150))
        sx = float(rng.choice([1, 1.7, 2.5]))
        mat = scenarios._m(sx, sx * float(rng.choice([1, 0.8, 1.3])), int(rng.integers(-300, W * 10)), int(rng.integers(-300, H * 10)),
                           float(rng.choice([0, 0, 0.2])), float(rng.choice([0, 0, -0.15])))
        kids.append({"type": "morph-shape", "definition": tag, "ratio": float(rng.uniform(0, 1)), "matrix": mat} if morph else
                    {"type": "shape", "definition": tag, "matrix": mat})
    return dict(width=W, height=H, even_odd=bool(rng.integers(0, 2)), stage={"children": kids})


def soak_scene(name, seed, index):
    """Scene `index` of generator `name` in tools/soak.py's numbering (the generators are seeded per name)."""
    gens = {"mixed": rand_mixed_scene, "bitmap": rand_bitmap_scene, "radial": rand_radial_scene, "big": rand_big_scene, "long": rand_long_scene}
    rng = np.random.default_rng(seed + sum(map(ord, name)))
    for _ in range(index + 1):
        sc = gens[name](rng)
    return sc
