"""Scenario definitions shared by tools/make_goldens.py (libcairo, container only) and the parity tests.

Every scenario is a swf-tree Stage (JSON-like dicts) plus frame size, optional bitmaps and the fill
rule; goldens are premultiplied RGBA8 arrays rendered by the system libcairo through
oracle/cairo_backend.py, committed as tests/golden/cairo_<name>.npz.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "fixtures")


def load_fixture(name):
    with open(os.path.join(FIX, name + ".ast.json")) as f:
        return json.load(f)


def _m(sx=1.0, sy=1.0, tx=0, ty=0, r0=0.0, r1=0.0):
    return {"scale_x": int(round(sx * 65536)), "scale_y": int(round(sy * 65536)), "rotate_skew0": int(round(r0 * 65536)),
            "rotate_skew1": int(round(r1 * 65536)), "translate_x": int(tx), "translate_y": int(ty)}


def _poly_shape(points_twips, fill, line=None, line_width=0):
    """DefineShape with one closed polygon (straight edges), fill style 1 on the left, optional line style."""
    p = [(int(x), int(y)) for x, y in points_twips]
    sc = {"type": "style-change", "move_to": {"x": p[0][0], "y": p[0][1]}}
    if fill is not None:
        sc["left_fill"] = 1
    if line is not None:
        sc["line_style"] = 1
    recs = [sc]
    for k in range(1, len(p) + 1):
        a, b = p[k - 1], p[k % len(p)]
        recs.append({"type": "edge", "delta": {"x": b[0] - a[0], "y": b[1] - a[1]}})
    xs, ys = [q[0] for q in p], [q[1] for q in p]
    lines = [] if line is None else [{"width": line_width, "fill": {"type": "solid", "color": line}}]
    return {"id": 1, "bounds": {"x_min": min(xs), "x_max": max(xs), "y_min": min(ys), "y_max": max(ys)},
            "shape": {"initial_styles": {"fill": [] if fill is None else [fill], "line": lines}, "records": recs}}


def _rgba(r, g, b, a=255):
    return {"r": r, "g": g, "b": b, "a": a}


def _grad(stops):
    return {"spread": "pad", "color_space": "s-rgb", "colors": [{"ratio": t, "color": _rgba(*c)} for t, c in stops]}


def _circleish(cx, cy, r, n=24):
    return [(cx + r * math.cos(2 * math.pi * k / n), cy + r * math.sin(2 * math.pi * k / n)) for k in range(n)]


def scenarios():
    """name -> dict(width, height, stage, bitmaps, even_odd, exact)"""
    out = {}
    # --- reference fixtures at their own size
    for name in ("squares", "triangle", "homestuck-beta-1"):
        tag = load_fixture(name)
        b = tag["bounds"]
        w, h = math.ceil((b["x_max"] - b["x_min"]) / 20), math.ceil((b["y_max"] - b["y_min"]) / 20)
        out["fixture_" + name] = dict(width=w, height=h, exact=True, stage={"children": [
            {"type": "shape", "definition": tag, "matrix": _m(tx=-b["x_min"], ty=-b["y_min"])}]})
    # --- BASELINE config 2: flat shapes scaled to 1024x1024 through the placement matrix
    for name in ("squares", "triangle", "homestuck-beta-1"):
        tag = load_fixture(name)
        b = tag["bounds"]
        sx, sy = 1024 * 20 / (b["x_max"] - b["x_min"]), 1024 * 20 / (b["y_max"] - b["y_min"])
        out["config2_" + name] = dict(width=1024, height=1024, exact=True, stage={"children": [
            {"type": "shape", "definition": tag, "matrix": _m(sx, sy, -b["x_min"] * sx, -b["y_min"] * sy)}]})
    # --- morph fixture (BASELINE config 3 at native size, a few ratios incl. non-golden ones)
    tag = load_fixture("homestuck-beta-29")
    b, mb = tag["bounds"], tag["morph_bounds"]
    x0, x1 = min(b["x_min"], mb["x_min"]), max(b["x_max"], mb["x_max"])
    y0, y1 = min(b["y_min"], mb["y_min"]), max(b["y_max"], mb["y_max"])
    w, h = math.ceil((x1 - x0) / 20), math.ceil((y1 - y0) / 20)
    for k in (0, 37, 128, 200, 255):
        out["morph_%03d" % k] = dict(width=w, height=h, exact=True, stage={"children": [
            {"type": "morph-shape", "definition": tag, "ratio": k / 255, "matrix": _m(tx=-x0, ty=-y0)}]})
    sx, sy = 480 * 20 / (x1 - x0), 270 * 20 / (y1 - y0)
    out["config3_morph_480x270_r100"] = dict(width=480, height=270, exact=True, stage={"children": [
        {"type": "morph-shape", "definition": tag, "ratio": 100 / 255, "matrix": _m(sx, sy, -x0 * sx, -y0 * sy)}]})
    # --- gradients (radial, focal, alpha stops; linear = documented extension)
    circ = _circleish(1300, 1100, 1000)
    gm = _m(1000 / 16384, 1000 / 16384, 1300, 1100)
    rad = {"type": "radial-gradient", "matrix": gm, "gradient": _grad([(0, (255, 0, 0)), (128, (0, 255, 0)), (255, (0, 0, 255))])}
    out["gradient_radial"] = dict(width=130, height=115, exact=True, stage={"children": [{"type": "shape", "definition": _poly_shape(circ, rad)}]})
    foc = {"type": "focal-gradient", "matrix": gm, "focal_point": {"epsilons": 128},
           "gradient": _grad([(0, (255, 255, 255)), (255, (10, 20, 200))])}
    out["gradient_focal"] = dict(width=130, height=115, exact=True, stage={"children": [{"type": "shape", "definition": _poly_shape(circ, foc)}]})
    alp = {"type": "radial-gradient", "matrix": _m(1200 / 16384, 700 / 16384, 1300, 1100, 0.01, -0.02),
           "gradient": _grad([(0, (255, 200, 0, 255)), (100, (0, 100, 255, 60)), (255, (255, 0, 255, 200))])}
    back = _poly_shape([(100, 100), (2500, 300), (2300, 2200), (200, 1900)], {"type": "solid", "color": _rgba(40, 90, 20)})
    out["gradient_alpha_over"] = dict(width=130, height=115, exact=True, stage={"children": [
        {"type": "shape", "definition": back}, {"type": "shape", "definition": _poly_shape(circ, alp)}]})
    lin = {"type": "linear-gradient", "matrix": _m(1000 / 16384, 1000 / 16384, 1300, 1100, 0.02, 0.0),
           "gradient": _grad([(0, (0, 0, 0)), (80, (255, 128, 0)), (255, (255, 255, 255))])}
    out["gradient_linear_ext"] = dict(width=130, height=115, exact=False, stage={"children": [{"type": "shape", "definition": _poly_shape(circ, lin)}]})
    # --- BASELINE config 2, gradient half: the same radial / focal / linear(-extension) fills scaled to 1024x1024 through the
    #     placement matrix ("scale the placement matrix, not the twips")
    gsx, gsy = 1024 / 130, 1024 / 115
    for gname, gfill, gexact in (("radial", rad, True), ("focal", foc, True), ("linear_ext", lin, False)):
        out["config2_gradient_" + gname] = dict(width=1024, height=1024, exact=gexact, stage={"children": [
            {"type": "shape", "definition": _poly_shape(circ, gfill), "matrix": _m(gsx, gsy, 0, 0)}]})
    # --- colour quantisation pins: translucent solids at alpha 1, 37, 127, 128, 200, 254 over an opaque backdrop and over each other
    #     (the Cairo half of fromNormalizedColor: css-color.ts:11-13 -> node-canvas parse -> 8-bit premultiplied source), and an
    #     interpolated morph colour (fractional channels and alpha) at ratio 0.3
    kids = [{"type": "shape", "definition": _poly_shape([(100, 100), (1900, 150), (1850, 900), (150, 850)], {"type": "solid", "color": _rgba(250, 240, 20)})}]
    for i, a in enumerate((1, 37, 127, 128, 200, 254)):
        x = 60 + 300 * i
        kids.append({"type": "shape", "definition": _poly_shape([(x, 40 + 30 * i), (x + 520, 300), (x + 430, 1900 - 40 * i), (x - 30, 1500)],
                                                                {"type": "solid", "color": _rgba(30 + 35 * i, 200 - 30 * i, 90 + 25 * i, a)})})
    out["alpha_sweep"] = dict(width=110, height=100, exact=True, stage={"children": kids})
    ctag = json.loads(json.dumps(load_fixture("homestuck-beta-29")))
    for i, fl in enumerate(ctag["shape"]["initial_styles"]["fill"]):
        fl["color"], fl["morph_color"] = _rgba(200 - 60 * i, 40 + 50 * i, 40, 255), _rgba(40, 220 - 70 * i, 90 + 60 * i, 100 + 70 * i)
    out["morph_color_030"] = dict(width=w, height=h, exact=True, stage={"children": [
        {"type": "shape", "definition": _poly_shape([(0, 0), (w * 20, 0), (w * 20, h * 10), (0, h * 12)], {"type": "solid", "color": _rgba(10, 20, 30)})},
        {"type": "morph-shape", "definition": ctag, "ratio": 0.3, "matrix": _m(tx=-x0, ty=-y0)}]})
    # --- translucent paths over each other, nested containers, even-odd
    star = [(1000 + (900 if k % 2 == 0 else 350) * math.cos(2 * math.pi * k / 10 + 0.3),
             1000 + (900 if k % 2 == 0 else 350) * math.sin(2 * math.pi * k / 10 + 0.3)) for k in range(10)]
    penta = [star[(2 * k * 2) % 10] for k in range(5)]  # self-intersecting pentagram
    t1 = _poly_shape(star, {"type": "solid", "color": _rgba(255, 0, 0, 128)})
    t2 = _poly_shape(penta, {"type": "solid", "color": _rgba(0, 0, 255, 77)})
    t3 = _poly_shape([(300, 300), (1800, 500), (900, 1700)], {"type": "solid", "color": _rgba(20, 200, 30, 254)})
    out["translucent_stack"] = dict(width=100, height=100, exact=True, stage={"children": [
        {"type": "shape", "definition": t1},
        {"type": "container", "matrix": _m(0.9, 0.9, 60, 40), "children": [
            {"type": "shape", "definition": t2, "matrix": _m(1.0, 1.0, 30, -20)},
            {"type": "container", "matrix": _m(1.1, 0.8, 0, 200), "children": [{"type": "shape", "definition": t3}]}]}]})
    out["evenodd_pentagram"] = dict(width=100, height=100, exact=True, even_odd=True, stage={"children": [
        {"type": "shape", "definition": _poly_shape(penta, {"type": "solid", "color": _rgba(90, 60, 200)})}]})
    out["nonzero_pentagram"] = dict(width=100, height=100, exact=True, stage={"children": [
        {"type": "shape", "definition": _poly_shape(penta, {"type": "solid", "color": _rgba(90, 60, 200)})}]})
    # --- geometry leaving the frame (limit clipping) with a stroke
    big = [(-700, 300), (1500, -500), (2900, 900), (1200, 2600), (-300, 1800)]
    out["offframe_fill_stroke"] = dict(width=100, height=100, exact=True, stage={"children": [
        {"type": "shape", "definition": _poly_shape(big, {"type": "solid", "color": _rgba(200, 180, 40)}, line=_rgba(0, 0, 0), line_width=70)}]})
    # --- strokes beyond open miter/butt polylines: curves, the rectilinear box stroker, round caps + joins
    #     (morph shapes), hairlines that cairo drops, and round strokes leaving the frame
    def _path_shape(start, segs, fill=None, line=None, line_width=0):
        """segs: (dx, dy) straight or (dx, dy, cdx, cdy) quadratic with control delta; twips."""
        sc = {"type": "style-change", "move_to": {"x": int(start[0]), "y": int(start[1])}}
        if fill is not None:
            sc["left_fill"] = 1
        if line is not None:
            sc["line_style"] = 1
        recs = [sc]
        x, y = start
        xs, ys = [x], [y]
        for sg in segs:
            e = {"type": "edge", "delta": {"x": int(sg[0]), "y": int(sg[1])}}
            if len(sg) == 4:
                e["control_delta"] = {"x": int(sg[2]), "y": int(sg[3])}
                xs.append(x + sg[2]); ys.append(y + sg[3])
            x += sg[0]; y += sg[1]
            xs.append(x); ys.append(y)
            recs.append(e)
        lines = [] if line is None else [{"width": line_width, "fill": {"type": "solid", "color": line}}]
        return {"id": 1, "bounds": {"x_min": int(min(xs)), "x_max": int(max(xs)), "y_min": int(min(ys)), "y_max": int(max(ys))},
                "shape": {"initial_styles": {"fill": [] if fill is None else [fill], "line": lines}, "records": recs}}

    blob = _path_shape((300, 900), [(700, -600, 100, -500), (600, 500, 500, -100), (-300, 600, 200, 500), (-1000, -500, -600, 300)],
                       fill={"type": "solid", "color": _rgba(250, 220, 90)}, line=_rgba(30, 30, 120), line_width=90)
    out["stroke_curves"] = dict(width=100, height=90, exact=True, stage={"children": [{"type": "shape", "definition": blob}]})
    thin_curve = _path_shape((200, 300), [(900, 300, 800, -250), (400, 900, -500, 300), (-1100, -200, -300, 500)],
                             line=_rgba(0, 0, 0, 160), line_width=14)
    out["stroke_curves_thin_translucent"] = dict(width=100, height=90, exact=True, stage={"children": [{"type": "shape", "definition": thin_curve}]})
    stairs = _path_shape((205, 310), [(400, 0), (0, 300), (350, 0), (0, -450), (500, 0), (0, 900), (-1100, 0)], line=_rgba(200, 30, 30), line_width=50)
    out["stroke_rectilinear_open"] = dict(width=90, height=70, exact=True, stage={"children": [{"type": "shape", "definition": stairs}]})
    frame_loop = _path_shape((300, 300), [(1000, 0), (0, 700), (-1000, 0), (0, -700)], fill={"type": "solid", "color": _rgba(20, 160, 90, 200)},
                             line=_rgba(0, 0, 0, 128), line_width=65)
    out["stroke_rectilinear_loop_scaled"] = dict(width=120, height=80, exact=True, stage={"children": [
        {"type": "shape", "definition": frame_loop, "matrix": _m(1.3, 0.9, 137, 55)}]})
    hair = {"children": [
        {"type": "shape", "definition": _path_shape((100, 100), [(1500, 900), (-700, 400)], line=_rgba(0, 0, 0), line_width=1)},      # 0.05 px: dropped
        {"type": "shape", "definition": _path_shape((100, 1300), [(1500, -900), (-200, -300)], line=_rgba(0, 0, 0), line_width=2)},   # 0.1 px: drawn
        {"type": "shape", "definition": _path_shape((200, 200), [(0, 1200), (1300, 0)], line=_rgba(0, 0, 0), line_width=1)}]}          # rectilinear, dropped too
    out["stroke_hairlines"] = dict(width=90, height=80, exact=True, stage=hair)
    mtag = json.loads(json.dumps(load_fixture("homestuck-beta-29")))
    for ln in mtag["shape"]["initial_styles"]["line"]:
        ln["width"], ln["morph_width"] = 60, 160
        ln["fill"]["color"], ln["fill"]["morph_color"] = _rgba(200, 40, 40, 255), _rgba(40, 40, 220, 140)
    for k in (0, 90, 255):
        out["morph_round_stroke_%03d" % k] = dict(width=w, height=h, exact=True, stage={"children": [
            {"type": "morph-shape", "definition": mtag, "ratio": k / 255, "matrix": _m(tx=-x0, ty=-y0)}]})
    out["morph_round_stroke_offframe"] = dict(width=w, height=h, exact=True, stage={"children": [
        {"type": "morph-shape", "definition": mtag, "ratio": 0.6, "matrix": _m(1.4, 1.2, -x0 * 1.4 - 400, -y0 * 1.2 + 300)}]})
    # --- bitmap fill, magnified (BASELINE config 4 at reduced size; bilinear region of FILTER_GOOD)
    tag4 = load_fixture("homestuck-beta-4")
    b = tag4["bounds"]
    sc = 5.0
    bmp = load_fixture("homestuck-beta-3.bitmap")
    out["bitmap_magnified"] = dict(width=math.ceil((b["x_max"] - b["x_min"]) / 20 * sc), height=math.ceil((b["y_max"] - b["y_min"]) / 20 * sc),
                                   exact=True, bitmaps=[bmp], stage={"children": [
        {"type": "shape", "definition": tag4, "matrix": _m(sc, sc, -b["x_min"] * sc, -b["y_min"] * sc)}]})
    # --- the reference's textured fixture at its own size: the bitmap is minified 2.58x, i.e. CAIRO_FILTER_GOOD's separable
    #     convolution (pixman) rather than bilinear; and a rotated, strongly minified repeat fill
    out["fixture_homestuck-beta-4"] = dict(width=math.ceil((b["x_max"] - b["x_min"]) / 20), height=math.ceil((b["y_max"] - b["y_min"]) / 20),
                                           exact=True, bitmaps=[bmp], stage={"children": [
        {"type": "shape", "definition": tag4, "matrix": _m(tx=-b["x_min"], ty=-b["y_min"])}]})
    out["bitmap_minified_rotated"] = dict(width=120, height=90, exact=True, bitmaps=[bmp], stage={"children": [
        {"type": "shape", "definition": tag4, "matrix": _m(0.55, 0.4, 500 - (0.55 * b["x_min"] - 0.15 * b["y_min"]),
                                                                 150 - (0.2 * b["x_min"] + 0.4 * b["y_min"]), 0.2, -0.15)}]})
    # --- non-repeating bitmaps smaller than the polygon they fill: the operation is bounded by the source's own extents, which also
    #     become the polygon limits (magnified: half a source pixel of filter blur, rounded; minified: rounded out); on top of a
    #     translucent solid so that both blend paths (clear surface / OVER) are hit
    big = [(100, 160), (2900, 60), (2940, 2100), (1500, 2160), (60, 2040)]
    under = {"type": "shape", "definition": _poly_shape([(40, 900), (2960, 700), (2960, 1500), (40, 1300)], {"type": "solid", "color": _rgba(40, 200, 90, 120)})}
    for name, k, rot, tx, ty, rep, first in (("bitmap_no_repeat_magnified", 1.45, 0.35, 700, 300, False, True),
                                             ("bitmap_no_repeat_minified", 0.31, -0.5, 600, 500, False, False),
                                             ("bitmap_repeat_over_solid", 0.9, 2.2, 300, 200, True, False)):
        c, sn = math.cos(rot), math.sin(rot)
        fill = {"type": "bitmap", "bitmap_id": 3, "repeating": rep, "smoothed": True, "matrix": _m(20 * k * c, 20 * k * c, tx, ty, 20 * k * sn, -20 * k * sn)}
        shape = {"type": "shape", "definition": _poly_shape(big, fill)}
        out[name] = dict(width=150, height=110, exact=True, bitmaps=[bmp], stage={"children": [shape, under] if first else [under, shape]})
    return out
