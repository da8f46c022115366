"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's TS render path.

This module restates, in plain Python (small inputs only), what the reference does
*above* the Canvas2D boundary:

  * ``decode_swf_shape``        <- ts/src/lib/shape/decode-swf-shape.ts:22-39,203-448
  * ``decode_swf_morph_shape``  <- ts/src/lib/shape/decode-swf-morph-shape.ts:21-41,170-425
  * ``decode_x_swf_bmp``        <- ts/src/lib/decode-x-swf-bmp.ts:9-41
  * ``CanvasReplay``            <- ts/src/lib/renderers/canvas-renderer.ts:48-351
  * ``stage_for_shape/morph``   <- ts/src/test/node-canvas-renderer.spec.ts:31-52,86-113
  * ``shape_to_ts_json``        <- the JSON.stringify(…, null, 2) golden format of
                                   ts/src/test/decode-shape.spec.ts:18-24

``CanvasReplay`` emits the exact sequence of Canvas2D calls the reference makes into a
*backend* object.  Two backends exist: ``oracle_backend.OracleBackend`` (the C restatement
of the Cairo arithmetic, oracle/swfr_oracle.c) and ``cairo_backend.CairoBackend`` (the
system libcairo through ctypes, used only to pin the restatement and to generate golden
vectors; never on the product path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import json
import math
import zlib

# CommandType enum of ts/src/lib/shape/path.ts:4-8
CMD_LINE_TO, CMD_CURVE_TO, CMD_MOVE_TO = 0, 1, 2
# FillStyleType enum of ts/src/lib/shape/fill-style.ts:5-10
FILL_BITMAP, FILL_FOCAL, FILL_LINEAR, FILL_SOLID = 0, 1, 2, 3


# --------------------------------------------------------------------------------------
# decode-swf-shape.ts
# --------------------------------------------------------------------------------------
def _js_num(v):
    """Numbers as JS prints them: integral doubles print without a fraction."""
    if isinstance(v, float) and v.is_integer():
        return int(v)
    return v


def _norm_color(c):
    # normalizeStraightSRgba, decode-swf-shape.ts:90-97
    return {k: _js_num(c[k] / 255) for k in "rgba"}


def _decode_gradient(g):
    # decodeGradient, decode-swf-shape.ts:99-105 ({...swfGradient, colors})
    out = dict(g)
    out["colors"] = [{"ratio": _js_num(s["ratio"] / 0xFF), "color": _norm_color(s["color"])} for s in g["colors"]]
    return out


def _sfixed8p8(v):
    # swf-tree Sfixed8P8: {"epsilons": n} in JSON; valueOf() = n / 256
    if isinstance(v, dict):
        return _js_num(v["epsilons"] / 256.0)
    return _js_num(float(v))


def _decode_fill(s):
    # decodeFillStyle, decode-swf-shape.ts:110-139
    t = s["type"]
    if t == "bitmap":
        return {"type": FILL_BITMAP, "bitmapId": s["bitmap_id"], "matrix": s["matrix"],
                "repeating": s["repeating"], "smoothed": s["smoothed"]}
    if t == "focal-gradient":
        return {"type": FILL_FOCAL, "matrix": s["matrix"], "gradient": _decode_gradient(s["gradient"]),
                "focalPoint": _sfixed8p8(s["focal_point"])}
    if t == "linear-gradient":
        return {"type": FILL_LINEAR, "matrix": s["matrix"], "gradient": _decode_gradient(s["gradient"])}
    if t == "radial-gradient":
        return {"type": FILL_FOCAL, "matrix": s["matrix"], "gradient": _decode_gradient(s["gradient"]),
                "focalPoint": 0}
    if t == "solid":
        return {"type": FILL_SOLID, "color": _norm_color(s["color"])}
    raise ValueError("UnknownFillStyle")


def _decode_line(s):
    # decodeLineStyle, decode-swf-shape.ts:144-149 (caps/joins are dropped)
    return {"width": s["width"], "fill": _decode_fill(s["fill"])}


def _extract_continuous(open_set, key=lambda v: v):
    # extractContinuous, decode-swf-shape.ts:203-234 (single greedy pass, splice semantics)
    first = open_set.pop(0)
    result = [first]
    sx, sy, ex, ey = key(first["sx"]), key(first["sy"]), key(first["ex"]), key(first["ey"])
    i, n = 0, len(open_set)
    while i < n:
        cur = open_set[i]
        if key(cur["sx"]) == ex and key(cur["sy"]) == ey:
            open_set.pop(i)
            n -= 1
            ex, ey = key(cur["ex"]), key(cur["ey"])
            result.append(cur)
            continue
        if key(cur["ex"]) == sx and key(cur["ey"]) == sy:
            open_set.pop(i)
            n -= 1
            sx, sy = key(cur["sx"]), key(cur["sy"])
            result.insert(0, cur)
            continue
        i += 1
    return result


def _segments_to_commands(segments, key=lambda v: v):
    # segmentsToCommands, decode-swf-shape.ts:239-273
    open_set = list(segments)
    out = []
    while open_set:
        seq = _extract_continuous(open_set, key)
        out.append({"type": CMD_MOVE_TO, "x": seq[0]["sx"], "y": seq[0]["sy"]})
        for s in seq:
            if "cx" in s:
                out.append({"type": CMD_CURVE_TO, "controlX": s["cx"], "controlY": s["cy"],
                            "endX": s["ex"], "endY": s["ey"]})
            else:
                out.append({"type": CMD_LINE_TO, "endX": s["ex"], "endY": s["ey"]})
    return out


def _layer_to_paths(layer, key=lambda v: v):
    # layerToPaths, decode-swf-shape.ts:278-293: all fills in style order, then all lines
    paths = []
    for fs in layer["fills"]:
        cmds = _segments_to_commands(fs["segments"], key)
        if cmds:
            paths.append({"commands": cmds, "fill": fs["style"]})
    for ls in layer["lines"]:
        cmds = _segments_to_commands(ls["segments"], key)
        if cmds:
            paths.append({"commands": cmds, "line": ls["style"]})
    return paths


def decode_swf_shape(tag):
    """SwfShapeDecoder, decode-swf-shape.ts:298-448."""
    layers = []
    st = {"left": None, "right": None, "line": None, "x": 0, "y": 0}

    def new_styles(fills, lines):
        layers.append({"fills": [{"style": _decode_fill(f), "segments": []} for f in fills],
                       "lines": [{"style": _decode_line(l), "segments": []} for l in lines]})
        st["left"] = st["right"] = st["line"] = None

    def pick(kind, sid):
        if sid == 0:
            return None
        arr = layers[-1][kind]
        if sid - 1 >= len(arr):
            raise ValueError("Invalid fill ID")
        return arr[sid - 1]

    ini = tag["shape"]["initial_styles"]
    new_styles(ini["fill"], ini["line"])
    for rec in tag["shape"]["records"]:
        if rec["type"] == "style-change":
            # applyStyleChange :337-356 (order: newStyles, leftFill, rightFill, lineStyle, moveTo)
            if rec.get("new_styles") is not None:
                new_styles(rec["new_styles"]["fill"], rec["new_styles"]["line"])
            if rec.get("left_fill") is not None:
                st["left"] = pick("fills", rec["left_fill"])
            if rec.get("right_fill") is not None:
                st["right"] = pick("fills", rec["right_fill"])
            if rec.get("line_style") is not None:
                st["line"] = pick("lines", rec["line_style"])
            if rec.get("move_to") is not None:
                st["x"], st["y"] = rec["move_to"]["x"], rec["move_to"]["y"]
        elif rec["type"] == "edge":
            # applyEdge :358-390 (left forward, right reversed, line forward)
            x, y = st["x"], st["y"]
            ex, ey = x + rec["delta"]["x"], y + rec["delta"]["y"]
            cd = rec.get("control_delta")
            if cd is None:
                fwd = {"sx": x, "sy": y, "ex": ex, "ey": ey}
                rev = {"sx": ex, "sy": ey, "ex": x, "ey": y}
            else:
                cx, cy = x + cd["x"], y + cd["y"]
                fwd = {"sx": x, "sy": y, "cx": cx, "cy": cy, "ex": ex, "ey": ey}
                rev = {"sx": ex, "sy": ey, "cx": cx, "cy": cy, "ex": x, "ey": y}
            if st["left"] is not None:
                st["left"]["segments"].append(dict(fwd))
            if st["right"] is not None:
                st["right"]["segments"].append(dict(rev))
            if st["line"] is not None:
                st["line"]["segments"].append(dict(fwd))
            st["x"], st["y"] = ex, ey
        else:
            raise ValueError("UnreachableCode")
    paths = []
    for layer in layers:
        paths.extend(_layer_to_paths(layer))
    return {"paths": paths}


# --------------------------------------------------------------------------------------
# decode-swf-morph-shape.ts
# --------------------------------------------------------------------------------------
def decode_swf_morph_shape(tag):
    """SwfMorphShapeDecoder, decode-swf-morph-shape.ts:265-425 (solid fills only :94-106)."""
    def dfill(s):
        if s["type"] != "solid":
            raise ValueError("Unknown fill type")
        return {"type": 0, "startColor": _norm_color(s["color"]), "endColor": _norm_color(s["morph_color"])}

    def dline(s):
        return {"width": [s["width"], s["morph_width"]], "fill": dfill(s["fill"])}

    layers = []
    st = {"left": None, "right": None, "line": None, "x": [0, 0], "y": [0, 0]}

    def new_styles(fills, lines):
        layers.append({"fills": [{"style": dfill(f), "segments": []} for f in fills],
                       "lines": [{"style": dline(l), "segments": []} for l in lines]})
        st["left"] = st["right"] = st["line"] = None

    def pick(kind, sid):
        if sid == 0:
            return None
        arr = layers[-1][kind]
        if sid - 1 >= len(arr):
            raise ValueError("Invalid fill ID")
        return arr[sid - 1]

    ini = tag["shape"]["initial_styles"]
    new_styles(ini["fill"], ini["line"])
    for rec in tag["shape"]["records"]:
        if rec["type"] == "style-change":
            # applyStyleChange :304-322 (no newStyles for morph shapes)
            if rec.get("left_fill") is not None:
                st["left"] = pick("fills", rec["left_fill"])
            if rec.get("right_fill") is not None:
                st["right"] = pick("fills", rec["right_fill"])
            if rec.get("line_style") is not None:
                st["line"] = pick("lines", rec["line_style"])
            if rec.get("move_to") is not None:
                if rec.get("morph_move_to") is None:
                    raise ValueError("Expected morphMoveTo to be defined")
                st["x"] = [rec["move_to"]["x"], rec["morph_move_to"]["x"]]
                st["y"] = [rec["move_to"]["y"], rec["morph_move_to"]["y"]]
        elif rec["type"] == "edge":
            # applyEdge :324-364; a missing control delta is delta/2 (:341-346)
            x, y = st["x"], st["y"]
            d, md = rec["delta"], rec["morph_delta"]
            ex = [x[0] + d["x"], x[1] + md["x"]]
            ey = [y[0] + d["y"], y[1] + md["y"]]
            cd, mcd = rec.get("control_delta"), rec.get("morph_control_delta")
            if cd is None and mcd is None:
                fwd = {"sx": x, "sy": y, "ex": ex, "ey": ey}
                rev = {"sx": ex, "sy": ey, "ex": x, "ey": y}
            else:
                if cd is None:
                    cd = {"x": _js_num(d["x"] / 2), "y": _js_num(d["y"] / 2)}
                if mcd is None:
                    mcd = {"x": _js_num(md["x"] / 2), "y": _js_num(md["y"] / 2)}
                cx = [_js_num(x[0] + cd["x"]), _js_num(x[1] + mcd["x"])]
                cy = [_js_num(y[0] + cd["y"]), _js_num(y[1] + mcd["y"])]
                fwd = {"sx": x, "sy": y, "cx": cx, "cy": cy, "ex": ex, "ey": ey}
                rev = {"sx": ex, "sy": ey, "cx": cx, "cy": cy, "ex": x, "ey": y}
            if st["left"] is not None:
                st["left"]["segments"].append(dict(fwd))
            if st["right"] is not None:
                st["right"]["segments"].append(dict(rev))
            if st["line"] is not None:
                st["line"]["segments"].append(dict(fwd))
            st["x"], st["y"] = ex, ey
        else:
            raise ValueError("UnreachableCode")
    paths = []
    for layer in layers:
        # extractContinuous chains on the START-state coordinates only (:170-201)
        paths.extend(_layer_to_paths(layer, key=lambda v: v[0]))
    return {"paths": paths}


# --------------------------------------------------------------------------------------
# golden JSON format (JSON.stringify(shape, null, 2) + "\n")
# --------------------------------------------------------------------------------------
def _matrix_ts(m):
    # swf-tree Matrix: Sfixed16P16 members serialise as {"epsilons": n}
    return {"scaleX": {"epsilons": m["scale_x"]}, "scaleY": {"epsilons": m["scale_y"]},
            "rotateSkew0": {"epsilons": m["rotate_skew0"]}, "rotateSkew1": {"epsilons": m["rotate_skew1"]},
            "translateX": m["translate_x"], "translateY": m["translate_y"]}


def _fill_ts(f):
    f = dict(f)
    if "matrix" in f:
        f["matrix"] = _matrix_ts(f["matrix"])
    return f


def shape_to_ts_json(shape):
    """Serialise a decoded (morph) shape the way the reference's decode specs do."""
    paths = []
    for p in shape["paths"]:
        q = {"commands": p["commands"]}
        if "fill" in p:
            q["fill"] = _fill_ts(p["fill"])
        if "line" in p:
            line = dict(p["line"])
            line["fill"] = _fill_ts(line["fill"])
            q["line"] = line
        paths.append(q)
    return json.dumps({"paths": paths}, indent=2) + "\n"


# --------------------------------------------------------------------------------------
# decode-x-swf-bmp.ts
# --------------------------------------------------------------------------------------
def decode_x_swf_bmp(data: bytes):
    """decodeXSwfBmpSync, decode-x-swf-bmp.ts:9-41 -> (width, height, straight RGBA bytes)."""
    if data[0] != 3:
        raise ValueError("UnsupportedXSwfBmpFormatId: %d" % data[0])
    width = data[1] | (data[2] << 8)
    height = data[3] | (data[4] << 8)
    padded = width + ((4 - (width % 4)) % 4)
    color_count = data[5] + 1
    src = zlib.decompress(bytes(data[6:]))
    table = 3 * color_count
    out = bytearray(width * height * 4)
    for y in range(height):
        row = table + y * padded
        for x in range(width):
            ci = src[row + x]
            o = 4 * (y * width + x)
            if ci < color_count:
                out[o:o + 3] = src[3 * ci:3 * ci + 3]
            out[o + 3] = 0xFF
    return width, height, bytes(out)


def image_to_pam(width, height, rgba: bytes) -> bytes:
    """imageDataToPam, ts/src/lib/image-data-to-pam.ts:8-28."""
    header = "\n".join(["P7", "WIDTH %d" % width, "HEIGHT %d" % height, "DEPTH 4", "MAXVAL 255",
                        "TUPLTYPE RGB_ALPHA", "ENDHDR", ""])
    return header.encode("ascii") + rgba


# --------------------------------------------------------------------------------------
# canvas-renderer.ts
# --------------------------------------------------------------------------------------
def _lerp(a, b, r):
    # lerp, canvas-renderer.ts:24-26
    return b * r + a * (1 - r)


def css_rgba(color):
    """fromNormalizedColor (css-color.ts:11-13) followed by node-canvas' CSS colour parse.

    Returns (r8, g8, b8, a) as handed to cairo_set_source_rgba(r8/255, g8/255, b8/255, a8/255).
    R is `& 0xff`-truncated by the reference; G/B are printed as doubles and parsed to ints
    (exact for n/255*255; non-integers truncate), alpha goes through an
    8-bit quantisation `(int)(a * 255)` in float (parity unpinned for 0 < a < 1: no fixture).
    """
    import struct
    r8 = int(color["r"] * 0xFF) & 0xFF
    g8 = max(0, min(255, int(color["g"] * 255)))
    b8 = max(0, min(255, int(color["b"] * 255)))
    a = color["a"]
    af = struct.unpack("f", struct.pack("f", max(0.0, min(1.0, a))))[0]
    a8 = int(struct.unpack("f", struct.pack("f", af * 255.0))[0])
    return r8, g8, b8, a8


def _sfixed(v):
    return v / 65536.0


class CanvasReplay:
    """CanvasRenderer (canvas-renderer.ts:48-351) emitting Canvas2D calls into `backend`."""

    GRAD_RADIUS = 16384  # canvas-renderer.ts:322

    def __init__(self, backend, linear_extension=False):
        self.be = backend
        # Linear gradients throw NotImplementedFillStyle in the reference (canvas-renderer.ts:332-333).
        # linear_extension=True enables the documented beyond-reference behaviour:
        # createLinearGradient(-16384, 0, 16384, 0) under fill.matrix.
        self.linear_extension = linear_extension
        self.bitmaps = {}
        self._shape_cache = {}
        self._morph_cache = {}

    def add_bitmap(self, tag):
        # NodeCanvasBitmapService.addBitmap, node-canvas-bitmap-service.ts:14-37
        if tag["media_type"] != "image/x-swf-bmp":
            raise NotImplementedError("Support for %s images" % tag["media_type"])
        data = tag["data"]
        if isinstance(data, str):
            data = bytes.fromhex(data)
        w, h, rgba = decode_x_swf_bmp(data)
        self.bitmaps[tag["id"]] = self.be.create_bitmap(w, h, rgba)

    def render(self, stage):
        # renderStage :69-78
        be = self.be
        be.set_transform_identity()
        be.clear_all()
        be.scale(1 / 20, 1 / 20)
        for child in stage["children"]:
            self._draw(child)

    def _draw(self, obj):
        t = obj["type"]
        if t == "container":
            self.be.save()
            try:
                if obj.get("matrix") is not None:
                    self._apply_matrix(obj["matrix"])
                for c in obj["children"]:
                    self._draw(c)
            finally:
                self.be.restore()
        elif t == "shape":
            self.be.save()
            try:
                if obj.get("matrix") is not None:
                    self._apply_matrix(obj["matrix"])
                key = id(obj["definition"])
                if key not in self._shape_cache:
                    self._shape_cache[key] = decode_swf_shape(obj["definition"])
                for p in self._shape_cache[key]["paths"]:
                    self._draw_path(p)
            finally:
                self.be.restore()
        elif t == "morph-shape":
            self.be.save()
            try:
                if obj.get("matrix") is not None:
                    self._apply_matrix(obj["matrix"])
                key = id(obj["definition"])
                if key not in self._morph_cache:
                    self._morph_cache[key] = decode_swf_morph_shape(obj["definition"])
                for p in self._morph_cache[key]["paths"]:
                    self._draw_morph_path(p, obj["ratio"])
            finally:
                self.be.restore()
        else:
            raise ValueError("UnexpectedDisplayObjectType")

    def _apply_matrix(self, m):
        # applyMatrix :179-188
        self.be.transform(_sfixed(m["scale_x"]), _sfixed(m["rotate_skew0"]), _sfixed(m["rotate_skew1"]),
                          _sfixed(m["scale_y"]), m["translate_x"], m["translate_y"])

    def _draw_path(self, path):
        # drawPath :269-350
        be = self.be
        if ("fill" not in path and "line" not in path) or not path["commands"]:
            return
        be.begin_path()
        for c in path["commands"]:
            if c["type"] == CMD_CURVE_TO:
                be.quadratic_curve_to(c["controlX"], c["controlY"], c["endX"], c["endY"])
            elif c["type"] == CMD_LINE_TO:
                be.line_to(c["endX"], c["endY"])
            else:
                be.move_to(c["x"], c["y"])
        if "fill" in path:
            f = path["fill"]
            be.save()
            if f["type"] == FILL_BITMAP:
                bmp = self.bitmaps.get(f["bitmapId"])
                if bmp is None:
                    # the reference throws BitmapNotFound for an unknown id; a known id
                    # without pixels falls back to this colour (:298-304)
                    raise KeyError("BitmapNotFound: %d" % f["bitmapId"])
                self._apply_matrix(f["matrix"])
                be.set_fill_pattern(bmp, bool(f["repeating"]))
            elif f["type"] == FILL_SOLID:
                be.set_fill_rgba(*css_rgba(f["color"]))
            elif f["type"] == FILL_FOCAL:
                self._apply_matrix(f["matrix"])
                stops = [(s["ratio"],) + css_rgba(s["color"]) for s in f["gradient"]["colors"]]
                be.set_fill_radial(_lerp(0, self.GRAD_RADIUS, f["focalPoint"]), 0, 0, 0, 0, self.GRAD_RADIUS, stops)
            elif f["type"] == FILL_LINEAR and self.linear_extension:
                self._apply_matrix(f["matrix"])
                stops = [(s["ratio"],) + css_rgba(s["color"]) for s in f["gradient"]["colors"]]
                be.set_fill_linear(-self.GRAD_RADIUS, 0, self.GRAD_RADIUS, 0, stops)
            else:
                be.restore()
                raise NotImplementedError("NotImplementedFillStyle")
            be.fill()
            be.restore()
        if "line" in path:
            ln = path["line"]
            if ln["fill"]["type"] != FILL_SOLID:
                raise NotImplementedError("NotImplementedLineStyle")
            be.set_line_width(ln["width"])
            be.set_stroke_rgba(*css_rgba(ln["fill"]["color"]))
            be.stroke()

    def _draw_morph_path(self, path, ratio):
        # drawMorphPath :207-267
        be = self.be
        if ("fill" not in path and "line" not in path) or not path["commands"]:
            return
        be.begin_path()
        L = lambda pair: _lerp(pair[0], pair[1], ratio)
        for c in path["commands"]:
            if c["type"] == CMD_CURVE_TO:
                be.quadratic_curve_to(L(c["controlX"]), L(c["controlY"]), L(c["endX"]), L(c["endY"]))
            elif c["type"] == CMD_LINE_TO:
                be.line_to(L(c["endX"]), L(c["endY"]))
            else:
                be.move_to(L(c["x"]), L(c["y"]))

        def lerp_rgba(a, b):
            return {k: _lerp(a[k], b[k], ratio) for k in "rgba"}

        if "fill" in path:
            f = path["fill"]
            be.set_fill_rgba(*css_rgba(lerp_rgba(f["startColor"], f["endColor"])))
            be.fill()
        if "line" in path:
            ln = path["line"]
            be.set_line_width(_lerp(ln["width"][0], ln["width"][1], ratio))
            be.set_stroke_rgba(*css_rgba(lerp_rgba(ln["fill"]["startColor"], ln["fill"]["endColor"])))
            be.set_line_cap_round()
            be.set_line_join_round()
            be.stroke()


# --------------------------------------------------------------------------------------
# node-canvas-renderer.spec.ts harness (frame size + placement matrix)
# --------------------------------------------------------------------------------------
def _placement(x_min, y_min, scale=1.0):
    s = int(round(scale * 65536))
    return {"scale_x": s, "scale_y": s, "rotate_skew0": 0, "rotate_skew1": 0,
            "translate_x": -x_min, "translate_y": -y_min}


def stage_for_shape(tag):
    """node-canvas-renderer.spec.ts:31-52 -> (width, height, stage)."""
    b = tag["bounds"]
    w = math.ceil((b["x_max"] - b["x_min"]) / 20)
    h = math.ceil((b["y_max"] - b["y_min"]) / 20)
    stage = {"width": w, "height": h, "children": [
        {"type": "shape", "definition": tag, "matrix": _placement(b["x_min"], b["y_min"])}]}
    return w, h, stage


def stage_for_morph_shape(tag, ratio):
    """node-canvas-renderer.spec.ts:86-113 -> (width, height, stage)."""
    b, mb = tag["bounds"], tag["morph_bounds"]
    x_min, x_max = min(b["x_min"], mb["x_min"]), max(b["x_max"], mb["x_max"])
    y_min, y_max = min(b["y_min"], mb["y_min"]), max(b["y_max"], mb["y_max"])
    w = math.ceil((x_max - x_min) / 20)
    h = math.ceil((y_max - y_min) / 20)
    stage = {"width": w, "height": h, "children": [
        {"type": "morph-shape", "definition": tag, "ratio": ratio, "matrix": _placement(x_min, y_min)}]}
    return w, h, stage


def unpremultiply(premul):
    """node-canvas getImageData / PNG encode: c' = (c*255 + a/2) / a, a == 0 -> 0 (numpy uint8 HxWx4)."""
    import numpy as np
    p = premul.astype(np.uint32)
    a = p[..., 3:4]
    safe = np.where(a == 0, 1, a)
    rgb = np.where(a == 0, 0, (p[..., :3] * 255 + a // 2) // safe)
    return np.concatenate([rgb, a], axis=-1).astype(np.uint8)
