"""ctypes binding of include/swfr.h and the host-side mirror of the reference's renderer interface.

`Renderer` mirrors ts/src/lib/renderer.ts:4-8 (`render(stage)`, `addBitmap(tag)`) and
ts/src/lib/renderers/node-canvas-renderer.ts:7-24 (`new NodeCanvasRenderer(width, height)`), with
`register_shape` / `register_morph_shape` from rs/src/asset.rs:9-12.  Stages and tags are the
swf-tree JSON objects of the reference fixtures (tests/<set>/<name>/ast.json).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, ERR_INVALID, ERR_NOT_IMPLEMENTED, ERR_NOT_FOUND, ERR_NO_DEVICE, ERR_DEVICE, ERR_CAPACITY = range(7)
DEVICE_HOST_ONLY = -1
FLAG_EVEN_ODD = 1
FLAG_BANDS_CONTIGUOUS = 2
PATH_TOR, PATH_BOXES = 0, 1
STYLE_SOLID, STYLE_RADIAL, STYLE_LINEAR, STYLE_BITMAP = 0, 1, 2, 3
MAX_STOPS = 16

EXPORTS = [
    "swfr_abi_version", "swfr_create", "swfr_destroy", "swfr_last_error", "swfr_register_shape",
    "swfr_register_morph_shape", "swfr_register_bitmap", "swfr_render", "swfr_render_batch", "swfr_read_image", "swfr_upload_edges",
    "swfr_render_resident", "swfr_render_edges", "swfr_build_frame", "swfr_shape_json", "swfr_last_timing",
    "swfr_band_slab_bytes", "swfr_copy_band_slab", "swfr_device_framebuffer", "swfr_debug_copy", "swfr_last_path_timing", "swfr_get_stats",
    "swfr_render_sequence", "swfr_set_targets", "swfr_render_resident_async", "swfr_stream_handle", "swfr_wait",
    "swfr_render_resident_batched", "swfr_read_image_async", "swfr_read_image_wait", "swfr_render_sequence_readback",
    "swfr_register_bitmap_tag", "swfr_decode_x_swf_bmp", "swfr_render_resident_async_to", "swfr_render_resident_group_to",
]


class SwfrError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("swfr error %d: %s" % (code, message))
        self.code = code


# ---- struct mirrors of include/swfr.h ---------------------------------------------------------
class Rgba8(C.Structure):
    _fields_ = [("r", C.c_uint8), ("g", C.c_uint8), ("b", C.c_uint8), ("a", C.c_uint8)]


class Rect(C.Structure):
    _fields_ = [("x_min", C.c_int32), ("x_max", C.c_int32), ("y_min", C.c_int32), ("y_max", C.c_int32)]


class Matrix(C.Structure):
    _fields_ = [("scale_x", C.c_int32), ("scale_y", C.c_int32), ("rotate_skew0", C.c_int32), ("rotate_skew1", C.c_int32),
                ("translate_x", C.c_int32), ("translate_y", C.c_int32)]


class ColorStop(C.Structure):
    _fields_ = [("ratio", C.c_uint8), ("color", Rgba8), ("morph_color", Rgba8)]


class FillStyle(C.Structure):
    _fields_ = [("type", C.c_uint32), ("color", Rgba8), ("morph_color", Rgba8), ("matrix", Matrix), ("n_stops", C.c_uint32),
                ("stops", C.POINTER(ColorStop)), ("focal_point", C.c_int32), ("bitmap_id", C.c_uint32),
                ("repeating", C.c_uint8), ("smoothed", C.c_uint8)]


class LineStyle(C.Structure):
    _fields_ = [("width", C.c_uint32), ("morph_width", C.c_uint32), ("fill", FillStyle)]


class Styles(C.Structure):
    _fields_ = [("n_fill", C.c_uint32), ("fill", C.POINTER(FillStyle)), ("n_line", C.c_uint32), ("line", C.POINTER(LineStyle))]


class ShapeRecord(C.Structure):
    _fields_ = [("type", C.c_uint32), ("delta_x", C.c_int32), ("delta_y", C.c_int32),
                ("has_control_delta", C.c_uint8), ("control_delta_x", C.c_int32), ("control_delta_y", C.c_int32),
                ("morph_delta_x", C.c_int32), ("morph_delta_y", C.c_int32),
                ("has_morph_control_delta", C.c_uint8), ("morph_control_delta_x", C.c_int32), ("morph_control_delta_y", C.c_int32),
                ("has_move_to", C.c_uint8), ("move_to_x", C.c_int32), ("move_to_y", C.c_int32),
                ("has_morph_move_to", C.c_uint8), ("morph_move_to_x", C.c_int32), ("morph_move_to_y", C.c_int32),
                ("has_left_fill", C.c_uint8), ("left_fill", C.c_uint32),
                ("has_right_fill", C.c_uint8), ("right_fill", C.c_uint32),
                ("has_line_style", C.c_uint8), ("line_style", C.c_uint32),
                ("new_styles", C.POINTER(Styles))]


class DefineShape(C.Structure):
    _fields_ = [("id", C.c_uint32), ("bounds", Rect), ("morph_bounds", Rect), ("initial_styles", Styles),
                ("n_records", C.c_uint32), ("records", C.POINTER(ShapeRecord))]


class DisplayObject(C.Structure):
    pass


DisplayObject._fields_ = [("type", C.c_uint32), ("id", C.c_uint32), ("has_matrix", C.c_uint8), ("matrix", Matrix),
                          ("ratio", C.c_double), ("n_children", C.c_uint32), ("children", C.POINTER(DisplayObject))]


class Stage(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("background_color", Rgba8), ("n_children", C.c_uint32),
                ("children", C.POINTER(DisplayObject))]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32), ("band_index", C.c_uint32), ("band_count", C.c_uint32)]


class Edge(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("x1", "y1", "x2", "y2", "top", "bottom", "dir", "reserved")]


class Path(C.Structure):
    _fields_ = [("first_edge", C.c_uint32), ("n_edges", C.c_uint32), ("kind", C.c_uint32), ("fill_rule", C.c_uint32),
                ("style", C.c_uint32), ("lerp", C.c_uint32), ("x_min", C.c_int32), ("y_min", C.c_int32),
                ("x_max", C.c_int32), ("y_max", C.c_int32)]


class Style(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("pixel", C.c_uint32), ("inv", C.c_double * 6),
                ("c0x", C.c_double), ("c0y", C.c_double), ("r0", C.c_double), ("c1x", C.c_double), ("c1y", C.c_double), ("r1", C.c_double),
                ("n_stops", C.c_uint32), ("stop_offset", C.c_float * MAX_STOPS), ("stop_rgba", (C.c_float * 4) * MAX_STOPS),
                ("bitmap", C.c_uint32), ("extend", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("setup_ms", C.c_float), ("rows_ms", C.c_float), ("tiles_ms", C.c_float),
                ("frames", C.c_uint32), ("n_edges", C.c_uint64), ("n_paths", C.c_uint64), ("n_row_tasks", C.c_uint64),
                ("n_records", C.c_uint64), ("timed_frames", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("frames", "queued_rows", "crowded_rows", "tie_rows", "pairtest_limit", "start_group_limit",
                                          "history_limit", "reserved")]


class PathTiming(C.Structure):
    _fields_ = [("build_ms", C.c_double), ("upload_host_ms", C.c_double), ("h2d_ms", C.c_double), ("device_ms", C.c_double),
                ("total_ms", C.c_double), ("h2d_bytes", C.c_uint64)]


EDGE_DTYPE = np.dtype([(n, "<i4") for n in ("x1", "y1", "x2", "y2", "top", "bottom", "dir", "reserved")])
PATH_DTYPE = np.dtype([("first_edge", "<u4"), ("n_edges", "<u4"), ("kind", "<u4"), ("fill_rule", "<u4"), ("style", "<u4"),
                       ("lerp", "<u4"), ("x_min", "<i4"), ("y_min", "<i4"), ("x_max", "<i4"), ("y_max", "<i4")])
assert EDGE_DTYPE.itemsize == C.sizeof(Edge) and PATH_DTYPE.itemsize == C.sizeof(Path)


def library_path() -> str:
    return os.path.join(_HERE, "libswfr.so")


def load_library():
    """Loads libswfr.so; fails loudly when the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("libswfr.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(needs hipcc); there is no CPU fallback for rasterization")
    L = C.CDLL(path)
    P, I, U = C.c_void_p, C.c_int, C.c_uint32
    L.swfr_abi_version.restype = U
    L.swfr_create.restype = I
    L.swfr_create.argtypes = [U, U, C.POINTER(Config), C.POINTER(P)]
    L.swfr_destroy.restype = None
    L.swfr_destroy.argtypes = [P]
    L.swfr_last_error.restype = C.c_char_p
    L.swfr_last_error.argtypes = [P]
    for fn in ("swfr_register_shape", "swfr_register_morph_shape"):
        getattr(L, fn).restype = I
        getattr(L, fn).argtypes = [P, C.POINTER(DefineShape), C.POINTER(U)]
    L.swfr_register_bitmap.restype = I
    L.swfr_register_bitmap.argtypes = [P, U, U, U, P, C.c_size_t]
    L.swfr_register_bitmap_tag.restype = I
    L.swfr_register_bitmap_tag.argtypes = [P, U, C.c_char_p, P, C.c_size_t]
    L.swfr_decode_x_swf_bmp.restype = I
    L.swfr_decode_x_swf_bmp.argtypes = [P, C.c_size_t, C.POINTER(U), C.POINTER(U), P, C.c_size_t]
    L.swfr_render.restype = I
    L.swfr_render.argtypes = [P, C.POINTER(Stage)]
    L.swfr_render_batch.restype = I
    L.swfr_render_batch.argtypes = [P, C.POINTER(Stage), C.c_uint32, C.c_void_p, C.c_size_t]
    L.swfr_read_image.restype = I
    L.swfr_read_image.argtypes = [P, P, C.c_size_t, I]
    for fn in ("swfr_upload_edges", "swfr_render_edges"):
        getattr(L, fn).restype = I
        getattr(L, fn).argtypes = [P, P, C.c_size_t, P, C.c_size_t, P, C.c_size_t]
    L.swfr_render_resident.restype = I
    L.swfr_render_resident.argtypes = [P, U]
    L.swfr_render_resident_batched.restype = I
    L.swfr_render_resident_batched.argtypes = [P, U, U, C.POINTER(C.c_float)]
    L.swfr_read_image_async.restype = I
    L.swfr_read_image_async.argtypes = [P, I]
    L.swfr_read_image_wait.restype = I
    L.swfr_read_image_wait.argtypes = [P, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    L.swfr_render_sequence_readback.restype = I
    L.swfr_render_sequence_readback.argtypes = [P, C.POINTER(Stage), U, U, I, I, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.swfr_build_frame.restype = I
    L.swfr_build_frame.argtypes = [P, C.POINTER(Stage), C.POINTER(P), C.POINTER(C.c_size_t), C.POINTER(P), C.POINTER(C.c_size_t),
                                   C.POINTER(P), C.POINTER(C.c_size_t)]
    L.swfr_shape_json.restype = I
    L.swfr_shape_json.argtypes = [P, U, I, C.POINTER(C.c_char_p)]
    L.swfr_last_timing.restype = I
    L.swfr_last_timing.argtypes = [P, C.POINTER(Timing)]
    L.swfr_band_slab_bytes.restype = C.c_size_t
    L.swfr_band_slab_bytes.argtypes = [P]
    L.swfr_copy_band_slab.restype = I
    L.swfr_copy_band_slab.argtypes = [P, P]
    L.swfr_debug_copy.restype = C.c_long
    L.swfr_debug_copy.argtypes = [P, I, P, C.c_size_t]
    L.swfr_device_framebuffer.restype = P
    L.swfr_device_framebuffer.argtypes = [P]
    L.swfr_set_targets.restype = I
    L.swfr_set_targets.argtypes = [P, C.POINTER(C.c_void_p), U]
    L.swfr_render_resident_async.restype = I
    L.swfr_render_resident_async.argtypes = [P, C.POINTER(U)]
    L.swfr_render_resident_async_to.restype = I
    L.swfr_render_resident_async_to.argtypes = [P, P, C.POINTER(U)]
    L.swfr_render_resident_group_to.restype = I
    L.swfr_render_resident_group_to.argtypes = [P, P, U, C.POINTER(U)]
    L.swfr_stream_handle.restype = P
    L.swfr_stream_handle.argtypes = [P, U]
    L.swfr_wait.restype = I
    L.swfr_wait.argtypes = [P]
    L.swfr_get_stats.restype = I
    L.swfr_get_stats.argtypes = [P, C.POINTER(Stats)]
    L.swfr_last_path_timing.restype = I
    L.swfr_last_path_timing.argtypes = [P, C.POINTER(PathTiming)]
    L.swfr_render_sequence.restype = I
    L.swfr_render_sequence.argtypes = [P, C.POINTER(Stage), U, U, C.POINTER(C.c_double), C.POINTER(PathTiming)]
    if L.swfr_abi_version() != 1:
        raise ImportError("libswfr.so ABI mismatch")
    _LIB = L
    return L


# ---- swf-tree JSON -> C structs ------------------------------------------------------------------
_FILL_TYPES = {"solid": 0, "linear-gradient": 1, "radial-gradient": 2, "focal-gradient": 3, "bitmap": 4}


class _Arena:
    """Keeps every ctypes object referenced by pointer alive for the duration of one call."""

    def __init__(self):
        self.keep = []

    def array(self, ctype, items):
        arr = (ctype * max(len(items), 1))(*items)
        self.keep.append(arr)
        return arr


def _rgba(c):
    return Rgba8(c["r"], c["g"], c["b"], c["a"])


def _matrix(m):
    return Matrix(m["scale_x"], m["scale_y"], m["rotate_skew0"], m["rotate_skew1"], m["translate_x"], m["translate_y"])


def _fill(arena, s):
    if s["type"] not in _FILL_TYPES:
        raise SwfrError(ERR_INVALID, "UnknownFillStyle")
    f = FillStyle()
    f.type = _FILL_TYPES[s["type"]]
    if "color" in s:
        f.color = _rgba(s["color"])
        f.morph_color = _rgba(s.get("morph_color", s["color"]))
    if "matrix" in s:
        f.matrix = _matrix(s["matrix"])
    if "gradient" in s:
        stops = [ColorStop(c["ratio"], _rgba(c["color"]), _rgba(c.get("morph_color", c["color"]))) for c in s["gradient"]["colors"]]
        arr = arena.array(ColorStop, stops)
        f.n_stops = len(stops)
        f.stops = C.cast(arr, C.POINTER(ColorStop))
    if "focal_point" in s:
        fp = s["focal_point"]
        f.focal_point = int(fp["epsilons"]) if isinstance(fp, dict) else int(round(float(fp) * 256))
    if "bitmap_id" in s:
        f.bitmap_id = s["bitmap_id"]
        f.repeating = 1 if s.get("repeating") else 0
        f.smoothed = 1 if s.get("smoothed") else 0
    return f


def _styles(arena, fills, lines):
    st = Styles()
    fa = arena.array(FillStyle, [_fill(arena, f) for f in fills])
    la = arena.array(LineStyle, [LineStyle(l["width"], l.get("morph_width", l["width"]), _fill(arena, l["fill"])) for l in lines])
    st.n_fill, st.fill = len(fills), C.cast(fa, C.POINTER(FillStyle))
    st.n_line, st.line = len(lines), C.cast(la, C.POINTER(LineStyle))
    return st


def _define_shape(arena, tag):
    d = DefineShape()
    d.id = tag.get("id", 0)
    b = tag["bounds"]
    d.bounds = Rect(b["x_min"], b["x_max"], b["y_min"], b["y_max"])
    mb = tag.get("morph_bounds", b)
    d.morph_bounds = Rect(mb["x_min"], mb["x_max"], mb["y_min"], mb["y_max"])
    ini = tag["shape"]["initial_styles"]
    d.initial_styles = _styles(arena, ini["fill"], ini["line"])
    recs = []
    for r in tag["shape"]["records"]:
        rec = ShapeRecord()
        if r["type"] == "edge":
            rec.type = 0
            rec.delta_x, rec.delta_y = r["delta"]["x"], r["delta"]["y"]
            md = r.get("morph_delta", r["delta"])
            rec.morph_delta_x, rec.morph_delta_y = md["x"], md["y"]
            if r.get("control_delta") is not None:
                rec.has_control_delta = 1
                rec.control_delta_x, rec.control_delta_y = r["control_delta"]["x"], r["control_delta"]["y"]
            if r.get("morph_control_delta") is not None:
                rec.has_morph_control_delta = 1
                rec.morph_control_delta_x, rec.morph_control_delta_y = r["morph_control_delta"]["x"], r["morph_control_delta"]["y"]
        elif r["type"] == "style-change":
            rec.type = 1
            if r.get("move_to") is not None:
                rec.has_move_to = 1
                rec.move_to_x, rec.move_to_y = r["move_to"]["x"], r["move_to"]["y"]
            if r.get("morph_move_to") is not None:
                rec.has_morph_move_to = 1
                rec.morph_move_to_x, rec.morph_move_to_y = r["morph_move_to"]["x"], r["morph_move_to"]["y"]
            for key in ("left_fill", "right_fill", "line_style"):
                if r.get(key) is not None:
                    setattr(rec, "has_" + key, 1)
                    setattr(rec, key, r[key])
            if r.get("new_styles") is not None:
                ns = _styles(arena, r["new_styles"]["fill"], r["new_styles"]["line"])
                arena.keep.append(ns)
                rec.new_styles = C.pointer(ns)
        else:
            raise SwfrError(ERR_INVALID, "UnreachableCode")
        recs.append(rec)
    ra = arena.array(ShapeRecord, recs)
    d.n_records, d.records = len(recs), C.cast(ra, C.POINTER(ShapeRecord))
    return d


def image_to_pam(width: int, height: int, rgba: bytes) -> bytes:
    """Straight RGBA8 -> Netpbm PAM (`P7`, `TUPLTYPE RGB_ALPHA`), byte for byte what the reference's writers emit
    (rs/src/pam.rs:3-34, ts/src/lib/image-data-to-pam.ts:8-28)."""
    rgba = bytes(rgba)
    if len(rgba) != width * height * 4:
        raise ValueError("image_to_pam: data length does not match the dimensions")
    header = "P7\nWIDTH %d\nHEIGHT %d\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n" % (width, height)
    return header.encode("ascii") + rgba


def image_to_png(width: int, height: int, rgba: bytes) -> bytes:
    """Straight RGBA8 -> PNG (8-bit RGBA, filter 0 on every row): what the reference's test harness keeps next to its goldens
    (`canvas.toBuffer("image/png")`, ts/src/test/node-canvas-renderer.spec.ts:67-84); the pixels round-trip exactly."""
    import struct
    import zlib
    rgba = bytes(rgba)
    if len(rgba) != width * height * 4:
        raise ValueError("image_to_png: data length does not match the dimensions")

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    rows = b"".join(b"\x00" + rgba[y * width * 4:(y + 1) * width * 4] for y in range(height))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(rows, 6)) + chunk(b"IEND", b""))


def decode_x_swf_bmp(data: bytes):
    """`image/x-swf-bmp` format 3 (zlib colour-mapped) -> (width, height, straight RGBA bytes).

    decodeXSwfBmpSync (ts/src/lib/decode-x-swf-bmp.ts:9-41) as libswfr.so restates it (csrc/bitmap_decode.cpp, its own
    inflater): rows are padded to 4 bytes, the palette is opaque, out-of-range indices are opaque black.  Needs no device.
    """
    L = load_library()
    data = bytes(data)
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    w, h = C.c_uint32(), C.c_uint32()
    rc = L.swfr_decode_x_swf_bmp(C.cast(buf, C.c_void_p), len(data), C.byref(w), C.byref(h), None, 0)
    if rc == ERR_NOT_IMPLEMENTED:
        raise SwfrError(rc, "UnsupportedXSwfBmpFormatId: %d" % (data[0] if data else -1))
    if rc != OK:
        raise SwfrError(rc, "corrupt image/x-swf-bmp data")
    out = (C.c_uint8 * (w.value * h.value * 4))()
    rc = L.swfr_decode_x_swf_bmp(C.cast(buf, C.c_void_p), len(data), C.byref(w), C.byref(h), C.cast(out, C.c_void_p), len(out))
    if rc != OK:
        raise SwfrError(rc, "corrupt image/x-swf-bmp data")
    return w.value, h.value, bytes(out)


class Renderer:
    """`new NodeCanvasRenderer(width, height)` + `Renderer{render, addBitmap}` over libswfr.so."""

    def __init__(self, width, height, device=0, even_odd=False, band_index=0, band_count=0, contiguous_bands=False):
        self.L = load_library()
        self.width, self.height = int(width), int(height)
        cfg = Config(int(device), (FLAG_EVEN_ODD if even_odd else 0) | (FLAG_BANDS_CONTIGUOUS if contiguous_bands else 0), band_index, band_count)
        h = C.c_void_p()
        rc = self.L.swfr_create(self.width, self.height, C.byref(cfg), C.byref(h))
        if rc != OK:
            raise SwfrError(rc, "swfr_create failed (no HIP device?)" if rc == ERR_NO_DEVICE else "swfr_create failed")
        self.h = h
        self._ids = {}        # id(tag) -> (registered id, tag)  [the reference's WeakMap caches]

    def close(self):
        if getattr(self, "h", None):
            self.L.swfr_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc != OK:
            raise SwfrError(rc, self.L.swfr_last_error(self.h).decode("utf-8", "replace"))

    # -- assets
    def register_shape(self, tag) -> int:
        arena = _Arena()
        d = _define_shape(arena, tag)
        out = C.c_uint32()
        self._check(self.L.swfr_register_shape(self.h, C.byref(d), C.byref(out)))
        return out.value

    def register_morph_shape(self, tag) -> int:
        arena = _Arena()
        d = _define_shape(arena, tag)
        out = C.c_uint32()
        self._check(self.L.swfr_register_morph_shape(self.h, C.byref(d), C.byref(out)))
        return out.value

    def add_bitmap(self, tag):
        """Renderer.addBitmap(tag: DefineBitmap)."""
        data = tag["data"]
        if isinstance(data, str):
            data = bytes.fromhex(data)
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        self._check(self.L.swfr_register_bitmap_tag(self.h, tag["id"], tag["media_type"].encode("utf-8"), C.cast(buf, C.c_void_p), len(data)))

    def register_bitmap(self, bitmap_id, width, height, rgba_straight: bytes):
        buf = (C.c_uint8 * len(rgba_straight)).from_buffer_copy(rgba_straight)
        self._check(self.L.swfr_register_bitmap(self.h, bitmap_id, width, height, C.cast(buf, C.c_void_p), width * 4))

    def shape_json(self, shape_id, morph=False) -> str:
        out = C.c_char_p()
        self._check(self.L.swfr_shape_json(self.h, shape_id, 1 if morph else 0, C.byref(out)))
        return out.value.decode("utf-8")

    # -- stage
    def _object(self, arena, obj):
        d = DisplayObject()
        t = obj["type"]
        if obj.get("matrix") is not None:
            d.has_matrix = 1
            d.matrix = _matrix(obj["matrix"])
        if t == "container":
            d.type = 2
            kids = arena.array(DisplayObject, [self._object(arena, c) for c in obj["children"]])
            d.n_children, d.children = len(obj["children"]), C.cast(kids, C.POINTER(DisplayObject))
            return d
        if t not in ("shape", "morph-shape"):
            raise SwfrError(ERR_INVALID, "UnexpectedDisplayObjectType")
        morph = t == "morph-shape"
        d.type = 1 if morph else 0
        if "definition" in obj:
            key = (id(obj["definition"]), morph)
            if key not in self._ids:
                reg = self.register_morph_shape if morph else self.register_shape
                self._ids[key] = (reg(obj["definition"]), obj["definition"])
            d.id = self._ids[key][0]
        else:
            d.id = obj["id"]
        if morph:
            d.ratio = float(obj["ratio"])
        return d

    def _stage(self, arena, stage):
        s = Stage()
        s.width, s.height = stage.get("width", self.width), stage.get("height", self.height)
        bg = stage.get("background_color") or {"r": 0, "g": 0, "b": 0, "a": 0}
        s.background_color = _rgba(bg)
        kids = arena.array(DisplayObject, [self._object(arena, c) for c in stage["children"]])
        s.n_children, s.children = len(stage["children"]), C.cast(kids, C.POINTER(DisplayObject))
        return s

    def render(self, stage):
        """Renderer.render(stage): blocking; the image stays in HBM until read_image()."""
        arena = _Arena()
        s = self._stage(arena, stage)
        self._check(self.L.swfr_render(self.h, C.byref(s)))

    def render_sequence(self, stages, repeat=1):
        """The reference's animation loop, timed below the C-ABI: `repeat` x (swfr_render of every stage in turn, each blocking).
        Returns (seconds, dict of the per-frame times added up)."""
        arena = _Arena()
        arr = (Stage * max(len(stages), 1))(*[self._stage(arena, st) for st in stages])
        secs, acc = C.c_double(), PathTiming()
        self._check(self.L.swfr_render_sequence(self.h, arr, len(stages), int(repeat), C.byref(secs), C.byref(acc)))
        return secs.value, {n: getattr(acc, n) for n, _ in PathTiming._fields_}

    def stats(self) -> dict:
        """Rows the general row kernels handled and capacity limits reached since the handle was created (swfr_stats)."""
        t = Stats()
        self._check(self.L.swfr_get_stats(self.h, C.byref(t)))
        return {n: int(getattr(t, n)) for n, _ in Stats._fields_ if n != "reserved"}

    def path_timing(self) -> dict:
        t = PathTiming()
        self.L.swfr_last_path_timing(self.h, C.byref(t))
        return {n: getattr(t, n) for n, _ in PathTiming._fields_}

    def marshal_stages(self, stages):
        """The ctypes form of a list of stages, reusable across render_batch / render_sequence calls (marshalling a Stage of a
        thousand display objects costs Python half a millisecond; a native caller has the structs already)."""
        arena = _Arena()
        arr = (Stage * max(len(stages), 1))(*[self._stage(arena, st) for st in stages])
        return (arena, arr, len(stages))

    def render_batch(self, stages, device_ptr=None, frame_stride=0):
        """A batch of different frames in one call (e.g. the 256 ratios of a morph shape): with a device destination the frames
        are rendered in groups, one launch per kernel and group, the host building one group while the GPU renders the other.
        Frame i lands at device_ptr + i * frame_stride (device memory: pass tensor.data_ptr()); without a destination only the
        last frame is kept for read_image().  `stages`: a list of stage dicts, or the result of marshal_stages().  Blocking."""
        _, arr, n = stages if isinstance(stages, tuple) else self.marshal_stages(stages)
        self._check(self.L.swfr_render_batch(self.h, arr, n, C.c_void_p(device_ptr) if device_ptr else None,
                                             int(frame_stride) if device_ptr else 0))

    def build_frame(self, stage):
        """Host half only: (edges, paths, styles) exactly as render() would upload them."""
        arena = _Arena()
        s = self._stage(arena, stage)
        pe, pp, ps = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ne, npth, ns = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._check(self.L.swfr_build_frame(self.h, C.byref(s), C.byref(pe), C.byref(ne), C.byref(pp), C.byref(npth),
                                            C.byref(ps), C.byref(ns)))
        edges = np.frombuffer((C.c_char * (ne.value * EDGE_DTYPE.itemsize)).from_address(pe.value), dtype=EDGE_DTYPE).copy() \
            if ne.value else np.zeros(0, EDGE_DTYPE)
        paths = np.frombuffer((C.c_char * (npth.value * PATH_DTYPE.itemsize)).from_address(pp.value), dtype=PATH_DTYPE).copy() \
            if npth.value else np.zeros(0, PATH_DTYPE)
        styles = [Style.from_buffer_copy((C.c_char * C.sizeof(Style)).from_address(ps.value + i * C.sizeof(Style)))
                  for i in range(ns.value)]
        return edges, paths, styles

    # -- low level (the hot path proper)
    @staticmethod
    def _styles_array(styles):
        arr = (Style * max(len(styles), 1))(*styles)
        return arr

    def upload_edges(self, edges: np.ndarray, paths: np.ndarray, styles):
        edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        paths = np.ascontiguousarray(paths, dtype=PATH_DTYPE)
        sarr = self._styles_array(styles)
        self._check(self.L.swfr_upload_edges(self.h, edges.ctypes.data, len(edges), paths.ctypes.data, len(paths),
                                             C.cast(sarr, C.c_void_p), len(styles)))

    def render_resident(self, frames=1):
        self._check(self.L.swfr_render_resident(self.h, frames))

    def render_resident_batched(self, frames_per_launch: int, launches: int) -> float:
        """The resident scene as `frames_per_launch` frames per kernel launch, `launches` launches; returns their HIP-event time in ms."""
        ms = C.c_float()
        self._check(self.L.swfr_render_resident_batched(self.h, frames_per_launch, launches, C.byref(ms)))
        return float(ms.value)

    def render_edges(self, edges, paths, styles):
        self.upload_edges(edges, paths, styles)
        self.render_resident(1)

    def timing(self) -> dict:
        t = Timing()
        self.L.swfr_last_timing(self.h, C.byref(t))
        return {n: getattr(t, n) for n, _ in Timing._fields_}

    def write_pam(self, path: str) -> None:
        """Dumps the last frame (straight RGBA) as PAM, like the reference's headless renderer does for its tests."""
        img = self.read_image(premultiplied=False)
        with open(path, "wb") as f:
            f.write(image_to_pam(self.width, self.height, np.ascontiguousarray(img).tobytes()))

    def write_png(self, path: str) -> None:
        """Dumps the last frame (straight RGBA) as PNG -- the format of the reference's golden images."""
        img = self.read_image(premultiplied=False)
        with open(path, "wb") as f:
            f.write(image_to_png(self.width, self.height, np.ascontiguousarray(img).tobytes()))

    def read_image(self, premultiplied=False) -> np.ndarray:
        """HxWx4 uint8 RGBA; straight (as the reference's PNG/getImageData) unless premultiplied=True."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._check(self.L.swfr_read_image(self.h, out.ctypes.data, self.width * 4, 1 if premultiplied else 0))
        return out

    def read_image_async(self, premultiplied=False) -> None:
        """Queues the read-back of the last frame into the handle's pinned staging buffer, behind the frame's kernels."""
        self._check(self.L.swfr_read_image_async(self.h, 1 if premultiplied else 0))

    def read_image_wait(self) -> np.ndarray:
        """Waits for read_image_async; returns the pinned staging buffer as an HxWx4 uint8 view (valid until the next read-back)."""
        data, stride = C.POINTER(C.c_uint8)(), C.c_size_t()
        self._check(self.L.swfr_read_image_wait(self.h, C.byref(data), C.byref(stride)))
        return np.ctypeslib.as_array(data, shape=(self.height, self.width, 4))

    def render_sequence_readback(self, stages, repeat=1, premultiplied=False, overlap=True):
        """render + mapped read-back of every frame (the reference's test loop), timed below the C-ABI; returns seconds."""
        _, arr, n = stages if isinstance(stages, tuple) else self.marshal_stages(stages)
        secs, chk = C.c_double(), C.c_uint64()
        self._check(self.L.swfr_render_sequence_readback(self.h, arr, n, int(repeat), 1 if premultiplied else 0, 1 if overlap else 0, C.byref(secs), C.byref(chk)))
        return secs.value

    def band_slab_bytes(self) -> int:
        return self.L.swfr_band_slab_bytes(self.h)

    def copy_band_slab(self, device_ptr: int):
        self._check(self.L.swfr_copy_band_slab(self.h, device_ptr))

    def set_targets(self, device_ptrs):
        """Frame set k renders into device_ptrs[k] (full-frame device buffers, e.g. tensor.data_ptr()); upload the scene afterwards."""
        arr = (C.c_void_p * max(len(device_ptrs), 1))(*[C.c_void_p(p) for p in device_ptrs])
        self._check(self.L.swfr_set_targets(self.h, arr, len(device_ptrs)))

    def render_resident_async(self) -> int:
        """Queues one frame of the resident scene without waiting; returns the frame set (target / stream) it runs on."""
        k = C.c_uint32()
        self._check(self.L.swfr_render_resident_async(self.h, C.byref(k)))
        return int(k.value)

    def render_resident_async_to(self, block_ptr: int) -> int:
        """Queues one frame whose pixels go to the device buffer at block_ptr -- this handle's block of tile-rows only (contiguous
        bands); returns the frame set (stream) it runs on."""
        k = C.c_uint32()
        self._check(self.L.swfr_render_resident_async_to(self.h, C.c_void_p(block_ptr), C.byref(k)))
        return int(k.value)

    def render_resident_group_to(self, block_ptrs) -> int:
        """Queues len(block_ptrs) frames, frame i into the block buffer at block_ptrs[i], in one call below Python; returns the bit
        mask of the frame sets (streams) that carry them."""
        arr = (C.c_void_p * len(block_ptrs))(*[C.c_void_p(p) for p in block_ptrs])
        used = C.c_uint32()
        self._check(self.L.swfr_render_resident_group_to(self.h, arr, len(block_ptrs), C.byref(used)))
        return int(used.value)

    def stream_handle(self, frame_set: int) -> int:
        return int(self.L.swfr_stream_handle(self.h, frame_set) or 0)

    def wait(self):
        self._check(self.L.swfr_wait(self.h))

    def band_slab(self) -> np.ndarray:
        """This handle's packed tile-rows ([rows, width, 4] uint8, premultiplied) through swfr_copy_band_slab and a device
        buffer -- what a rank hands to the gather."""
        import torch
        rows = self.band_slab_bytes() // (self.width * 4)
        t = torch.empty((rows, self.width, 4), dtype=torch.uint8, device="cuda")   # (every byte is written by the copy, padding included)
        torch.cuda.current_stream().synchronize()                                  # nothing of torch's still pending on that memory
        self.copy_band_slab(t.data_ptr())
        torch.cuda.synchronize()
        return t.cpu().numpy()


def solid_style(pixel_argb_premultiplied: int) -> Style:
    s = Style()
    s.kind = STYLE_SOLID
    s.pixel = pixel_argb_premultiplied
    return s


def _merge_collinear(poly):
    """Vertices of a closed polyline p0..pn-1 (explicit final line back to p0) after Cairo's
    line_to merge: a vertex is dropped when the segments before and after it have equal slope and
    do not reverse.  p0 is never dropped (the path starts there)."""
    pts = [tuple(int(v) for v in poly[0])]
    seq = [tuple(int(v) for v in q) for q in poly[1:]] + [pts[0]]
    for q in seq:
        if q == pts[-1]:
            continue
        if len(pts) >= 2:
            ax, ay = pts[-1][0] - pts[-2][0], pts[-1][1] - pts[-2][1]
            bx, by = q[0] - pts[-1][0], q[1] - pts[-1][1]
            if ay * bx == by * ax and not (((ax * bx) >> 8) + ((ay * by) >> 8) < 0):
                pts.pop()
        pts.append(q)
    return pts  # last == first


def polygons_to_scene(fixed_xy: np.ndarray, colors_rgba8: np.ndarray, width: int, height: int):
    """Polygons in 24.8 device coordinates (all inside the frame), drawn the way the reference draws a
    one-fill shape (moveTo p0, lineTo p1..pn-1, lineTo p0, fill) -> (edges, paths, styles).

    This is the edge list the host produces for the synthetic benchmark scenes (opaque solid stars);
    tests check it against swfr_build_frame on the same shapes.  fixed_xy: int32 [n, verts, 2].
    """
    n = fixed_xy.shape[0]
    edge_rows, counts = [], []
    for i in range(n):
        pts = _merge_collinear(fixed_xy[i])
        c = 0
        for a, b in zip(pts[:-1], pts[1:]):
            if a[1] == b[1]:
                continue
            if a[1] < b[1]:
                edge_rows.append((a[0], a[1], b[0], b[1], a[1], b[1], 1, 0))
            else:
                edge_rows.append((b[0], b[1], a[0], a[1], b[1], a[1], -1, 0))
            c += 1
        counts.append(c)
    edges = np.array(edge_rows, dtype=np.int32).view(EDGE_DTYPE).reshape(-1) if edge_rows else np.zeros(0, EDGE_DTYPE)
    counts = np.array(counts, dtype=np.uint32)
    edges["reserved"] = np.repeat(np.arange(int((counts > 0).sum()), dtype=np.int32), counts[counts > 0])   # owning path (paths without edges are dropped below)
    paths = np.zeros(n, dtype=PATH_DTYPE)
    paths["first_edge"] = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32)
    paths["n_edges"] = counts
    paths["kind"] = PATH_TOR
    paths["style"] = np.arange(n, dtype=np.uint32)
    paths["lerp"] = 1
    paths["x_min"] = np.maximum(fixed_xy[..., 0].min(axis=1) >> 8, 0)
    paths["y_min"] = np.maximum(fixed_xy[..., 1].min(axis=1) >> 8, 0)
    paths["x_max"] = np.minimum((fixed_xy[..., 0].max(axis=1) + 255) >> 8, width)
    paths["y_max"] = np.minimum((fixed_xy[..., 1].max(axis=1) + 255) >> 8, height)
    c = colors_rgba8.astype(np.uint32)
    af = c[:, 3] / 255.0

    def sh(v):  # cairo_set_source_rgba: premultiply in doubles, 16-bit shorts, >> 8
        return ((v * 65535.0 + 0.5).astype(np.uint32) & 0xFFFF) >> 8

    pix = (sh(af) << 24) | (sh(c[:, 0] / 255.0 * af) << 16) | (sh(c[:, 1] / 255.0 * af) << 8) | sh(c[:, 2] / 255.0 * af)
    styles = [solid_style(int(p)) for p in pix]
    keep = counts > 0
    return edges, paths[keep], styles


def stars_to_stage(twips: np.ndarray, colors_rgba8: np.ndarray):
    """The same polygons as swf-tree DefineShape tags + a Stage (one shape per polygon, painter's order =
    index): the input of the full reference-API path (register_shape + render)."""
    children = []
    for i in range(twips.shape[0]):
        p = twips[i]
        recs = [{"type": "style-change", "move_to": {"x": int(p[0, 0]), "y": int(p[0, 1])}, "left_fill": 1}]
        n = p.shape[0]
        for k in range(1, n + 1):
            a, b = p[k - 1], p[k % n]
            recs.append({"type": "edge", "delta": {"x": int(b[0] - a[0]), "y": int(b[1] - a[1])}})
        col = {"r": int(colors_rgba8[i, 0]), "g": int(colors_rgba8[i, 1]), "b": int(colors_rgba8[i, 2]), "a": int(colors_rgba8[i, 3])}
        tag = {"id": i + 1, "bounds": {"x_min": int(p[:, 0].min()), "x_max": int(p[:, 0].max()),
                                        "y_min": int(p[:, 1].min()), "y_max": int(p[:, 1].max())},
               "shape": {"initial_styles": {"fill": [{"type": "solid", "color": col}], "line": []}, "records": recs}}
        children.append({"type": "shape", "definition": tag})
    return {"children": children}
