"""TEST INFRASTRUCTURE ONLY -- Canvas2D backend over the system libcairo (ctypes).

node-canvas 2.6.1 (ts/yarn.lock:835-837) maps every Canvas2D call the reference makes
(ts/src/lib/renderers/canvas-renderer.ts:69-350) onto Cairo 1.16 / pixman; that third-party
code is the arithmetic of the hot path and is NOT under /root/reference.  This backend replays
the same calls into the container's `libcairo.so.2` (1.16.0) so that

  * the C restatement in oracle/swfr_oracle.c can be pinned by fuzzing, and
  * golden vectors can be generated (tools/make_goldens.py) and committed under tests/golden/.

It never ships, the product never imports it, and every test that uses it skips when the
library is absent (tests on the GPU box rely on the committed goldens only).
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

_lib = None


def available() -> bool:
    return _load() is not None


def _load():
    global _lib
    if _lib is not None:
        return _lib or None
    name = ctypes.util.find_library("cairo")
    if not name:
        _lib = False
        return None
    try:
        lib = ctypes.CDLL(name)
    except OSError:
        _lib = False
        return None
    c = ctypes
    P, D, I = c.c_void_p, c.c_double, c.c_int

    def sig(fn, res, *args):
        f = getattr(lib, fn)
        f.restype = res
        f.argtypes = list(args)

    sig("cairo_image_surface_create", P, I, I, I)
    sig("cairo_image_surface_create_for_data", P, P, I, I, I, I)
    sig("cairo_image_surface_get_data", c.POINTER(c.c_ubyte), P)
    sig("cairo_image_surface_get_stride", I, P)
    sig("cairo_surface_flush", None, P)
    sig("cairo_surface_mark_dirty", None, P)
    sig("cairo_surface_destroy", None, P)
    sig("cairo_create", P, P)
    sig("cairo_destroy", None, P)
    sig("cairo_save", None, P)
    sig("cairo_restore", None, P)
    sig("cairo_identity_matrix", None, P)
    sig("cairo_scale", None, P, D, D)
    sig("cairo_transform", None, P, P)
    sig("cairo_set_operator", None, P, I)
    sig("cairo_rectangle", None, P, D, D, D, D)
    sig("cairo_new_path", None, P)
    sig("cairo_move_to", None, P, D, D)
    sig("cairo_line_to", None, P, D, D)
    sig("cairo_curve_to", None, P, D, D, D, D, D, D)
    sig("cairo_close_path", None, P)
    sig("cairo_get_current_point", None, P, c.POINTER(D), c.POINTER(D))
    sig("cairo_set_source_rgba", None, P, D, D, D, D)
    sig("cairo_set_source", None, P, P)
    sig("cairo_fill", None, P)
    sig("cairo_fill_preserve", None, P)
    sig("cairo_stroke_preserve", None, P)
    sig("cairo_set_line_width", None, P, D)
    sig("cairo_set_line_cap", None, P, I)
    sig("cairo_set_line_join", None, P, I)
    sig("cairo_set_fill_rule", None, P, I)
    sig("cairo_pattern_create_for_surface", P, P)
    sig("cairo_pattern_create_radial", P, D, D, D, D, D, D)
    sig("cairo_pattern_create_linear", P, D, D, D, D)
    sig("cairo_pattern_add_color_stop_rgba", None, P, D, D, D, D, D)
    sig("cairo_pattern_set_extend", None, P, I)
    sig("cairo_pattern_set_filter", None, P, I)
    sig("cairo_pattern_destroy", None, P)
    sig("cairo_status", I, P)
    sig("cairo_version_string", c.c_char_p)
    _lib = lib
    return lib


class _Matrix(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("xx", "yx", "xy", "yy", "x0", "y0")]


def version() -> str:
    return _load().cairo_version_string().decode()


class CairoBackend:
    """Canvas2D subset used by CanvasRenderer, mapped as node-canvas 2.6.1 maps it."""

    def __init__(self, width, height):
        lib = _load()
        if lib is None:
            raise RuntimeError("libcairo not available")
        self.lib = lib
        self.w, self.h = width, height
        self.surf = lib.cairo_image_surface_create(0, width, height)  # CAIRO_FORMAT_ARGB32
        self.cr = lib.cairo_create(self.surf)
        lib.cairo_set_line_width(self.cr, 1.0)         # node-canvas Context2d ctor
        self._keep = []
        self._fill = ("rgba", 0, 0, 0, 255)
        self._stroke = ("rgba", 0, 0, 0, 255)

    def close(self):
        for kind, obj in self._keep:
            (self.lib.cairo_pattern_destroy if kind == "p" else self.lib.cairo_surface_destroy)(obj)
        self._keep = []
        self.lib.cairo_destroy(self.cr)
        self.lib.cairo_surface_destroy(self.surf)

    # -- transform / state
    def set_transform_identity(self):
        self.lib.cairo_identity_matrix(self.cr)

    def clear_all(self):
        # Context2d::ClearRect: save, operator CLEAR, rectangle, fill, restore
        lib, cr = self.lib, self.cr
        lib.cairo_save(cr)
        lib.cairo_set_operator(cr, 0)
        lib.cairo_rectangle(cr, 0, 0, self.w, self.h)
        lib.cairo_fill(cr)
        lib.cairo_restore(cr)

    def scale(self, sx, sy):
        self.lib.cairo_scale(self.cr, sx, sy)

    def transform(self, a, b, c, d, e, f):
        m = _Matrix(a, b, c, d, e, f)
        self.lib.cairo_transform(self.cr, ctypes.byref(m))

    def save(self):
        self.lib.cairo_save(self.cr)
        self._stack = getattr(self, "_stack", [])
        self._stack.append((self._fill, self._stroke))

    def restore(self):
        self.lib.cairo_restore(self.cr)
        self._fill, self._stroke = self._stack.pop()

    # -- path
    def begin_path(self):
        self.lib.cairo_new_path(self.cr)

    def move_to(self, x, y):
        self.lib.cairo_move_to(self.cr, x, y)

    def line_to(self, x, y):
        self.lib.cairo_line_to(self.cr, x, y)

    def close_path(self):
        self.lib.cairo_close_path(self.cr)

    def quadratic_curve_to(self, x1, y1, x2, y2):
        # Context2d::QuadraticCurveTo (node-canvas 2.6.1)
        x, y = ctypes.c_double(), ctypes.c_double()
        self.lib.cairo_get_current_point(self.cr, ctypes.byref(x), ctypes.byref(y))
        x, y = x.value, y.value
        if x == 0 and y == 0:
            x, y = x1, y1
        k = 2.0 / 3.0
        self.lib.cairo_curve_to(self.cr, x + k * (x1 - x), y + k * (y1 - y),
                                x2 + k * (x1 - x2), y2 + k * (y1 - y2), x2, y2)

    # -- sources
    def create_bitmap(self, w, h, rgba_straight: bytes):
        # putImageData: straight RGBA -> premultiplied ARGB32 (opaque palette bitmaps: alpha 255)
        src = np.frombuffer(rgba_straight, dtype=np.uint8).reshape(h, w, 4).astype(np.uint32)
        a = src[..., 3]
        pm = lambda ch: (src[..., ch] * a // 255)  # node-canvas: c * a / 255
        argb = (a << 24) | (pm(0) << 16) | (pm(1) << 8) | pm(2)
        surf = self.lib.cairo_image_surface_create(0, w, h)
        self.lib.cairo_surface_flush(surf)
        stride = self.lib.cairo_image_surface_get_stride(surf)
        ptr = self.lib.cairo_image_surface_get_data(surf)
        buf = np.ctypeslib.as_array(ptr, shape=(h, stride))
        buf[:, : w * 4] = argb.astype("<u4").view(np.uint8).reshape(h, w * 4)
        self.lib.cairo_surface_mark_dirty(surf)
        self._keep.append(("s", surf))
        return surf

    def set_fill_rgba(self, r8, g8, b8, a8):
        self._fill = ("rgba", r8, g8, b8, a8)

    def set_stroke_rgba(self, r8, g8, b8, a8):
        self._stroke = ("rgba", r8, g8, b8, a8)

    def set_fill_pattern(self, bitmap, repeat):
        pat = self.lib.cairo_pattern_create_for_surface(bitmap)
        self._keep.append(("p", pat))
        self._fill = ("pattern", pat, 1 if repeat else 0)

    def set_fill_radial(self, x0, y0, r0, x1, y1, r1, stops):
        pat = self.lib.cairo_pattern_create_radial(x0, y0, r0, x1, y1, r1)
        for t, r8, g8, b8, a8 in stops:
            self.lib.cairo_pattern_add_color_stop_rgba(pat, t, r8 / 255.0, g8 / 255.0, b8 / 255.0, a8 / 255.0)
        self._keep.append(("p", pat))
        self._fill = ("gradient", pat)

    def set_fill_linear(self, x0, y0, x1, y1, stops):
        pat = self.lib.cairo_pattern_create_linear(x0, y0, x1, y1)
        for t, r8, g8, b8, a8 in stops:
            self.lib.cairo_pattern_add_color_stop_rgba(pat, t, r8 / 255.0, g8 / 255.0, b8 / 255.0, a8 / 255.0)
        self._keep.append(("p", pat))
        self._fill = ("gradient", pat)

    def _apply_source(self, src):
        lib, cr = self.lib, self.cr
        if src[0] == "rgba":
            _, r8, g8, b8, a8 = src
            lib.cairo_set_source_rgba(cr, r8 / 255.0, g8 / 255.0, b8 / 255.0, a8 / 255.0)
        elif src[0] == "pattern":
            lib.cairo_set_source(cr, src[1])
            lib.cairo_pattern_set_extend(src[1], src[2])   # REPEAT=1 | NONE=0
            lib.cairo_pattern_set_filter(src[1], 1)        # CAIRO_FILTER_GOOD
        else:
            lib.cairo_set_source(cr, src[1])

    def set_fill_rule(self, even_odd: bool):
        self.lib.cairo_set_fill_rule(self.cr, 1 if even_odd else 0)

    def fill(self):
        self._apply_source(self._fill)
        self.lib.cairo_fill_preserve(self.cr)

    # -- stroke
    def set_line_width(self, w):
        if w > 0:  # node-canvas ignores non-positive widths
            self.lib.cairo_set_line_width(self.cr, w)

    def set_line_cap_round(self):
        self.lib.cairo_set_line_cap(self.cr, 1)

    def set_line_join_round(self):
        self.lib.cairo_set_line_join(self.cr, 1)

    def set_line_cap(self, k):
        """0 butt, 1 round, 2 square (cairo_line_cap_t)."""
        self.lib.cairo_set_line_cap(self.cr, int(k))

    def set_line_join(self, k):
        """0 miter, 1 round, 2 bevel (cairo_line_join_t)."""
        self.lib.cairo_set_line_join(self.cr, int(k))

    def stroke(self):
        self._apply_source(self._stroke)
        self.lib.cairo_stroke_preserve(self.cr)

    # -- read back
    def premultiplied_rgba(self) -> np.ndarray:
        """HxWx4 uint8, R,G,B,A byte order, premultiplied."""
        self.lib.cairo_surface_flush(self.surf)
        stride = self.lib.cairo_image_surface_get_stride(self.surf)
        ptr = self.lib.cairo_image_surface_get_data(self.surf)
        buf = np.ctypeslib.as_array(ptr, shape=(self.h, stride))[:, : self.w * 4].reshape(self.h, self.w, 4)
        return np.ascontiguousarray(buf[..., [2, 1, 0, 3]])  # B,G,R,A -> R,G,B,A
