"""TEST INFRASTRUCTURE ONLY -- Canvas2D backend over oracle/swfr_oracle.c (ctypes).

Same interface as cairo_backend.CairoBackend so canvas_replay.CanvasReplay can drive either.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libswfr_oracle.so")
    src = os.path.join(_HERE, "swfr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libswfr_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        c = ctypes
        P, D, I = c.c_void_p, c.c_double, c.c_int

        def sig(fn, res, *args):
            f = getattr(L, fn)
            f.restype = res
            f.argtypes = list(args)

        sig("swfo_create", P, I, I)
        sig("swfo_destroy", None, P)
        for fn in ("swfo_save", "swfo_restore", "swfo_identity_matrix", "swfo_clear_all", "swfo_new_path",
                   "swfo_close_path"):
            sig(fn, None, P)
        sig("swfo_transform", None, P, D, D, D, D, D, D)
        sig("swfo_scale", None, P, D, D)
        sig("swfo_move_to", None, P, D, D)
        sig("swfo_line_to", None, P, D, D)
        sig("swfo_curve_to", None, P, D, D, D, D, D, D)
        sig("swfo_quadratic_curve_to", None, P, D, D, D, D)
        sig("swfo_set_source_rgba", None, P, D, D, D, D)
        sig("swfo_set_source_gradient", None, P, I, D, D, D, D, D, D, I, c.POINTER(D), c.POINTER(D))
        sig("swfo_set_source_surface", None, P, P, I, I, I)
        sig("swfo_set_line_width", None, P, D)
        sig("swfo_set_line_cap", None, P, I)
        sig("swfo_set_line_join", None, P, I)
        sig("swfo_set_fill_rule", None, P, I)
        sig("swfo_fill_preserve", I, P)
        sig("swfo_stroke_preserve", I, P)
        sig("swfo_pixels", c.POINTER(c.c_uint32), P)
        sig("swfo_is_clear", I, P)
        sig("swfo_fill_polygons_fixed", None, P, P, P, P, I, I)
        _LIB = L
    return _LIB


class OracleBackend:
    def __init__(self, width, height):
        self.L = lib()
        self.w, self.h = width, height
        self.ctx = self.L.swfo_create(width, height)
        self.L.swfo_set_line_width(self.ctx, 1.0)     # node-canvas Context2d ctor: cairo_set_line_width(1)
        self._keep = []
        self._fill = ("rgba", 0, 0, 0, 255)
        self._stroke = ("rgba", 0, 0, 0, 255)
        self._stack = []
        self.unsupported = 0

    def close(self):
        if self.ctx:
            self.L.swfo_destroy(self.ctx)
            self.ctx = None

    def set_transform_identity(self):
        self.L.swfo_identity_matrix(self.ctx)

    def clear_all(self):
        self.L.swfo_clear_all(self.ctx)

    def scale(self, sx, sy):
        self.L.swfo_scale(self.ctx, sx, sy)

    def transform(self, a, b, c, d, e, f):
        self.L.swfo_transform(self.ctx, a, b, c, d, e, f)

    def save(self):
        self.L.swfo_save(self.ctx)
        self._stack.append((self._fill, self._stroke))

    def restore(self):
        self.L.swfo_restore(self.ctx)
        self._fill, self._stroke = self._stack.pop()

    def begin_path(self):
        self.L.swfo_new_path(self.ctx)

    def move_to(self, x, y):
        self.L.swfo_move_to(self.ctx, x, y)

    def line_to(self, x, y):
        self.L.swfo_line_to(self.ctx, x, y)

    def close_path(self):
        self.L.swfo_close_path(self.ctx)

    def quadratic_curve_to(self, x1, y1, x2, y2):
        self.L.swfo_quadratic_curve_to(self.ctx, x1, y1, x2, y2)

    def create_bitmap(self, w, h, rgba_straight: bytes):
        src = np.frombuffer(rgba_straight, dtype=np.uint8).reshape(h, w, 4).astype(np.uint32)
        a = src[..., 3]
        pm = lambda ch: (src[..., ch] * a // 255)
        argb = np.ascontiguousarray(((a << 24) | (pm(0) << 16) | (pm(1) << 8) | pm(2)).astype(np.uint32))
        self._keep.append(argb)
        return (argb, w, h)

    def set_fill_rgba(self, r8, g8, b8, a8):
        self._fill = ("rgba", r8, g8, b8, a8)

    def set_stroke_rgba(self, r8, g8, b8, a8):
        self._stroke = ("rgba", r8, g8, b8, a8)

    def set_fill_pattern(self, bitmap, repeat):
        argb, w, h = bitmap
        self.L.swfo_set_source_surface(self.ctx, argb.ctypes.data, w, h, 1 if repeat else 0)
        self._fill = ("locked",)

    def _gradient(self, linear, x0, y0, r0, x1, y1, r1, stops):
        n = len(stops)
        offs = (ctypes.c_double * max(n, 1))(*[s[0] for s in stops])
        cols = (ctypes.c_double * max(4 * n, 1))(*[v / 255.0 for s in stops for v in s[1:5]])
        self.L.swfo_set_source_gradient(self.ctx, linear, x0, y0, r0, x1, y1, r1, n, offs, cols)
        self._fill = ("locked",)

    def set_fill_radial(self, x0, y0, r0, x1, y1, r1, stops):
        self._gradient(0, x0, y0, r0, x1, y1, r1, stops)

    def set_fill_linear(self, x0, y0, x1, y1, stops):
        self._gradient(1, x0, y0, 0, x1, y1, 0, stops)

    def _apply(self, src):
        if src[0] == "rgba":
            _, r8, g8, b8, a8 = src
            self.L.swfo_set_source_rgba(self.ctx, r8 / 255.0, g8 / 255.0, b8 / 255.0, a8 / 255.0)

    def set_fill_rule(self, even_odd: bool):
        self.L.swfo_set_fill_rule(self.ctx, 1 if even_odd else 0)

    def fill(self):
        self._apply(self._fill)
        self.unsupported |= self.L.swfo_fill_preserve(self.ctx)

    def set_line_width(self, w):
        if w > 0:
            self.L.swfo_set_line_width(self.ctx, w)

    def set_line_cap_round(self):
        self.L.swfo_set_line_cap(self.ctx, 1)

    def set_line_join_round(self):
        self.L.swfo_set_line_join(self.ctx, 1)

    def set_line_cap(self, k):
        self.L.swfo_set_line_cap(self.ctx, int(k))

    def set_line_join(self, k):
        self.L.swfo_set_line_join(self.ctx, int(k))

    def stroke(self):
        self._apply(self._stroke)
        self.unsupported |= self.L.swfo_stroke_preserve(self.ctx)

    def premultiplied_rgba(self) -> np.ndarray:
        ptr = self.L.swfo_pixels(self.ctx)
        px = np.ctypeslib.as_array(ptr, shape=(self.h, self.w)).copy()
        out = np.empty((self.h, self.w, 4), dtype=np.uint8)
        out[..., 0] = (px >> 16) & 255
        out[..., 1] = (px >> 8) & 255
        out[..., 2] = px & 255
        out[..., 3] = px >> 24
        return out
