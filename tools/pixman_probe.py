"""Container-only probe: feeds the pixman parameters the oracle derived for a radial-gradient fill (transform, circles, stops)
straight into libpixman-1 (ctypes) and compares pixman's pixels with the oracle's own evaluation -- separates "Cairo hands pixman
different parameters" from "pixman evaluates them differently".   usage: python tools/pixman_probe.py <generator> <seed> <index> <child>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from helpers import soak_scene
from oracle import oracle_backend as ob, canvas_replay as cr
import soak

PX = C.CDLL("libpixman-1.so.0")
class Pt(C.Structure): _fields_ = [("x", C.c_int32), ("y", C.c_int32)]
class Color(C.Structure): _fields_ = [("red", C.c_uint16), ("green", C.c_uint16), ("blue", C.c_uint16), ("alpha", C.c_uint16)]
class Stop(C.Structure): _fields_ = [("x", C.c_int32), ("color", Color)]
class Transform(C.Structure): _fields_ = [("m", (C.c_int32 * 3) * 3)]
PX.pixman_image_create_radial_gradient.restype = C.c_void_p
PX.pixman_image_create_radial_gradient.argtypes = [C.POINTER(Pt), C.POINTER(Pt), C.c_int32, C.c_int32, C.POINTER(Stop), C.c_int]
PX.pixman_image_create_bits.restype = C.c_void_p
PX.pixman_image_create_bits.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
PX.pixman_image_set_transform.argtypes = [C.c_void_p, C.POINTER(Transform)]
PX.pixman_image_set_repeat.argtypes = [C.c_void_p, C.c_int]
PX.pixman_image_composite32.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 8
PIXMAN_a8r8g8b8, OP_SRC, REPEAT_PAD = 0x20028888, 1, 2

def main():
    name, seed, idx, child = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sc = soak_scene(name, seed, idx)
    kid = sc["stage"]["children"][child]
    W, H = sc["width"], sc["height"]
    be = ob.OracleBackend(W, H)
    cr.CanvasReplay(be, linear_extension=True).render({"children": [kid]})
    img = be.premultiplied_rgba().astype(int)
    out = (C.c_int64 * 16)(); sx = (C.c_int64 * 20)(); ramp = (C.c_float * 160)()
    be.L.swfo_debug_radial.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert be.L.swfo_debug_radial(be.ctx, out, sx, ramp)
    pm = list(out[:6]); pox, poy = out[6], out[7]
    c1x, c1y, c1r, dx, dy, dr, n_int = out[8:15]
    f = kid["definition"]["shape"]["initial_styles"]["fill"][0]
    cols = sorted(f["gradient"]["colors"], key=lambda s: s["ratio"])
    n = len(cols)
    stops = (Stop * n)()
    for i, s in enumerate(cols):
        c = s["color"]
        stops[i].x = int(np.rint(s["ratio"] / 255.0 * 65536.0))
        stops[i].color = Color(int(c["r"] / 255.0 * 65535.0 + 0.5), int(c["g"] / 255.0 * 65535.0 + 0.5), int(c["b"] / 255.0 * 65535.0 + 0.5), int(c["a"] / 255.0 * 65535.0 + 0.5))
    assert [stops[i].x for i in range(n)] == [sx[i + 1] for i in range(n)], "stop offsets differ"
    p1, p2 = Pt(c1x, c1y), Pt(c1x + dx, c1y + dy)
    g = PX.pixman_image_create_radial_gradient(C.byref(p1), C.byref(p2), c1r, c1r + dr, stops, n)
    t = Transform()
    t.m[0][0], t.m[0][1], t.m[0][2] = pm[0], pm[1], pm[2]
    t.m[1][0], t.m[1][1], t.m[1][2] = pm[3], pm[4], pm[5]
    t.m[2][0], t.m[2][1], t.m[2][2] = 0, 0, 65536
    ok_t = PX.pixman_image_set_transform(g, C.byref(t)); PX.pixman_image_set_repeat(g, REPEAT_PAD)
    print('transform set:', ok_t, 'pm', pm, 'offset', pox, poy, 'circles', c1x, c1y, c1r, dx, dy, dr)
    buf = np.zeros((H, W), np.uint32)
    dst = PX.pixman_image_create_bits(PIXMAN_a8r8g8b8, W, H, buf.ctypes.data, W * 4)
    assert g and dst
    ys, xs = np.nonzero(img[..., 3] > 0)
    y0, y1, x0, x1 = int(ys.min()), int(ys.max()) + 1, int(xs.min()), int(xs.max()) + 1
    # only the shape's own rectangle: pixman gives up on a scanline whose first sample position overflows 16.16
    PX.pixman_image_composite32(OP_SRC, g, None, dst, x0 + int(pox), y0 + int(poy), 0, 0, x0, y0, x1 - x0, y1 - y0)
    be.L.swfo_debug_sample.restype = C.c_uint32
    be.L.swfo_debug_sample.argtypes = [C.c_void_p, C.c_int, C.c_int]
    cairo = np.asarray(soak.cairo_render(dict(sc, stage={"children": [kid]}))).astype(int)
    n_cmp = n_bad = 0
    for y in range(y0, y1):
        for x in range(x0, x1):
            o = be.L.swfo_debug_sample(be.ctx, x, y); p = int(buf[y, x]); n_cmp += 1
            if o != p:
                n_bad += 1
                if n_bad <= 8: print("  source differs at", x, y, "oracle %08x pixman %08x" % (o, p))
    print("source samples compared", n_cmp, "differing", n_bad, "(pixman non-zero:", int((buf != 0).sum()), ")")
    ys, xs = np.nonzero((np.abs(cairo - img).max(-1) > 0))
    for y, x in zip(ys, xs):
        print("px", x, y, "cairo", cairo[y, x], "oracle", img[y, x], "oracle source %08x pixman source %08x" % (be.L.swfo_debug_sample(be.ctx, int(x), int(y)), int(buf[y, x])))
    # which single change of the transform makes pixman reproduce cairo on the fully covered pixels?
    solid = (img[..., 3] > 0)
    full_cov = np.zeros_like(solid)
    for y in range(y0, y1):
        for x in range(x0, x1):
            o = be.L.swfo_debug_sample(be.ctx, x, y)
            ov = np.array([(o >> 16) & 255, (o >> 8) & 255, o & 255, o >> 24])
            full_cov[y, x] = (ov == img[y, x]).all() and o != 0
    base_bad = int(((np.abs(cairo - img).max(-1) > 0) & full_cov).sum())
    print("fully covered pixels", int(full_cov.sum()), "of which cairo != oracle:", base_bad)
    def pix_with(pmv, ox, oy):
        t2 = Transform()
        t2.m[0][0], t2.m[0][1], t2.m[0][2] = pmv[0], pmv[1], pmv[2]
        t2.m[1][0], t2.m[1][1], t2.m[1][2] = pmv[3], pmv[4], pmv[5]
        t2.m[2][0], t2.m[2][1], t2.m[2][2] = 0, 0, 65536
        PX.pixman_image_set_transform(g, C.byref(t2))
        b2 = np.zeros((H, W), np.uint32)
        d2 = PX.pixman_image_create_bits(PIXMAN_a8r8g8b8, W, H, b2.ctypes.data, W * 4)
        PX.pixman_image_composite32(OP_SRC, g, None, d2, x0 + int(ox), y0 + int(oy), 0, 0, x0, y0, x1 - x0, y1 - y0)
        return np.stack([(b2 >> 16) & 255, (b2 >> 8) & 255, b2 & 255, b2 >> 24], -1).astype(int)
    for k in range(6):
        for dlt in (-64, -16, -4, -2, -1, 1, 2, 4, 16, 64):
            pv = list(pm); pv[k] += dlt
            bad = int(((np.abs(pix_with(pv, pox, poy) - cairo).max(-1) > 0) & full_cov).sum())
            if bad < base_bad: print("  pm[%d] %+d -> %d differing from cairo" % (k, dlt, bad))
    be.close()

main()
