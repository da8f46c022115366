"""Reproduces one scene of tools/soak.py:  python tools/soak_case.py <generator> <seed> <index> [gpu]
CPU: the host frame builder's edge list against the oracle's polygons; with `gpu`: differing pixels HIP vs oracle, per prefix of
the display list and with the k_tiles / k_rows route knobs."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import soak
from helpers import oracle_render


def scene(name, seed, idx):
    from helpers import soak_scene
    return soak_scene(name, seed, idx)


def host_vs_oracle(sc):
    import swf_renderer_amd as S
    from swf_renderer_amd import api
    from oracle import canvas_replay as cr
    from test_host import _Tap
    r = S.Renderer(sc["width"], sc["height"], device=api.DEVICE_HOST_ONLY, even_odd=bool(sc.get("even_odd")))
    for b in sc.get("bitmaps", []):
        r.add_bitmap(b)
    edges, paths, styles = r.build_frame(sc["stage"])
    tap = _Tap(sc["width"], sc["height"])
    if sc.get("even_odd"):
        tap.set_fill_rule(True)
    rp = cr.CanvasReplay(tap, linear_extension=True)
    for b in sc.get("bitmaps", []):
        rp.add_bitmap(b)
    rp.render(sc["stage"])
    polys = [(pe, rect) for pe, rect in tap.polys]
    print("host paths", len(paths), "oracle polygons", len(polys))
    k = 0
    for pth in paths:
        e = edges[pth["first_edge"]: pth["first_edge"] + pth["n_edges"]]
        if pth["kind"] != api.PATH_TOR:
            continue
        g = np.stack([e[f] for f in ("x1", "y1", "x2", "y2", "top", "bottom", "dir")], 1)
        ok = any(g.shape == pe.shape and (g == pe).all() for pe, rect in polys if not rect)
        print(" tor path", k, "edges", len(g), "rect", pth["x_min"], pth["y_min"], pth["x_max"], pth["y_max"], "lerp", pth["lerp"], "matches an oracle polygon:", ok)
        k += 1
    tap.close(); r.close()


def main():
    name, seed, idx = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    sc = scene(name, seed, idx)
    print("scene", name, seed, idx, "frame", sc["width"], sc["height"], "even_odd", sc.get("even_odd"), "children", len(sc["stage"]["children"]))
    host_vs_oracle(sc)
    if len(sys.argv) > 4 and sys.argv[4] == "gpu":
        from helpers import product_render
        kids = sc["stage"]["children"]
        for k in range(1, len(kids) + 1):
            sub = dict(sc); sub["stage"] = {"children": kids[:k]}
            ref = np.asarray(oracle_render(sub)).astype(int); got = np.asarray(product_render(sub)).astype(int)
            d = np.abs(ref - got).max(-1)
            print(" first", k, "children:", int((d > 0).sum()), "differing px", kids[k - 1]["type"])
            if (d > 0).sum():
                ys, xs = np.nonzero(d)
                for y, x in list(zip(ys, xs))[:12]:
                    print("   px", x, y, "hip", got[y, x], "oracle", ref[y, x])
                for env in ({"SWFR_FAST_LIMIT": "0"}, {"SWFR_CHUNK_ROWS": "64"}, {"SWFR_CHUNK_ROWS": "16"}, {"SWFR_STRIP_ORDER": "0"}):
                    os.environ.update(env)
                    g2 = np.asarray(product_render(sub)).astype(int)
                    for kk in env: del os.environ[kk]
                    print("   with", env, ":", int((np.abs(ref - g2).max(-1) > 0).sum()), "differing px")
                break


if __name__ == "__main__":
    main()
