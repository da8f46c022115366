"""Workgroup timeline of k2_bin / k2_rows / k2_tiles from the -DSWFR_TRACE build (build/trace/libswfr.so): every workgroup stamps
the 100 MHz wall clock at entry, after each dependent memory round trip (k2_tiles) and at exit.  One frame in flight, scene S1
(or S2 with argv[1] = s2).  Prints, per kernel: workgroups, span first-entry -> last-exit, workgroup duration percentiles, and for
k2_tiles the median time of each dependent step.      gpurun: bash tools/build_variant.sh trace -DSWFR_TRACE && python tools/trace_wg.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWFR_FRAMES_IN_FLIGHT"] = "1"
import numpy as np
from swf_renderer_amd import api
LIB = os.path.join(ROOT, "build", os.environ.get("TRACE_BUILD", "trace"), "libswfr.so")
api.library_path = lambda: LIB
import swf_renderer_amd as S
from swf_renderer_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "s1"
if which in ("s1", "s2"):
    cfg = synth.S1 if which == "s1" else synth.S2
    W, H = cfg["width"], cfg["height"]
    pts, cols = synth.scene(**cfg)
    stage = api.stars_to_stage(pts, cols)
else:                                                   # config3: the morph shape at ratio 1 (1080p); config2h: homestuck-beta-1 (1024x1024)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenarios
    from helpers import fixture
    if which == "config3":
        tag = fixture("homestuck-beta-29")
        b, mb = tag["bounds"], tag["morph_bounds"]
        x0, x1 = min(b["x_min"], mb["x_min"]), max(b["x_max"], mb["x_max"])
        y0, y1 = min(b["y_min"], mb["y_min"]), max(b["y_max"], mb["y_max"])
        W, H = 1920, 1080
        sx, sy = W * 20 / (x1 - x0), H * 20 / (y1 - y0)
        stage = {"children": [{"type": "morph-shape", "definition": tag, "ratio": 1.0, "matrix": scenarios._m(sx, sy, -x0 * sx, -y0 * sy)}]}
    else:
        W, H = 1024, 1024
        stage = scenarios.scenarios()["config2_homestuck-beta-1"]["stage"]
host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY)
scene = host.build_frame(stage); host.close()
share = int(sys.argv[2]) if len(sys.argv) > 2 else 1
r = S.Renderer(W, H, band_index=share // 2, band_count=share, contiguous_bands=True)
r.upload_edges(*scene)
r.render_resident(10)
r.render_resident(1)
tm = r.timing()
L = ctypes.CDLL(LIB)
buf = np.zeros((3, 32768, 8), dtype=np.uint32)
L.swfr_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
n = L.swfr_debug_trace(buf.ctypes.data, buf.nbytes)
assert n > 0
print("events: k2_bin %.1f us, k2_rows %.1f us, k2_tiles %.1f us" % (tm["setup_ms"] * 1e3 / max(tm["timed_frames"], 1), tm["rows_ms"] * 1e3 / max(tm["timed_frames"], 1), tm["tiles_ms"] * 1e3 / max(tm["timed_frames"], 1)))
t_all0 = None
for k, name in enumerate(("k2_bin", "k2_rows", "k2_tiles")):
    b = buf[k]
    used = b[:, 7] != 0
    b = b[used].astype(np.int64)
    if not len(b):
        continue
    t0, t1 = b[:, 0].min(), b[:, 7].max()
    if t_all0 is None: t_all0 = t0
    dur = (b[:, 7] - b[:, 0]) * 0.01
    print("%s: %d workgroups, starts at %.2f us, span %.2f us; workgroup duration us: p10 %.2f p50 %.2f p90 %.2f max %.2f; entries spread over %.2f us" %
          (name, len(b), (t0 - t_all0) * 0.01, (t1 - t0) * 0.01, *np.percentile(dur, [10, 50, 90]), dur.max(), (b[:, 0].max() - t0) * 0.01))
    if name == "k2_bin":
        # which workgroups are the long ones: tile-row workgroups first, then the edge / path workgroups, the last eight order the strips
        used_idx = np.nonzero(used)[0]
        worst = np.argsort(-dur)[:10]
        print("   slowest workgroups (blockIdx: us): " + ", ".join("%d: %.1f" % (int(used_idx[i]), dur[i]) for i in worst) + "; the last eight blockIdx are the ordering workgroups (grid %d)" % (int(used_idx.max()) + 1))
        # entry time and duration along blockIdx (tile-rows' workgroups, edge workgroups, path workgroups, the ordering eight)
        ent = (b[:, 0] - t0) * 0.01
        g = int(used_idx.max()) + 1
        cuts = sorted(set(list(range(0, g - 8, max((g - 8) // 12, 1))) + [g - 8, g]))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            m = (used_idx >= lo) & (used_idx < hi)
            if m.any(): print("   blockIdx %4d..%4d: entry %.2f..%.2f us, duration mean %.2f max %.2f us" % (lo, hi - 1, ent[m].min(), ent[m].max(), dur[m].mean(), dur[m].max()))
    if name == "k2_tiles":
        names = ["descriptor fields", "strip descriptor", "class bytes", "entries + row headers", "first cells", "blend (to the store)", "stores acknowledged"]
        prev = b[:, 0].copy()
        for i in range(1, 8):
            cur = b[:, i]
            ok = cur != 0
            d = (cur[ok] - prev[ok]) * 0.01
            if len(d): print("   %-24s p50 %.2f  p90 %.2f us  (%d workgroups)" % (names[i - 1], *np.percentile(d, [50, 90]), len(d)))
            prev = np.where(ok, cur, prev)
        # concurrency: workgroups alive over time
        ev = np.concatenate([np.stack([b[:, 0], np.ones(len(b), np.int64)], 1), np.stack([b[:, 7], -np.ones(len(b), np.int64)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        alive = np.cumsum(ev[:, 1])
        print("   workgroups alive: max %d, time-weighted mean %.0f" % (alive.max(), float((alive[:-1] * np.diff(ev[:, 0])).sum() / max(ev[-1, 0] - ev[0, 0], 1))))
    if name == "k2_rows":
        names = ["chunk, path, band records", "edges staged, row masks", "gather + evaluate", "sort, roles, FULL cells, masks", "sample passes", "", "headers, queue, classification"]
        prev = b[:, 0].copy()
        for i in range(1, 8):
            cur = b[:, i]
            ok = cur != 0
            d = (cur[ok] - prev[ok]) * 0.01
            if len(d) and names[i - 1]: print("   %-32s p50 %.2f  p90 %.2f  max %.2f us  (%d wavefronts)" % (names[i - 1], *np.percentile(d, [50, 90]), d.max(), len(d)))
            prev = np.where(ok, cur, prev)
        # which wavefronts are the slow ones: duration against the chunk's path (width, rows of the path in the chunk)
        raw = buf[k].astype(np.int64)
        edges_a, paths_a, _ = scene
        cr = int(os.environ.get("SWFR_CHUNK_ROWS", "64"))
        rows_, widths_, nedge_ = [], [], []
        for pth in paths_a:
            if pth["y_max"] <= pth["y_min"]: continue
            a0 = int(pth["y_min"]) // 16 * 16
            for c0 in range(a0, int(pth["y_max"]), cr):
                rows_.append(min(c0 + cr, int(pth["y_max"])) - max(c0, int(pth["y_min"])))
                widths_.append(int(pth["x_max"]) - int(pth["x_min"]))
                nedge_.append(int(pth["n_edges"]))
        nck = min(len(rows_), len(raw))
        dd = (raw[:nck, 7] - raw[:nck, 0]) * 0.01
        rows_, widths_ = np.array(rows_[:nck]), np.array(widths_[:nck])
        order = np.argsort(-dd)[:12]
        print("   slowest wavefronts (us, path width, live rows): " + ", ".join("%.1f/%d/%d" % (dd[i], widths_[i], rows_[i]) for i in order))
        for lo, hi in ((0, 64), (64, 128), (128, 256), (256, 384), (384, 1024)):
            m = (widths_ >= lo) & (widths_ < hi)
            if m.any(): print("   path width %4d..%4d: %5d wavefronts, duration p50 %.1f max %.1f us, live rows mean %.0f" % (lo, hi, m.sum(), np.percentile(dd[m], 50), dd[m].max(), rows_[m].mean()))
        d = (b[:, 1] - b[:, 0]) * 0.01
        print("   descriptor + chunk count p50 %.2f p90 %.2f us" % tuple(np.percentile(d, [50, 90])))
r.close()
