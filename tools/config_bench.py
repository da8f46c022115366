"""Throughput of BASELINE.json's configs 2-4 on one MI355X (the parity of the same configurations is in tests/test_gpu_parity.py).

For every configuration two figures: the full path as the reference's API is used (register once, then per frame: Stage -> host edge
list -> H2D -> kernels -> sync, i.e. `Renderer.render`), and the device path alone on the uploaded edge list (`render_resident`).
usage (GPU box): python tools/config_bench.py
"""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
if os.environ.get("SWFR_LIB"):                          # another build of the library (tuning experiments)
    from swf_renderer_amd import api as _api
    _lib = os.path.abspath(os.environ["SWFR_LIB"])
    _api.library_path = lambda: _lib
import swf_renderer_amd as S
import scenarios
from helpers import fixture


ONLY = sys.argv[1] if len(sys.argv) > 1 else ""


def run(name, W, H, stages, bitmaps=(), resident_frames=200):
    if ONLY and ONLY not in name:
        return None
    r = S.Renderer(W, H)
    for bm in bitmaps:
        r.add_bitmap(bm)
    for st in stages[:2]:
        r.render(st)                                    # warm-up (definitions are registered and cached here)
    t0 = time.perf_counter()
    for st in stages:
        r.render(st)
    dt_full = (time.perf_counter() - t0) / len(stages)
    # the same frames as one pipelined batch into a device tensor (swfr_render_batch)
    import torch
    out_t = torch.empty((len(stages), H, W, 4), dtype=torch.uint8, device="cuda")
    r.render_batch(stages, out_t.data_ptr(), H * W * 4)        # warm-up with the full batch (pinned buffers, lazily loaded runtime kernels)
    t0 = time.perf_counter()
    r.render_batch(stages, out_t.data_ptr(), H * W * 4)
    dt_batch = (time.perf_counter() - t0) / len(stages)
    tb = r.timing()
    dt_batch_dev = tb["total_ms"] * 1e-3 / max(tb["frames"], 1)     # the groups' kernels only (HIP events)
    del out_t
    # device path alone: the last stage's edge list, resident
    edges, paths, styles = r.build_frame(stages[-1])
    r.upload_edges(edges, paths, styles)
    r.render_resident(20)
    r.render_resident(resident_frames)
    tm = r.timing()
    dt_dev = tm["total_ms"] * 1e-3 / tm["frames"]
    r.close()
    out = {"config": name, "frame": "%dx%d" % (W, H), "frames": len(stages), "edges": int(len(edges)), "paths": int(len(paths)),
           "full_path_frames_per_s": round(1 / dt_full, 1), "full_path_Mpx_per_s": round(W * H / dt_full / 1e6, 1),
           "batch_frames_per_s": round(1 / dt_batch, 1), "batch_Mpx_per_s": round(W * H / dt_batch / 1e6, 1),
           "batch_device_frames_per_s": round(1 / dt_batch_dev, 1) if dt_batch_dev > 0 else None,
           "device_frames_per_s": round(1 / dt_dev, 1), "device_Mpx_per_s": round(W * H / dt_dev / 1e6, 1),
           "kernel_us": {k: round(tm[k + "_ms"] * 1e3 / max(tm["timed_frames"], 1), 1) for k in ("setup", "rows", "tiles")}}
    print(json.dumps(out), flush=True)
    return out


def main():
    import torch
    assert torch.cuda.is_available()                 # torch initialises HIP first (it does not find the GPU after another HIP user did)
    m = scenarios._m
    # config 2: the flat shapes (+ gradient variants) at 1024x1024
    SC = scenarios.scenarios()
    for nm in ("config2_squares", "config2_triangle", "config2_homestuck-beta-1"):
        run(nm, 1024, 1024, [SC[nm]["stage"]] * 50)
    circ = scenarios._circleish(10240, 10240, 9000, 64)
    gm = m(9000 / 16384, 9000 / 16384, 10240, 10240)
    rad = {"type": "radial-gradient", "matrix": gm, "gradient": scenarios._grad([(0, (255, 0, 0)), (128, (0, 255, 0)), (255, (0, 0, 255))])}
    run("config2_radial_gradient_disc", 1024, 1024, [{"children": [{"type": "shape", "definition": scenarios._poly_shape(circ, rad)}]}] * 50)
    # config 3: 256 morph ratios at 1080p
    tag = fixture("homestuck-beta-29")
    b, mb = tag["bounds"], tag["morph_bounds"]
    x0, x1 = min(b["x_min"], mb["x_min"]), max(b["x_max"], mb["x_max"])
    y0, y1 = min(b["y_min"], mb["y_min"]), max(b["y_max"], mb["y_max"])
    sx, sy = 1920 * 20 / (x1 - x0), 1080 * 20 / (y1 - y0)
    stages = [{"children": [{"type": "morph-shape", "definition": tag, "ratio": k / 255, "matrix": m(sx, sy, -x0 * sx, -y0 * sy)}]} for k in range(256)]
    run("config3_morph_256_ratios", 1920, 1080, stages)
    # config 4: textured shape at 4K (139x208 texture magnified) and a 4096x4096 synthetic texture sampled ~1:1
    tag4 = fixture("homestuck-beta-4")
    b = tag4["bounds"]
    sx, sy = 3840 * 20 / (b["x_max"] - b["x_min"]), 2160 * 20 / (b["y_max"] - b["y_min"])
    st4 = {"children": [{"type": "shape", "definition": tag4, "matrix": m(sx, sy, -b["x_min"] * sx, -b["y_min"] * sy)}]}
    run("config4_textured_4k_magnified", 3840, 2160, [st4] * 20, bitmaps=[fixture("homestuck-beta-3.bitmap")])
    # ... and the HBM-bound variant: a 4096 x 4096 texture (64 MB) sampled about 1:1
    from helpers import large_texture_scene
    big = large_texture_scene()
    out = run("config4_textured_4k_large_texture", 3840, 2160, [big["stage"]] * 10, bitmaps=big["bitmaps"])
    if out:
        texels = 4096 * int(2160 * 4096 / 3840) * 4                       # bytes of the visible part of the texture
        us = out["kernel_us"]["tiles"]
        print(json.dumps({"config": "config4_textured_4k_large_texture", "k2_tiles_us": us, "algorithmic_bytes": 4 * 3840 * 2160 + texels,
                          "achieved_GBps": round((4 * 3840 * 2160 + texels) / us / 1e3, 1), "frac_of_8TBps": round((4 * 3840 * 2160 + texels) / us / 1e3 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
