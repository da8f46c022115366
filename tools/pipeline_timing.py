"""FramePipeline step rate on one GPU over RCCL (world 1: the gather is a send-to-self of the whole frame), S1 and S2, and --
for the sharded case -- the device time of ONE rank's block when the frame is cut into 2/4/8 contiguous blocks (the other
blocks are other GPUs' work).  Feeds DESIGN.md's multi-GPU expectation table."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import numpy as np, torch, torch.distributed as dist
import swf_renderer_amd as S
from swf_renderer_amd import api, synth, distributed as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
res = {}
for name, cfg in (("s1", synth.S1), ("s2", synth.S2)):
    W, H = cfg["width"], cfg["height"]
    pts, cols = synth.scene(**cfg)
    host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY)
    scene = host.build_frame(api.stars_to_stage(pts, cols)); host.close()
    rb = S.Renderer(W, H)
    pipe = D.FramePipeline(rb, W, H, 0, 1)
    pipe.upload(*scene)
    for _ in range(10): pipe.step()
    pipe.finish()
    n = 200 if name == "s1" else 60
    t0 = time.perf_counter()
    for _ in range(n): pipe.step()
    pipe.finish()
    dt = time.perf_counter() - t0
    res[name] = {"pipeline_world1_ms_per_step": round(dt / n * 1e3, 4)}
    rb.close()
    for world in (1, 2, 4, 8):
        per = []
        for rank in sorted({0, world // 2, world - 1}):
            r = S.Renderer(W, H, band_index=rank, band_count=world, contiguous_bands=True)
            r.upload_edges(*scene)
            r.render_resident(20)
            r.render_resident(60)
            tm = r.timing()
            per.append({"rank": rank, "ms_per_frame": round(tm["total_ms"] / max(tm["timed_frames"], 1), 4),
                        "k2_bin": round(tm["setup_ms"] / max(tm["timed_frames"], 1), 4),
                        "k2_rows": round(tm["rows_ms"] / max(tm["timed_frames"], 1), 4), "k2_tiles": round(tm["tiles_ms"] / max(tm["timed_frames"], 1), 4)})
            r.close()
        res[name]["blocks_%d" % world] = per
dist.destroy_process_group()
print(json.dumps(res))
