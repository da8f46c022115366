"""One rank of the N>1 path on CPU: the kernels under the wavefront emulator, the gather over gloo.

Each rank creates a handle that owns one contiguous block of tile-rows, renders it through FramePipeline into host tensors
(the emulator's "device" memory is host memory) and gathers in place to rank 0, which compares the assembled frames with the
oracle.  Launched by tests/test_distributed.py with RANK / WORLD_SIZE / MASTER_* set.
"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
import run as emurun
emurun.use_emulator()
import numpy as np
import torch
import torch.distributed as dist
import swf_renderer_amd as S
from swf_renderer_amd import distributed as D
from helpers import oracle_render, diff_stats
import scenarios

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
SC = scenarios.scenarios()
bad = 0
for name in sys.argv[1:] or ["stroke_curves", "translucent_stack"]:
    sc = SC[name]
    w, h = sc["width"], sc["height"]
    r = S.Renderer(w, h, band_index=rank, band_count=world, contiguous_bands=True)
    pipe = D.FramePipeline(r, w, h, rank, world, device="cpu", frames_device="cpu")
    edges, paths, styles = r.build_frame(sc["stage"])
    pipe.upload(edges, paths, styles)
    for step in range(4):                       # more steps than buffers: every frame set is used, the first one twice
        out = pipe.step()
        if rank == 0:
            d = diff_stats(out.numpy(), oracle_render(sc))
            print(name, step, d)
            bad += d != (0, 0)
    pipe.finish()
    r.close()
flag = torch.tensor([bad])
dist.broadcast(flag, src=0)
dist.barrier(); dist.destroy_process_group()
print("DIST_FAILED" if int(flag) else "DIST_OK")
sys.exit(1 if int(flag) else 0)
