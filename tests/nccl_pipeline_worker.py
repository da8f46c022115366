"""One rank of FramePipeline over RCCL (backend "nccl"), started by test_gpu_parity.py: the same code path the N>1 bench runs --
render into a tensor on the handle's streams, torch's stream waits for them, in-place gather, the event that frees the buffer.
With WORLD_SIZE=1 the gather is the root's send-to-self, which is all a one-GPU box can carry; prints PIPELINE_OK."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import swf_renderer_amd as S
from swf_renderer_amd import api, synth, distributed as D

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
cfg = synth.S1
W, H = cfg["width"], cfg["height"]
pts, cols = synth.scene(**cfg)
host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY)
scene = host.build_frame(api.stars_to_stage(pts, cols))
host.close()
rb = S.Renderer(W, H, band_index=rank, band_count=world, contiguous_bands=True)
pipe = D.FramePipeline(rb, W, H, rank, world)
pipe.upload(*scene)
outs = []
for step in range(8):                               # every buffer is reused at least once: the event ordering is exercised
    out = pipe.step()
    if step in (0, 3, 7) and rank == 0:
        outs.append(out)                            # (views of the rotating images: read them only after finish())
out = pipe.finish()
ok = True
if rank == 0:
    digest = hashlib.sha256(np.ascontiguousarray(out.cpu().numpy()).tobytes()).hexdigest()
    ok = digest == synth.S1_SHA256_PREMUL
rb.close()
# the rotating assembly on the same backend: groups of `world` frames, block buffers, one all-to-all per group (world 1: a copy to self)
rb = S.Renderer(W, H, band_index=rank, band_count=world, contiguous_bands=True)
rot = D.RotatingPipeline(rb, W, H, rank, world)
rot.upload(*scene)
for step in range(5):                               # both group buffers are reused: the event ordering is exercised
    rot.step_group()
out2 = rot.finish()
if rank == 0:
    ok = ok and hashlib.sha256(np.ascontiguousarray(out2.cpu().numpy()).tobytes()).hexdigest() == synth.S1_SHA256_PREMUL
dist.barrier()
dist.destroy_process_group()
rb.close()
print("PIPELINE_OK" if ok else "PIPELINE_MISMATCH", flush=True)
sys.exit(0 if ok else 1)
