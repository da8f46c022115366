"""One rank of the rotating assembly on CPU (RotatingPipeline): the kernels under the wavefront emulator, the all-to-all over gloo.
Frame f is assembled on rank f mod N; every rank compares every frame it assembled with the oracle.  Launched by
tests/test_distributed.py with RANK / WORLD_SIZE / MASTER_* set."""
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
for name in sys.argv[1:] or ["stroke_curves"]:
    sc = SC[name]
    w, h = sc["width"], sc["height"]
    r = S.Renderer(w, h, band_index=rank, band_count=world, contiguous_bands=True)
    pipe = D.RotatingPipeline(r, w, h, rank, world, device="cpu", frames_device="cpu")
    edges, paths, styles = r.build_frame(sc["stage"])
    pipe.upload(edges, paths, styles)
    ref = oracle_render(sc)
    for group in range(3):                      # more groups than buffers: both group buffers are used, the first one twice
        out = pipe.step_group()
        d = diff_stats(out.numpy(), ref)
        print("rank", rank, name, "frame", pipe.assembled[-1][0], d, flush=True)
        bad += d != (0, 0)
    out = pipe.step_group(1)                    # a last, partial group: only the frame of rank 0 is rendered
    if rank == 0:
        d = diff_stats(out.numpy(), ref)
        print("rank", rank, name, "frame", pipe.assembled[-1][0], d, "(partial group)", flush=True)
        bad += d != (0, 0)
    else:
        bad += out is not None
    pipe.finish()
    r.close()
flag = torch.tensor([bad])
dist.all_reduce(flag)
dist.barrier(); dist.destroy_process_group()
print("ROTATE_FAILED" if int(flag) else "ROTATE_OK")
sys.exit(1 if int(flag) else 0)
