"""Multi-GPU frame assembly: tile-row bands sharded over ranks, one gather to the root.

The frame is cut into tile-rows of TILE_H pixel rows; rank k of N rasterizes the tile-rows t with
t % N == k (interleaving balances edge density).  Every rank holds the whole (small) edge list, so the
data path has exactly one exchange step: each rank's packed band slab goes to the root
(`torch.distributed.gather`, RCCL over xGMI on GPUs, gloo on CPU tensors for tests) where the bands are
de-interleaved into the final image.  There is no collective inside rasterization.
"""
from __future__ import annotations

import numpy as np

TILE_H = 16


def local_tile_rows(height: int, rank: int, world: int) -> int:
    tile_rows = (height + TILE_H - 1) // TILE_H
    if world <= 1:
        return tile_rows
    return 0 if tile_rows <= rank else (tile_rows - rank + world - 1) // world


def slab_shape(width: int, height: int, rank: int, world: int):
    return (local_tile_rows(height, rank, world) * TILE_H, width, 4)


def max_slab_rows(height: int, world: int) -> int:
    return local_tile_rows(height, 0, world) * TILE_H


def extract_slab(image: np.ndarray, rank: int, world: int) -> np.ndarray:
    """The band slab rank `rank` would produce from a full image (rows past the frame are zero)."""
    h, w = image.shape[:2]
    n = local_tile_rows(h, rank, world)
    out = np.zeros((n * TILE_H, w, 4), dtype=image.dtype)
    for lt in range(n):
        y0 = (lt * world + rank) * TILE_H
        rows = min(TILE_H, h - y0)
        out[lt * TILE_H: lt * TILE_H + rows] = image[y0:y0 + rows]
    return out


def assemble(slabs, width: int, height: int):
    """De-interleave per-rank slabs (numpy or torch, [rows, width, 4]) into the final HxWx4 image.

    With equally padded slabs this is one permute + copy: [world, n, TILE_H, W, 4] -> [n, world, TILE_H, W, 4].
    """
    world = len(slabs)
    n = max_slab_rows(height, world) // TILE_H
    is_torch = hasattr(slabs[0], "new_zeros")
    padded = []
    for slab in slabs:
        if slab.shape[0] != n * TILE_H:
            full = slab.new_zeros((n * TILE_H, width, 4)) if is_torch else np.zeros((n * TILE_H, width, 4), dtype=slab.dtype)
            full[: slab.shape[0]] = slab
            slab = full
        padded.append(slab)
    if is_torch:
        import torch
        stacked = torch.stack(padded).view(world, n, TILE_H, width, 4)
        return stacked.permute(1, 0, 2, 3, 4).reshape(n * world * TILE_H, width, 4)[:height].contiguous()
    stacked = np.stack(padded).reshape(world, n, TILE_H, width, 4)
    return np.ascontiguousarray(stacked.transpose(1, 0, 2, 3, 4).reshape(n * world * TILE_H, width, 4)[:height])


def gather_slabs(slab, width: int, height: int, rank: int, world: int, dst: int = 0):
    """One gather of the padded band slabs to `dst`; returns the assembled image there, None elsewhere.

    `slab` is a torch tensor [rows, width, 4] uint8 on the device of the process group's backend.
    Slabs are padded to the largest rank's row count so the collective is uniform.
    """
    import torch
    import torch.distributed as dist
    rows = max_slab_rows(height, world)
    padded = slab
    if slab.shape[0] != rows:
        padded = slab.new_zeros((rows, width, 4))
        padded[: slab.shape[0]] = slab
    padded = padded.contiguous()
    if world == 1:
        return assemble([padded], width, height)
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, gather_list=bufs, dst=dst)
    if rank != dst:
        return None
    return assemble(bufs, width, height)


class FramePipeline:
    """N>1 step = render this rank's bands, pack them, gather to rank 0, assemble -- with the gather and the
    assembly of frame k overlapping the rasterization of frame k+1.

    The renderer works on its own HIP stream; the collective and the de-interleave run on torch's streams.
    `depth` slabs rotate; a slab is reused only after the event recorded behind its assembly has completed.
    """

    def __init__(self, renderer, width, height, rank, world, device="cuda", depth=3, dst=0):
        import torch
        self.r, self.w, self.h, self.rank, self.world, self.dst = renderer, width, height, rank, world, dst
        self.device = device
        rows = max_slab_rows(height, world)
        on_gpu = device != "cpu"
        self.stage = [torch.zeros((rows, width, 4), dtype=torch.uint8, device="cuda") for _ in range(depth)]
        self.cpu_slabs = None if on_gpu else [torch.zeros((rows, width, 4), dtype=torch.uint8) for _ in range(depth)]
        self.bufs = None
        if rank == dst:
            self.bufs = [[torch.empty((rows, width, 4), dtype=torch.uint8, device=device) for _ in range(world)] for _ in range(depth)]
        self.events = [None] * depth
        self.works = [None] * depth
        self.k = 0
        self.last = None

    def step(self):
        import torch
        import torch.distributed as dist
        i = self.k % len(self.stage)
        self.k += 1
        if self.works[i] is not None:
            self.works[i].wait()
            self.works[i] = None
        if self.events[i] is not None:
            self.events[i].synchronize()                   # the slab's previous gather + assembly are done
        self.r.render_resident(1)                          # blocking on the renderer's stream
        self.r.copy_band_slab(self.stage[i].data_ptr())    # packed tile-rows, zero padded to the common size
        send = self.stage[i]
        if self.cpu_slabs is not None:                     # gloo rehearsal: collectives on CPU tensors
            self.cpu_slabs[i].copy_(send)
            send = self.cpu_slabs[i]
        work = dist.gather(send, gather_list=self.bufs[i] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        if self.cpu_slabs is not None:
            work.wait()
        else:
            work.wait()                                    # NCCL: makes the current stream wait, not the host
        if self.rank == self.dst:
            self.last = assemble(self.bufs[i], self.w, self.h)
        if self.cpu_slabs is None:
            ev = torch.cuda.Event()
            ev.record()
            self.events[i] = ev
        return self.last

    def finish(self):
        import torch
        for w in self.works:
            if w is not None:
                w.wait()
        torch.cuda.synchronize()
        return self.last
