"""Multi-GPU frame assembly: the frame's tile-rows sharded over the ranks, one gather to the root.

The frame is cut into tile-rows of TILE_H pixel rows.  Every rank holds the whole (small) edge list and rasterizes only its share,
so the data path has exactly one exchange step per frame: each rank's rows go to the root (`torch.distributed.gather`: RCCL over
xGMI on GPUs, gloo on CPU tensors for tests).  There is no collective inside rasterization.

Two ways to share the tile-rows out (swfr_config.band_index / band_count):
  * contiguous blocks (SWFR_FLAG_BANDS_CONTIGUOUS; what FramePipeline uses): rank k owns tile-rows [k*n, (k+1)*n), n = ceil(T / N).
    A rank's rows are one contiguous piece of the image, so it renders them straight into a tensor and the gather deposits them
    in the root's image in place -- no pack kernel on the sender, no de-interleave pass on the root.
  * interleaved (t % N == k): balances a scene whose edge density varies from top to bottom; needs `assemble` on the root.
"""
from __future__ import annotations

import numpy as np

TILE_H = 16


def tile_rows(height: int) -> int:
    return (height + TILE_H - 1) // TILE_H


def local_tile_rows(height: int, rank: int, world: int) -> int:
    """Interleaved split: tile-rows t with t % world == rank."""
    t = tile_rows(height)
    if world <= 1:
        return t
    return 0 if t <= rank else (t - rank + world - 1) // world


def block_rows(height: int, world: int) -> int:
    """Contiguous split: tile-rows per rank (the last ranks may own fewer, or none)."""
    return (tile_rows(height) + max(world, 1) - 1) // max(world, 1)


def padded_height(height: int, world: int) -> int:
    """Pixel rows of an image that holds `world` equally sized contiguous blocks (>= height)."""
    return block_rows(height, world) * TILE_H * max(world, 1)


def slab_shape(width: int, height: int, rank: int, world: int):
    return (local_tile_rows(height, rank, world) * TILE_H, width, 4)


def max_slab_rows(height: int, world: int) -> int:
    return local_tile_rows(height, 0, world) * TILE_H


def extract_slab(image: np.ndarray, rank: int, world: int) -> np.ndarray:
    """The interleaved band slab rank `rank` would produce from a full image (rows past the frame are zero)."""
    h, w = image.shape[:2]
    n = local_tile_rows(h, rank, world)
    out = np.zeros((n * TILE_H, w, 4), dtype=image.dtype)
    for lt in range(n):
        y0 = (lt * world + rank) * TILE_H
        rows = min(TILE_H, h - y0)
        out[lt * TILE_H: lt * TILE_H + rows] = image[y0:y0 + rows]
    return out


def assemble(slabs, width: int, height: int):
    """De-interleave per-rank slabs of the interleaved split (numpy or torch, [rows, width, 4]) into the final HxWx4 image.

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


def assemble_blocks(slabs, width: int, height: int):
    """Contiguous split: the padded per-rank slabs are simply stacked (what the in-place gather produces without any copy)."""
    is_torch = hasattr(slabs[0], "new_zeros")
    if is_torch:
        import torch
        return torch.cat(list(slabs))[:height].contiguous()
    return np.ascontiguousarray(np.concatenate(list(slabs))[:height])


def gather_slabs(slab, width: int, height: int, rank: int, world: int, dst: int = 0):
    """One gather of the padded interleaved band slabs to `dst`; returns the assembled image there, None elsewhere.

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
    """N>1 step = render this rank's block of tile-rows, gather the blocks to the root -- in place, and overlapped.

    Every rank's handle (contiguous bands) renders straight into torch tensors (`swfr_set_targets`): `depth` full-frame buffers
    that the handle's frame sets rotate over, each on its own HIP stream.  A step queues one frame (`swfr_render_resident_async`,
    no host wait), lets torch's stream wait for that frame set's stream, and starts the gather of the rank's rows -- a contiguous
    view of the buffer -- into the matching views of the root's padded image, where they land at their final place: no pack kernel,
    no de-interleave pass.  The gather of frame f runs while frame f + 1 is rasterized into the next buffer; a buffer is rendered
    into again only after the event behind its gather (the renderer's stream waits for it, not the host).
    """

    def __init__(self, renderer, width, height, rank, world, device="cuda", depth=None, dst=0, frames_device="cuda"):
        import os
        import torch
        if depth is None:                                            # one buffer per frame set of the handle (SWFR_FRAMES_IN_FLIGHT, default 4)
            depth = min(4, max(1, int(os.environ.get("SWFR_FRAMES_IN_FLIGHT", "4"))))
        self.r, self.w, self.h, self.rank, self.world, self.dst = renderer, width, height, rank, world, dst
        self.on_gpu = device != "cpu"                                # where the collective runs (NCCL/RCCL on device tensors, gloo on CPU ones)
        self.streams = frames_device != "cpu"                        # False only under tools/emu, whose "device" memory is host memory
        self.rows = block_rows(height, world) * TILE_H               # pixel rows per rank (padded: the last block may reach past the frame)
        self.hp = self.rows * world
        self.depth = depth
        self.frames = [torch.zeros((self.hp, width, 4), dtype=torch.uint8, device=frames_device) for _ in range(depth)]
        renderer.set_targets([f.data_ptr() for f in self.frames])
        self.image = None
        if rank == dst:
            self.image = [torch.zeros((self.hp, width, 4), dtype=torch.uint8, device=device) for _ in range(depth)]
        self.cpu_send = None if self.on_gpu else [torch.zeros((self.rows, width, 4), dtype=torch.uint8) for _ in range(depth)]
        self.ext = {}
        self.events = [None] * depth
        self.last = None
        self.uploaded = False
        self._n = 0

    def upload(self, edges, paths, styles):
        self.r.upload_edges(edges, paths, styles)                    # (after set_targets: the descriptors carry the buffers' addresses)
        self.r.render_resident(1)                                    # one blocking frame: the handle learns which queued-row kernels this scene needs
        self.uploaded = True

    def _stream(self, k):
        import torch
        if k not in self.ext:
            self.ext[k] = torch.cuda.ExternalStream(self.r.stream_handle(k))
        return self.ext[k]

    def step(self):
        import torch
        import torch.distributed as dist
        k = self._queue()
        i = k % self.depth
        if self.streams:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._stream(k))                         # torch's stream: behind this frame's kernels
        else:
            self.r.wait()
        y0 = self.rank * self.rows
        send = self.frames[i][y0:y0 + self.rows]                     # this rank's rows: one contiguous view
        out = None
        if self.rank == self.dst:
            out = self.image[i]
            views = [out[q * self.rows:(q + 1) * self.rows] for q in range(self.world)]
        else:
            views = None
        if not self.on_gpu:                                          # gloo rehearsal: the collective runs on CPU tensors
            if self.streams:
                self.cpu_send[i].copy_(send)                         # (blocking device-to-host copy on torch's stream)
                send = self.cpu_send[i]
            dist.gather(send, gather_list=views, dst=self.dst)
        else:
            dist.gather(send, gather_list=views, dst=self.dst)       # NCCL: queued on torch's stream, lands in place
            ev = torch.cuda.Event()
            ev.record(cur)
            self.events[i] = ev
        if self.rank == self.dst:
            self.last = out[: self.h]
        return self.last

    def _queue(self):
        # order the next frame set's stream behind the gather that last read its buffer, then queue the frame
        nxt = self._n
        self._n = nxt + 1
        i = nxt % self.depth
        if self.streams and self.events[i] is not None:
            self._stream(i).wait_event(self.events[i])
        k = self.r.render_resident_async()
        assert k % self.depth == i, "frame sets and buffers rotate together"
        return k

    def finish(self):
        import torch
        self.r.wait()
        if self.streams:
            torch.cuda.synchronize()
        return self.last


class RotatingPipeline:
    """N>1 step with an assembly that is NOT bound by one rank's inbound links: frame f is assembled on rank f mod N.

    FramePipeline lands every frame on one root, whose inbound xGMI links then carry the whole frame whatever the other ranks do
    (multi-GPU).  Here N consecutive frames form a group: every rank renders its block of tile-rows of each of them into
    `send[g][j]` -- a buffer of the block alone (`swfr_render_resident_group_to`: one call below Python queues the whole group) --
    and ONE all-to-all per group moves block (frame j, rows of rank q) from rank q to rank j, where it lands at its final place in
    that rank's image: every rank sends N - 1 blocks and receives N - 1 at once, all links of all ranks busy in both directions.
    The consumer of a frame (an encoder, a presenter) sits on the rank that assembled it.  Two groups are kept in flight: the
    exchange of group g overlaps the rasterization of group g + 1; a group's buffers are rendered into again only behind the event
    of the exchange that read them (on the device: no host wait in step()).
    """

    def __init__(self, renderer, width, height, rank, world, device="cuda", frames_device="cuda", groups=2):
        import torch
        self.r, self.w, self.h, self.rank, self.world = renderer, width, height, rank, world
        self.on_gpu = device != "cpu"                                # where the collective runs (RCCL on device tensors, gloo on CPU ones)
        self.streams = frames_device != "cpu"                        # False only under tools/emu, whose "device" memory is host memory
        self.rows = block_rows(height, world) * TILE_H               # pixel rows per rank (padded: the last block may reach past the frame)
        self.hp = self.rows * world
        self.G = groups
        # non-assembling buffers hold the rank's block only: N blocks per group (one per frame of the group)
        self.send = [torch.zeros((world, self.rows, width, 4), dtype=torch.uint8, device=frames_device) for _ in range(groups)]
        self.image = [torch.zeros((world, self.rows, width, 4), dtype=torch.uint8, device=device) for _ in range(groups)]   # [rank q's rows] = the padded frame
        self.cpu_send = None if (self.on_gpu or not self.streams) else [torch.zeros((world, self.rows, width, 4), dtype=torch.uint8) for _ in range(groups)]
        self.events = [None] * groups
        self.ext = {}
        self.n_groups = 0
        self.assembled = []                                          # (frame number, image view) this rank has assembled, newest last (at most `groups` are still intact)

    def upload(self, edges, paths, styles):
        self.r.upload_edges(edges, paths, styles)
        self.r.render_resident(1)                                    # one blocking frame: the handle learns which queued-row kernels this scene needs

    def _stream(self, k):
        import torch
        if k not in self.ext:
            self.ext[k] = torch.cuda.ExternalStream(self.r.stream_handle(k))
        return self.ext[k]

    def step_group(self, n_frames=None):
        """Queues the next N frames (one per rank to assemble) and their exchange; returns this rank's assembled frame of the group
        (a view: valid on the device once the exchange has run -- after finish(), or for work queued on torch's current stream).
        n_frames < N: a last, partial group -- only the frames for ranks 0 .. n_frames - 1 are rendered (the exchange still runs whole;
        the other ranks' images of this group are not new frames and None is returned there)."""
        import torch
        import torch.distributed as dist
        n_frames = self.world if n_frames is None else int(n_frames)
        gi = self.n_groups
        g = gi % self.G
        self.n_groups = gi + 1
        if self.streams and self.events[g] is not None:
            for k in range(4):                                       # the handle's streams: behind the exchange that last read this group's buffers
                if self.r.stream_handle(k):
                    self._stream(k).wait_event(self.events[g])
        used = self.r.render_resident_group_to([self.send[g][j].data_ptr() for j in range(n_frames)])
        if self.streams:
            cur = torch.cuda.current_stream()
            for k in range(4):
                if (used >> k) & 1:
                    cur.wait_stream(self._stream(k))                 # torch's stream: behind the group's kernels
        else:
            self.r.wait()
        src = self.send[g]
        if not self.on_gpu and self.streams:                         # gloo rehearsal on a GPU box: the collective runs on CPU tensors
            self.cpu_send[g].copy_(src)
            src = self.cpu_send[g]
        dist.all_to_all_single(self.image[g], src)                   # block (frame j, my rows) -> rank j; lands as rows of rank q in image[g]
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.events[g] = ev
        if self.rank >= n_frames:
            return None
        out = self.image[g].view(self.hp, self.w, 4)[: self.h]
        self.assembled.append((gi * self.world + self.rank, out))
        self.assembled = self.assembled[-self.G:]
        return out

    def finish(self):
        import torch
        self.r.wait()
        if self.streams:
            torch.cuda.synchronize()
        return self.assembled[-1][1] if self.assembled else None
