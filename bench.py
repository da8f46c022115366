#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mpixels/s rasterized (+ frames/s) at 4K on the 10k-edge
synthetic shape set (scene S1).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one frame, starting from the raw edge list resident in HBM: binning on the device
(band lists in painter's order, row chunks, scan-converter constants) -> per-row winding / cells -> launch order of the
strips -> tile raster/shade/blend -> RGBA8 framebuffer in HBM.  Consecutive frames rotate over SWFR_FRAMES_IN_FLIGHT
(default 4) sets of per-frame buffers, each on its own HIP stream.

N > 1 (one process per GPU): the frame's tile-rows are sharded over the ranks (north star: "frame tiles shard naturally
across the 8 GPUs ... with an RCCL gather") -- `--sharding bands`, the default, strong scaling: every step is one frame,
rank k rasterizes its contiguous block of tile-rows straight into a tensor, and one gather per frame deposits the blocks
in rank 0's image in place, overlapped with the next frame's kernels (swf_renderer_amd/distributed.py).  The same run also measures
BASELINE.json's config 5 (S2: 8K, 100k edges, same split) and reports it as `config5_s2_bands`.  `--sharding frames`
(whole frames per rank, no data-path collective, weak scaling) is kept as an explicitly named alternative.

Rank 0 prints ONE JSON line: the BASELINE metric, the HBM roofline of the tile kernel from HIP events in the timed region
(and the same kernel with one frame in flight), the full reference-style path (a new Stage every frame through
swfr_render, timed below the C-ABI), the frame checked against the libcairo known answer, and the CPU oracle timed on
this box (all cores and one).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured ceiling
DEFAULT_IN_FLIGHT = 4
BATCH_FRAMES, BATCH_LAUNCHES = 8, 12     # roofline.batched: frames per kernel launch, timed launches


def _oracle_args(fx, cols):
    import numpy as np
    argb = ((cols[:, 3].astype(np.uint32) << 24) | (cols[:, 0].astype(np.uint32) << 16) |
            (cols[:, 1].astype(np.uint32) << 8) | cols[:, 2]).astype(np.uint32)
    counts = np.full(len(fx), fx.shape[1], dtype=np.int32)
    xy = np.ascontiguousarray(fx.reshape(-1))
    return xy, counts, argb


def cpu_baseline(fx, cols, W, H, budget_s=8.0, name="S1"):
    """The oracle (C restatement of the reference's Cairo arithmetic) timed on this box's host cores on a bounded sample:
    whole frames of the workload, first on one thread, then one frame stream per core (ctypes releases the GIL)."""
    import threading
    from oracle import oracle_backend as ob
    L = ob.lib()
    xy, counts, argb = _oracle_args(fx, cols)

    def frames_for(seconds, out, k):
        n, t0 = 0, time.perf_counter()
        while True:
            ctx = L.swfo_create(W, H)
            L.swfo_fill_polygons_fixed(ctx, xy.ctypes.data, counts.ctypes.data, argb.ctypes.data, len(fx), 0)
            L.swfo_destroy(ctx)
            n += 1
            if time.perf_counter() - t0 > seconds or n >= 200:
                break
        out[k] = (n, time.perf_counter() - t0)

    one = [None]
    frames_for(budget_s / 2, one, 0)
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))     # (a GPU box's CPU share per GPU is 16 cores)
    res = [None] * cores
    t0 = time.perf_counter()
    th = [threading.Thread(target=frames_for, args=(budget_s / 2, res, k)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    return {"value": round(W * H * total / wall / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%d full %s frames (%dx%d, %d stars) in %.1f s on %d threads, one frame stream per thread, oracle/swfr_oracle.c" % (total, name, W, H, len(fx), wall, cores),
            "single_thread": {"value": round(W * H * one[0][0] / one[0][1] / 1e6, 2), "cores": 1,
                              "sample": "%d frames in %.1f s" % one[0]}}


def oracle_frame(fx, cols, W, H):
    """One frame of the scene by the CPU oracle as premultiplied RGBA8 (the checker of S2)."""
    import numpy as np
    from oracle import oracle_backend as ob
    L = ob.lib()
    ctx = L.swfo_create(W, H)
    xy, counts, argb = _oracle_args(fx, cols)
    L.swfo_fill_polygons_fixed(ctx, xy.ctypes.data, counts.ctypes.data, argb.ctypes.data, len(fx), 0)
    px = np.ctypeslib.as_array(L.swfo_pixels(ctx), shape=(H, W)).copy()
    L.swfo_destroy(ctx)
    return np.stack([(px >> 16) & 255, (px >> 8) & 255, px & 255, px >> 24], -1).astype(np.uint8)


def profile_commit(path):
    """' taken at commit <hash>' for a committed profile file profiles/<tag>_*.*: the tree the evidence was taken from (profiles/<tag>_commit.txt)."""
    import re
    m = re.match(r"(r\d+[a-z])_", os.path.basename(path or ""))
    if not m:
        return ""
    try:
        return " taken at commit " + open(os.path.join(ROOT, "profiles", m.group(1) + "_commit.txt")).read().split()[0]
    except Exception:
        return ""


def latest_traffic(kernel=None):
    """HBM bytes per launch from the newest committed PMC profile of the S1 bench (FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections:
    tools/profile_r02.py), with its source: of `kernel` (a substring of its name), or -- kernel None -- of the whole frame (the sum
    over the frame's kernels); (None, None) when there is none."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))):
        if re.fullmatch(r"r\d+[a-z]_pmc_summary\.json", os.path.basename(f)):      # (the S1 profiles, not the shaded-kernel / batched ones)
            best = f
    if not best:
        return None, None
    try:
        d = json.load(open(best))
        if kernel is None:
            return sum(int(v.get("hbm_bytes_per_launch", 0)) for k, v in d.items() if any(n in k for n in ("k2_bin_b", "k2_rows_b", "k2_tiles_solid_b"))), os.path.relpath(best, ROOT)
        for k, v in d.items():
            if kernel in k:
                return v.get("hbm_bytes_per_launch"), os.path.relpath(best, ROOT)
    except Exception:
        pass
    return None, None


def latest_rocprof_kernel_us(kernel):
    """rocprofv3's average duration (us) of `kernel` with one frame in flight from the newest committed kernel-trace summary of this
    bench (tools/profile_r02.sh), with its source; (None, None) when there is none."""
    import csv
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_one_frame_in_flight.csv"))):
        if re.fullmatch(r"r\d+[a-z]_kernel_stats_one_frame_in_flight\.csv", os.path.basename(f)):
            best = f
    if not best:
        return None, None
    try:
        for row in csv.DictReader(open(best)):
            if kernel in row["Name"]:
                return float(row["AverageNs"]) / 1000.0, os.path.relpath(best, ROOT)
    except Exception:
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-path", action="store_true", help="skip the reference-style full path (new Stage per frame) measurement")
    ap.add_argument("--no-batched", action="store_true", help="skip the saturated-GPU (frames per launch) and S0 store-roof measurements")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the frame against the known answer after the timed region")
    ap.add_argument("--verify", action="store_true", help="(default) kept for compatibility")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on one GPU: every rank renders on cuda:0, slabs are gathered as CPU tensors")
    ap.add_argument("--sharding", default="bands", choices=["frames", "bands"],
                    help="N>1: bands (default) = one frame's tile-rows sharded over the ranks + one RCCL gather per frame to rank 0 (strong scaling, "
                         "the north star's split); frames = every rank rasterizes whole frames, no data-path collective (weak scaling)")
    ap.add_argument("--assembly", default="rotate", choices=["rotate", "root"],
                    help="N>1, bands: rotate (default) = frame f is assembled on rank f mod N, the blocks of N consecutive frames move in ONE grouped "
                         "all-to-all (every rank's links busy in both directions); root = every frame gathered to rank 0 (bound by rank 0's inbound links)")
    ap.add_argument("--workload", default="s1", choices=["s1", "s2"],
                    help="s1 = BASELINE.json's metric configuration (4K, 10k edges; the default and the judged line); s2 = 8K, 100k edges")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched like the single-GPU line (python bench.py --gpus N ...): this process never touches the GPU; it starts the N
        # ranks as fresh child processes under torch.distributed.run, relays rank 0's JSON line and exits with the launcher's code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    import swf_renderer_amd as S
    from swf_renderer_amd import api, synth, distributed as D

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    in_flight = int(os.environ.get("SWFR_FRAMES_IN_FLIGHT", str(DEFAULT_IN_FLIGHT)))

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def scene_of(cfg):
        W, H = cfg["width"], cfg["height"]
        pts, cols = synth.scene(**cfg)
        fx = synth.twips_to_fixed(pts)
        # host half through the product's own C++ frame builder (register_shape + scene walk), as swfr_render does
        host = S.Renderer(W, H, device=api.DEVICE_HOST_ONLY)
        t0 = time.perf_counter()
        stage = api.stars_to_stage(pts, cols)
        edges, paths, styles = host.build_frame(stage)
        t_host = time.perf_counter() - t0
        host.close()
        return W, H, pts, cols, fx, stage, (edges, paths, styles), t_host

    def run_bands(cfg_name, cfg, steps, warmup):
        """One frame per step, tile-rows sharded over the ranks, one gather per frame; returns (seconds, image on rank 0, scene)."""
        W, H, pts, cols, fx, stage, scene, _ = scene_of(cfg)
        rb = S.Renderer(W, H, device=local_rank, band_index=rank, band_count=world, contiguous_bands=True)
        if args.assembly == "rotate":
            # frame f assembled on rank f mod N: N frames per group, one call below Python queues a group, ONE all-to-all moves its blocks
            pipe = D.RotatingPipeline(rb, W, H, rank, world, device="cpu" if rehearsal else "cuda")
            pipe.upload(*scene)

            def run(k):
                out = None
                for _ in range(k // world):
                    out = pipe.step_group()
                if k % world:
                    o = pipe.step_group(k % world)                # the last, partial group: exactly k frames are rendered
                    out = o if o is not None else out
                pipe.finish()
                return out
            run(max(warmup, 1))
            sync_all()
            t0 = time.perf_counter()
            out = run(steps)                                      # (this rank's newest assembled frame; None if it assembled none)
            sync_all()
            dt = time.perf_counter() - t0
        else:
            pipe = D.FramePipeline(rb, W, H, rank, world, device="cpu" if rehearsal else "cuda")   # (targets first: the upload bakes their addresses in)
            pipe.upload(*scene)
            for _ in range(max(warmup, 1)):
                pipe.step()
            pipe.finish()
            sync_all()
            t0 = time.perf_counter()
            for _ in range(steps):
                pipe.step()                                           # queue a frame + its gather; no host wait inside the loop
            out = pipe.finish()
            sync_all()
            dt = time.perf_counter() - t0
        if out is not None:
            out = out.clone()
        rb.render_resident(8)                                     # per-kernel HIP-event times of this rank's share (outside the timed region)
        tm = rb.timing()
        rb.close()
        return dt, out, (W, H, fx, cols, scene), tm

    bands = world > 1 and args.sharding == "bands"
    cfg = synth.S1 if args.workload == "s1" else synth.S2
    extra = {}
    t1 = None
    bms = None
    s0_info = None
    if bands:
        dt, out, (W, H, fx, cols, scene), tm = run_bands(args.workload, cfg, args.steps, args.warmup)
        edges, paths, styles = scene
        t_host = 0.0
        r = None
    else:
        W, H, pts, cols, fx, stage, (edges, paths, styles), t_host = scene_of(cfg)
        # the timed handle first (its allocations and its upload leave the GPU idle for milliseconds) ...
        os.environ["SWFR_FRAMES_IN_FLIGHT"] = str(in_flight)
        os.environ["SWFR_EVENT_STRIDE"] = "1000000"               # (the timed region carries no per-kernel events: each costs a queue packet)
        r = S.Renderer(W, H, device=local_rank)
        r.upload_edges(edges, paths, styles)                      # inputs resident in HBM before the timed region
        r1 = None
        if world == 1 and in_flight > 1:
            # ... then, directly before the W warm-up frames, the per-kernel times: the timed region below overlaps consecutive frames
            # on several streams, which stretches every kernel's own duration; the same kernels timed with one frame in flight (a second
            # handle, its own 80 frames) are reported beside it -- and the device is out of its idle clocks when the warm-up starts
            # (20 frames after 1-50 ms of idling and 5 warm-up frames take 7 % longer than back to back: tools/short_run_probe.py)
            os.environ["SWFR_FRAMES_IN_FLIGHT"] = "1"
            os.environ["SWFR_EVENT_STRIDE"] = "4"
            r1 = S.Renderer(W, H, device=local_rank)
            r1.upload_edges(edges, paths, styles)
            r1.render_resident(16)
            r1.render_resident(64)
            t1 = r1.timing()
            os.environ["SWFR_FRAMES_IN_FLIGHT"] = str(in_flight)
            os.environ["SWFR_EVENT_STRIDE"] = "1000000"
        r.render_resident(min(max(args.warmup, 1), 2048))
        sync_all()
        t0 = time.perf_counter()
        left = args.steps                                         # K frames queued back to back on the handle's streams
        while left > 0:                                           # (the library takes at most 4096 frames per call)
            n = min(left, 2048)
            r.render_resident(n)
            left -= n
        sync_all()
        dt = time.perf_counter() - t0
        tm = r.timing()
        out = None
        if r1 is not None:
            r1.close()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    verified = None
    if not args.no_verify and rank == 0:
        import hashlib
        img = out.cpu().numpy() if bands else r.read_image(premultiplied=True)
        if args.workload == "s1":
            verified = hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == synth.S1_SHA256_PREMUL
        else:
            verified = bool(np.array_equal(np.asarray(img), oracle_frame(fx, cols, W, H)))
        if not verified:
            print("verify: the frame of the timed region DOES NOT match the %s" % ("libcairo known answer" if args.workload == "s1" else "CPU oracle"), file=sys.stderr, flush=True)
    if bands and args.assembly == "rotate" and not args.no_verify:
        # every rank assembled frames of its own: their digests must all be rank 0's (whose frame was checked against the known answer)
        import hashlib
        mine = hashlib.sha256(np.ascontiguousarray(out.cpu().numpy()).tobytes()).hexdigest() if out is not None else None
        digests = [None] * world
        dist.all_gather_object(digests, mine)
        if rank == 0:
            same = all(d is None or d == digests[0] for d in digests) and digests[0] is not None
            if not same:
                print("verify: frames assembled on different ranks differ", file=sys.stderr, flush=True)
            verified = bool(verified) and same
            extra["frames_assembled_on_ranks"] = [i for i, d in enumerate(digests) if d is not None]
    if world > 1:
        # every rank learns the verdict and leaves together (a rank that exits alone leaves the others in their next collective)
        flag = torch.tensor([0 if verified is False else 1], dtype=torch.int32, device="cpu" if rehearsal else "cuda")
        dist.broadcast(flag, src=0)
        if int(flag.item()) == 0:
            dist.destroy_process_group()
            raise SystemExit(3)
    elif verified is False:
        raise SystemExit(3)

    if world == 1 and not args.no_batched and rank == 0:
        # ---- after the timed region, on a handle of its own (one frame set): the saturated GPU and the store roof
        os.environ["SWFR_FRAMES_IN_FLIGHT"] = "1"
        os.environ["SWFR_EVENT_STRIDE"] = "4"
        r1 = S.Renderer(W, H, device=local_rank)
        r1.upload_edges(edges, paths, styles)
        r1.render_resident(4)
        # the saturated GPU: the same resident scene as 8 frames per kernel launch (blockIdx.y = frame), 12 launches
        bms = r1.render_resident_batched(BATCH_FRAMES, BATCH_LAUNCHES)
        if not args.no_verify and args.workload == "s1":
            import hashlib
            if hashlib.sha256(np.ascontiguousarray(r1.read_image(premultiplied=True)).tobytes()).hexdigest() != synth.S1_SHA256_PREMUL:
                print("verify: the last frame of the batched launches DOES NOT match the libcairo known answer", file=sys.stderr, flush=True)
                raise SystemExit(3)
        # S0: one full-frame opaque rectangle through the same path -- the store roof of the tile pass
        rect = np.array([[[0, 0], [W * 256, 0], [W * 256, H * 256], [0, H * 256]]], dtype=np.int32)
        e0, p0, s0 = api.polygons_to_scene(rect, np.array([[30, 60, 90, 255]], dtype=np.uint8), W, H)
        r1.upload_edges(e0, p0, s0)
        r1.render_resident(16)
        r1.render_resident(64)
        t0s = r1.timing()
        s0_ok = bool((r1.read_image(premultiplied=True) == np.array([30, 60, 90, 255], dtype=np.uint8)).all())
        b0ms = r1.render_resident_batched(BATCH_FRAMES, BATCH_LAUNCHES)
        s0_info = (t0s, b0ms, len(e0), len(p0), s0_ok)
        r1.close()
        os.environ["SWFR_FRAMES_IN_FLIGHT"] = str(in_flight)
    # ---- secondary measurements of the same run (outside the timed region)
    if bands and args.workload == "s1":
        # BASELINE.json config 5: the 100k-edge 8K scene over the same ranks
        dt5, out5, (W5, H5, fx5, cols5, scene5), _ = run_bands("s2", synth.S2, max(5, min(args.steps, 40)), 3)
        t5 = torch.tensor([dt5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t5, op=dist.ReduceOp.MAX)
        k5 = max(5, min(args.steps, 40))
        v5 = None
        if rank == 0 and not args.no_verify and out5 is not None:
            v5 = bool(np.array_equal(out5.cpu().numpy(), oracle_frame(fx5, cols5, W5, H5)))
            if not v5:
                print("verify: the gathered S2 frame of config 5 DOES NOT match the CPU oracle", file=sys.stderr, flush=True)
        extra["config5_s2_bands"] = {"workload": "S2: 7680x4320, 10000 stars (99909 edges), tile-rows sharded over %d ranks, one gather per frame" % world, "verified": v5,
                                     "value": round(W5 * H5 * k5 / float(t5.item()) / 1e6, 2), "unit": "Mpixels/s",
                                     "frames_per_sec": round(k5 / float(t5.item()), 1), "steps": k5}
    if bands:
        # the same ranks with NO exchange step: every rank rasterizes whole frames of its own (weak scaling), for comparison with the
        # gather-bound figure above -- `--sharding frames` makes this the headline instead
        scene_w = scene
        rw = S.Renderer(W, H, device=local_rank)
        rw.upload_edges(*scene_w)
        rw.render_resident(max(args.warmup, 2))
        sync_all()
        tw0 = time.perf_counter()
        rw.render_resident(min(args.steps, 2048))
        sync_all()
        tw = torch.tensor([time.perf_counter() - tw0], dtype=torch.float64, device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        rw.close()
        kw = min(args.steps, 2048)
        extra["whole_frames_per_rank"] = {"what": "no data-path collective: every rank renders %d whole %s frames of its own" % (kw, args.workload.upper()),
                                          "value": round(W * H * kw * world / float(tw.item()) / 1e6, 2), "unit": "Mpixels/s", "scaling": "weak"}
    if rank == 0:
        n_edges, n_paths = len(edges), len(paths)
        algo_bytes = 4 * W * H + 16 * n_edges + 16 * n_paths           # SURVEY.md 8(d), per frame = per tile-kernel launch
        if bands:
            algo_bytes = 4 * W * min(D.block_rows(H, world) * D.TILE_H, H) + 16 * n_edges + 16 * n_paths
        frames_total = args.steps * (1 if bands or world == 1 else world)
        ms_per_step = dt / args.steps * 1e3
        # per-kernel durations: HIP events on the kernels' own stream with ONE frame in flight (a kernel has the GPU to itself, so the
        # interval between its events is its duration; under several frames in flight an event interval also holds queue waits).
        # N > 1: this rank's share of the frame, eight frames after the timed region.
        tk = t1 if t1 is not None else tm
        nk = max(tk["timed_frames"], 1)
        bin_ms, rows_ms, tiles_ms = tk["setup_ms"] / nk, tk["rows_ms"] / nk, tk["tiles_ms"] / nk
        gbs = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # the roofline's lead entry is the kernel with the LONGEST isolated duration (the frame is bounded by it), against the frame's
        # algorithmic bytes; every kernel of the frame is listed beside it, and frame_frac prices the frame against the sum of them
        kern = {"k2_bin": ("k2_bin_b", bin_ms), "k2_rows": ("k2_rows_b", rows_ms), "k2_tiles": ("k2_tiles_solid_b", tiles_ms)}
        lead = max(("k2_rows", "k2_tiles"), key=lambda k: kern[k][1])
        lead_sym, lead_ms = kern[lead]
        achieved = gbs(algo_bytes, lead_ms)
        single_s1 = not bands and args.workload == "s1"
        traffic, traffic_src = latest_traffic(lead_sym) if single_s1 else (None, None)
        frame_traffic, _ = latest_traffic(None) if single_s1 else (None, None)
        sum_ms = bin_ms + rows_ms + tiles_ms
        line = {
            "metric": "Mpixels/sec rasterized @ 4K, 10k-edge synthetic shape set" if args.workload == "s1" else "Mpixels/sec rasterized @ 8K, 100k-edge synthetic shape set",
            # bands sharding: one frame over all ranks per step; frames sharding: a step is one frame on every rank (N frames)
            "value": round(W * H * frames_total / dt / 1e6, 2),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "frames_per_sec": round(frames_total / dt, 1),
            "higher_is_better": True,
            "scaling": "strong" if bands else "weak",
            "vs_baseline": None,
            "dtype": "i32+f64(exact integers)/u8",
            "data": "synthetic",
            "verified": verified,
            "config": {"workload": "%s: %dx%d, %d ten-vertex stars, opaque solid, nonzero, seed 0xC0FFEE" % (args.workload.upper(), W, H, len(fx)),
                       "n_edges": n_edges, "n_paths": n_paths,
                       "sharding": (("contiguous blocks of tile-rows over %d ranks; frame f assembled on rank f mod %d, the blocks of %d consecutive frames exchanged in ONE grouped all-to-all, overlapped with the next group's kernels" % (world, world, world)) if args.assembly == "rotate" else
                                    ("contiguous blocks of tile-rows over %d ranks, rendered in place, one RCCL gather per frame to rank 0 overlapped with the next frame" % world)) if bands else
                                   ("whole frames, one per rank and step, no data-path collective" if world > 1 else "single GPU"),
                       "frames_in_flight": in_flight,
                       "device_path": "raw edge list -> k2_bin -> k2_rows -> k2_tiles, every frame"},
            "kernel_ms_per_frame": {"k2_bin": round(bin_ms, 4), "k2_rows": round(rows_ms, 4), "k2_tiles": round(tiles_ms, 4),
                                    "how": "HIP events on the kernels' stream, one frame in flight (each kernel alone on the GPU), %d frames; in the timed region "
                                           "%d frames overlap, so these add up to more than ms_per_step" % (nk, in_flight)},
            "roofline": {"bound": "hbm", "kernel": lead, "kernel_why": "the kernel of the frame with the longest duration when it has the GPU to itself",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": ("committed profile %s%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench; not measured in this run)" % (traffic_src, profile_commit(traffic_src))) if traffic_src else None,
                         "frame_traffic": frame_traffic,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel_ms": round(lead_ms, 4), "kernel_ms_how": "one frame in flight, HIP events around the kernel on its own stream",
                         # every kernel of the frame against the same bytes, the sum of them, and the whole pipelined step
                         "kernels": {k: {"kernel_ms": round(v[1], 4), "achieved": round(gbs(algo_bytes, v[1]), 2), "frac": round(gbs(algo_bytes, v[1]) / HBM_PEAK_GBS, 5)} for k, v in kern.items()},
                         "frame_frac": round(gbs(algo_bytes, sum_ms) / HBM_PEAK_GBS, 5),
                         "frame_frac_how": "algorithmic bytes / (k2_bin + k2_rows + k2_tiles durations, one frame in flight)",
                         "step_achieved": round(gbs(algo_bytes * (frames_total / args.steps), ms_per_step), 2),
                         "step_frac": round(gbs(algo_bytes * (frames_total / args.steps), ms_per_step) / HBM_PEAK_GBS, 5)},
        }
        if single_s1:
            committed = {}
            for k, (sym, _) in kern.items():
                rp_us, rp_src = latest_rocprof_kernel_us(sym)
                if rp_us:
                    # (the HIP events of this run bracket a kernel on its stream and so include the ~3 us between the end of the kernel before
                    #  it and its own start; rocprofv3's kernel trace does not)
                    committed[k] = {"kernel_us": round(rp_us, 2), "achieved": round(gbs(algo_bytes, rp_us * 1e-3), 2), "frac": round(gbs(algo_bytes, rp_us * 1e-3) / HBM_PEAK_GBS, 5)}
                    committed["source"] = "%s%s (rocprofv3 --kernel-trace --stats of this bench, one frame in flight; not measured in this run)" % (rp_src, profile_commit(rp_src))
            if committed:
                line["roofline"]["rocprofv3_committed"] = committed
        line.update(extra)
        if t1 is not None:
            line["roofline"]["one_frame_in_flight"] = {"k2_tiles_ms": round(tiles_ms, 4), "k2_rows_ms": round(rows_ms, 4), "k2_bin_ms": round(bin_ms, 4),
                                                       "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5),
                                                       "value": round(W * H * 64 / (t1["total_ms"] * 1e-3) / 1e6, 2)}
        if bms is not None:
            bf = bms / (BATCH_FRAMES * BATCH_LAUNCHES)
            line["roofline"]["batched"] = {"what": "the resident scene as %d frames per kernel launch (blockIdx.y = frame, every frame recomputed from the raw edge list into "
                                                   "its own buffers and framebuffer), %d launches, HIP events around them; last frame verified" % (BATCH_FRAMES, BATCH_LAUNCHES),
                                           "frames_per_launch": BATCH_FRAMES, "launches": BATCH_LAUNCHES, "ms_per_frame": round(bf, 4),
                                           "Mpixels_per_sec": round(W * H / (bf * 1e-3) / 1e6, 1),
                                           "achieved": round(gbs(algo_bytes, bf), 2), "frac": round(gbs(algo_bytes, bf) / HBM_PEAK_GBS, 5)}
        if s0_info is not None:
            t0s, b0ms, ne0, np0, s0_ok = s0_info
            n0 = max(t0s["timed_frames"], 1)
            a0 = 4 * W * H + 16 * ne0 + 16 * np0
            k0 = t0s["tiles_ms"] / n0
            b0 = b0ms / (BATCH_FRAMES * BATCH_LAUNCHES)
            line["roofline"]["s0"] = {"what": "S0: one full-frame opaque rectangle through the same path (the store roof of the tile pass)", "verified": s0_ok,
                                      "algorithmic_bytes_per_launch": a0, "k2_tiles_ms": round(k0, 4), "achieved": round(gbs(a0, k0), 2), "frac": round(gbs(a0, k0) / HBM_PEAK_GBS, 5),
                                      "frame_ms_one_in_flight": round(t0s["total_ms"] / 64, 4),
                                      "batched": {"frames_per_launch": BATCH_FRAMES, "ms_per_frame": round(b0, 4), "achieved": round(gbs(a0, b0), 2), "frac": round(gbs(a0, b0) / HBM_PEAK_GBS, 5)}}
        if world == 1 and not args.no_full_path and args.workload == "s1":
            # ---- the reference's calling pattern: a different Stage every frame through swfr_render (blocking), timed below the
            #      C-ABI (swfr_render_sequence); four variants of S1 (the stars shifted by 0..3 px) so that no frame repeats the last
            variants = []
            for k in range(4):
                p2 = pts.copy()
                p2[..., 0] += 20 * k
                variants.append(api.stars_to_stage(p2, cols))
            r.render_sequence(variants, 1)                        # registers the definitions, warms the buffers
            secs, acc = r.render_sequence(variants, 8)
            nfr = 4 * 8
            t0 = time.perf_counter()
            r.read_image(premultiplied=True)
            d2h = time.perf_counter() - t0
            full = {"frames_per_sec": round(nfr / secs, 1), "Mpixels_per_sec": round(W * H * nfr / secs / 1e6, 1),
                    "what": "swfr_render of a new Stage per frame (blocking), %d frames, timed in C" % nfr,
                    "build_host_ms": round(acc["build_ms"] / nfr, 4), "upload_host_ms": round(acc["upload_host_ms"] / nfr, 4),
                    "h2d_ms": round(acc["h2d_ms"] / nfr, 4), "h2d_bytes": int(acc["h2d_bytes"] / nfr), "device_ms": round(acc["device_ms"] / nfr, 4),
                    "d2h_ms": round(d2h * 1e3, 3), "d2h_what": "swfr_read_image of the 33 MB frame into pageable host memory"}
            # render + get_image of every frame (the reference's test loop): mapped read-back through the handle's pinned staging buffer
            r.read_image_async(premultiplied=True)                # (the first call allocates the pinned staging buffer)
            r.read_image_wait()
            t0 = time.perf_counter()
            r.read_image_async(premultiplied=True)
            r.read_image_wait()
            full["d2h_pinned_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            mv = r.marshal_stages(variants)
            r.render_sequence_readback(mv, 1)
            full["with_readback_frames_per_sec"] = round(nfr / r.render_sequence_readback(mv, 8, premultiplied=True, overlap=True), 1)
            full["with_readback_blocking_frames_per_sec"] = round(nfr / r.render_sequence_readback(mv, 8, premultiplied=True, overlap=False), 1)
            full["with_readback_what"] = "swfr_render + swfr_read_image_async/_wait (pinned staging, mapped) per frame; the copy of frame i overlaps the host build of frame i+1 / is waited for at once"
            # the same frames as one pipelined batch (swfr_render_batch: the host builds frame i+1 while frame i is rasterized)
            frames_t = torch.empty((16, H, W, 4), dtype=torch.uint8, device="cuda")
            batch = r.marshal_stages(variants * 4)                # (the ctypes form once: the C call is what is timed)
            r.render_batch(batch, frames_t.data_ptr(), H * W * 4)
            t0 = time.perf_counter()
            for _ in range(2):
                r.render_batch(batch, frames_t.data_ptr(), H * W * 4)
            tb = time.perf_counter() - t0
            full["batch_frames_per_sec"] = round(32 / tb, 1)
            full["batch_what"] = "swfr_render_batch of 16 frames (four groups of four: the host builds one group while the GPU renders the other), two calls"
            del frames_t
            line["full_path"] = full
        line["config"]["host_edge_list_build_ms_python"] = round(t_host * 1e3, 2)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(fx, cols, W, H, name=args.workload.upper())
        print(json.dumps(line), flush=True)
    if r is not None:
        r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
