#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mpixels/s rasterized (+ frames/s) at 4K on the 10k-edge
synthetic shape set (scene S1).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one frame: edge setup -> per-row winding -> tile raster/shade/blend
-> RGBA8 framebuffer in HBM, with the edge list already resident in HBM; consecutive frames alternate between two
sets of per-frame buffers on two HIP streams (SWFR_FRAMES_IN_FLIGHT).  With N > 1 every rank rasterizes whole frames
of the batch (a step is one frame per rank, weak scaling, no data-path collective); `--sharding bands` instead
shards one frame's tile-rows over the ranks and ends every step with one RCCL gather to rank 0 (strong scaling).
Rank 0 prints ONE JSON line: the BASELINE metric, the HBM roofline of k_tiles from HIP events in the timed region
(and the same kernel with one frame in flight beside it), and the CPU oracle timed on this box.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured ceiling


def cpu_baseline(fx, cols, W, H, budget_s=12.0, name="S1"):
    """The oracle (single-thread C restatement) timed on this box's host cores on a bounded sample."""
    import numpy as np
    from oracle import oracle_backend as ob
    L = ob.lib()
    argb = ((cols[:, 3].astype(np.uint32) << 24) | (cols[:, 0].astype(np.uint32) << 16) |
            (cols[:, 1].astype(np.uint32) << 8) | cols[:, 2]).astype(np.uint32)
    counts = np.full(len(fx), fx.shape[1], dtype=np.int32)
    xy = np.ascontiguousarray(fx.reshape(-1))
    frames, t0 = 0, time.perf_counter()
    while True:
        ctx = L.swfo_create(W, H)
        L.swfo_fill_polygons_fixed(ctx, xy.ctypes.data, counts.ctypes.data, argb.ctypes.data, len(fx), 0)
        L.swfo_destroy(ctx)
        frames += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or frames >= 200:
            break
    return {"value": round(W * H * frames / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": "%d full %s frames (%dx%d, %d stars) in %.1f s, single thread, oracle/swfr_oracle.c" % (frames, name, W, H, len(fx), dt)}


def oracle_frame(fx, cols, W, H):
    """One frame of the scene by the CPU oracle as premultiplied RGBA8 (checker for --verify on S2)."""
    import numpy as np
    from oracle import oracle_backend as ob
    L = ob.lib()
    ctx = L.swfo_create(W, H)
    argb = ((cols[:, 3].astype(np.uint32) << 24) | (cols[:, 0].astype(np.uint32) << 16) |
            (cols[:, 1].astype(np.uint32) << 8) | cols[:, 2]).astype(np.uint32)
    counts = np.full(len(fx), fx.shape[1], dtype=np.int32)
    xy = np.ascontiguousarray(fx.reshape(-1))
    L.swfo_fill_polygons_fixed(ctx, xy.ctypes.data, counts.ctypes.data, argb.ctypes.data, len(fx), 0)
    px = np.ctypeslib.as_array(L.swfo_pixels(ctx), shape=(H, W)).copy()
    L.swfo_destroy(ctx)
    return np.stack([(px >> 16) & 255, (px >> 8) & 255, px & 255, px >> 24], -1).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on one GPU: every rank renders on cuda:0, slabs are gathered as CPU tensors")
    ap.add_argument("--verify", action="store_true", help="rank 0 checks the assembled frame against the S1 known answer (S2: against the oracle)")
    ap.add_argument("--sharding", default="frames", choices=["frames", "bands"],
                    help="N>1: frames = every rank rasterizes whole frames of the batch, no data-path collective (weak scaling, default); "
                         "bands = one frame's tile-rows interleaved over the ranks + one RCCL gather per frame to rank 0 (strong scaling)")
    ap.add_argument("--workload", default="s1", choices=["s1", "s2"],
                    help="s1 = BASELINE.json's metric configuration (4K, 10k edges; the default and the judged line); s2 = 8K, 100k edges")
    args = ap.parse_args()

    import numpy as np
    import torch
    import swf_renderer_amd as S
    from swf_renderer_amd import api, synth, distributed as D

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
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

    cfg = synth.S1 if args.workload == "s1" else synth.S2
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

    bands = world > 1 and args.sharding == "bands"
    r = S.Renderer(W, H, device=local_rank, band_index=rank if bands else 0, band_count=world if bands else 0)
    r.upload_edges(edges, paths, styles)                      # inputs resident in HBM before the timed region
    pipe = None
    if bands:
        pipe = D.FramePipeline(r, W, H, rank, world, device="cpu" if rehearsal else "cuda")

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step_multi():
        return pipe.step()

    # ---- warmup
    if not bands:
        r.render_resident(min(max(args.warmup, 1), 2048))
    else:
        for _ in range(max(args.warmup, 1)):
            step_multi()
    sync_all()
    # ---- timed region: exactly K steps
    t0 = time.perf_counter()
    if not bands:
        left = args.steps                                     # K frames queued back to back on the handle's streams
        while left > 0:                                       # (the library takes at most 4096 frames per call)
            n = min(left, 2048)
            r.render_resident(n)
            left -= n
    else:
        out = None
        for _ in range(args.steps):
            out = step_multi()
        out = pipe.finish()
    sync_all()
    dt = time.perf_counter() - t0
    tm = r.timing()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if args.verify and rank == 0:
        import hashlib
        img = out.cpu().numpy() if bands else r.read_image(premultiplied=True)
        if args.workload == "s1":
            ok = hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == synth.S1_SHA256_PREMUL
        else:
            ok = np.array_equal(np.asarray(img), oracle_frame(fx, cols, W, H))
        print("verify: assembled frame %s the %s" % ("matches" if ok else "DOES NOT match", "libcairo known answer" if args.workload == "s1" else "CPU oracle"), file=sys.stderr, flush=True)
        if not ok:
            raise SystemExit(3)
    if rank == 0:
        n_edges, n_paths = len(edges), len(paths)
        algo_bytes = 4 * W * H + 16 * n_edges + 16 * n_paths           # SURVEY.md 8(d), per frame = per k_tiles launch
        if bands:
            algo_bytes = 4 * W * D.local_tile_rows(H, 0, world) * D.TILE_H + 16 * n_edges + 16 * n_paths
        tiles_ms = tm["tiles_ms"] / max(tm["timed_frames"], 1)
        achieved = algo_bytes / (tiles_ms * 1e-3) / 1e9 if tiles_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01m_pmc_k_tiles.json")
        if os.path.exists(pmc) and not bands and args.workload == "s1":
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        line = {
            "metric": "Mpixels/sec rasterized @ 4K, 10k-edge synthetic shape set" if args.workload == "s1" else "Mpixels/sec rasterized @ 8K, 100k-edge synthetic shape set",
            # frames sharding: a step is one frame on every rank (N frames); bands sharding: one frame over all ranks
            "value": round(W * H * args.steps * (1 if bands or world == 1 else world) / dt / 1e6, 2),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "frames_per_sec": round(args.steps * (1 if bands or world == 1 else world) / dt, 1),
            "higher_is_better": True,
            "scaling": "strong" if bands else "weak",
            "vs_baseline": None,
            "dtype": "int64/u8",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d ten-vertex stars, opaque solid, nonzero, seed 0xC0FFEE" % (args.workload.upper(), W, H, len(fx)),
                       "n_edges": n_edges, "n_paths": n_paths,
                       "sharding": ("tile-row bands interleaved over %d ranks, one RCCL gather per frame" % world) if bands else
                                   ("whole frames, one per rank and step, no data-path collective" if world > 1 else "single GPU"),
                       "host_edge_list_build_ms": round(t_host * 1e3, 2)},
            "kernel_ms_per_frame": {"k_setup": round(tm["setup_ms"] / max(tm["timed_frames"], 1), 4), "k_rows": round(tm["rows_ms"] / max(tm["timed_frames"], 1), 4),
                                    "k_tiles": round(tiles_ms, 4)},
            "roofline": {"bound": "hbm", "kernel": "k_tiles", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes},
        }
        line["config"]["frames_in_flight"] = int(os.environ.get("SWFR_FRAMES_IN_FLIGHT", "2")) if not bands else 1
        if world == 1 and line["config"]["frames_in_flight"] > 1:
            # the timed region overlaps consecutive frames on two streams, which stretches every kernel's own duration;
            # the same kernel timed with one frame in flight (a second handle, outside the timed region) is reported beside it
            os.environ["SWFR_FRAMES_IN_FLIGHT"] = "1"
            r1 = S.Renderer(W, H, device=local_rank)
            r1.upload_edges(edges, paths, styles)
            r1.render_resident(16)
            r1.render_resident(64)
            t1 = r1.timing()
            r1.close()
            iso_ms = t1["tiles_ms"] / max(t1["timed_frames"], 1)
            iso = algo_bytes / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
            line["roofline"]["one_frame_in_flight"] = {"k_tiles_ms": round(iso_ms, 4), "achieved": round(iso, 2), "frac": round(iso / HBM_PEAK_GBS, 5),
                                                       "value": round(W * H * 64 / (t1["total_ms"] * 1e-3) / 1e6, 2)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(fx, cols, W, H, name=args.workload.upper())
        print(json.dumps(line), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
