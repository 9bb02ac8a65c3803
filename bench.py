#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric "primary + diffuse bounce" is quoted on): the reference's
bunny scene (3 instanced bunnies + glossy floor + real skybox) at 1080x960, antialiasing 8 (64 spp),
max_depth 5, shadow_rays 1, scatter_rays 1.  One step = one full frame through the wavefront kernels
(ray generation, closest-hit trace, shade, shadow trace, compaction), scene and skybox resident in HBM.
Metric: Mrays/s = (closest-hit + any-hit rays traced) / wall time, whole job.

For N > 1 (launched by torch.distributed.run, one rank per GPU) the frame grows with N at the same camera, aspect
ratio and samples per pixel (N x 1080x960 pixels: 1528x1358, 2160x1920, 3056x2715 for N = 2, 4, 8), its 64x64 tiles
are dealt round-robin to the ranks, every rank renders its tiles, and the framebuffer is gathered to rank 0 with
one RCCL gather inside the timed region.  Per-GPU work is fixed, so this is weak scaling: a 1080x960 frame cut in
eight leaves each GPU launches of a few ten thousand rays, which measure launch latency rather than the path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

WORKLOADS = {
    # name: (camera, antialiasing, max_depth, shadow_rays)
    "bunny_1080x960_64spp_depth5": ((1080, 960), 8, 5, 1),
    "bunny_3840x2160_primary": ((3840, 2160), 1, 1, 1),
}
TILE = 64
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def cpu_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env))
    return max(1, min(n, 64))


def cpu_baseline(sky, cam, aa, max_depth, shadow_rays):
    """Times the CPU oracle (C++ restatement of the reference renderer) on a bounded sample of the workload and
    returns its throughput plus the reference-order visit counters used for the algorithmic byte model."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    scene = ol.OracleScene.main(sky=sky, cam=cam)
    sample_aa = max(1, aa // 2)  # same frame, scene, depth and lights at a quarter of the samples per pixel
    threads = cpu_threads()
    _, sec, cnt = scene.render(max_depth=max_depth, shadow_rays=shadow_rays, antialiasing=sample_aa, threads=threads)
    rays = cnt["closest_rays"] + cnt["any_rays"]
    base = {
        "value": rays / sec / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"full {cam[0]}x{cam[1]} frame at antialiasing {sample_aa} ({sample_aa * sample_aa} spp) instead of {aa * aa} spp, "
                  f"{rays} rays in {sec:.2f} s, 100-pixel chunks over {threads} threads",
    }
    return base, cnt


def byte_model(cnt):
    """Algorithmic bytes per ray from the oracle's reference-order visit counters:
    closest ray: 32 B in + 16 B out + 32 B per node visited + 36 B per triangle tested + 36 B per cuboid test;
    any-hit ray: 32 B in + 4 B out + the same traversal terms."""
    nc = cnt["nodes_visited"] - cnt["nodes_visited_any"]
    tc = cnt["tris_tested"] - cnt["tris_tested_any"]
    cc = cnt["cuboids_tested"] - cnt["cuboids_tested_any"]
    closest = 48.0 + (32.0 * nc + 36.0 * tc + 36.0 * cc) / max(1, cnt["closest_rays"])
    anyb = 36.0 + (32.0 * cnt["nodes_visited_any"] + 36.0 * cnt["tris_tested_any"] + 36.0 * cnt["cuboids_tested_any"]) / max(1, cnt["any_rays"])
    return closest, anyb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="bunny_1080x960_64spp_depth5", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--small-sky", action="store_true", help="use a strided skybox (debug)")
    ap.add_argument("--opt", default="", help="tuning overrides, e.g. chunk=128,blocks=3,lds=0,max_paths=2097152")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the render path)")
    # RTAMD_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # the gather goes through host memory); the driver's runs use the default, RCCL with one GPU per rank.
    backend = os.environ.get("RTAMD_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    rt = entry.load_package()
    from rust_raytracer_amd import dist_tiles

    cam_wh, aa, max_depth, shadow_rays = WORKLOADS[args.workload]
    base_wh = cam_wh
    if world > 1:  # weak scaling: world x the pixels, same aspect ratio
        w = int(round(cam_wh[0] * (world ** 0.5) / 8.0)) * 8
        cam_wh = (w, int(round(world * cam_wh[0] * cam_wh[1] / w)))
    sky = rt.load_skybox(max_width=1024 if args.small_sky else None)
    scene = rt.Scene.main(sky=sky, cam=cam_wh).upload(dev_index)
    opt_ids = {"chunk": rt.OPT_CHUNK, "blocks": rt.OPT_BLOCKS_PER_CU, "lds": rt.OPT_LDS_SCENE, "max_paths": rt.OPT_MAX_PATHS, "streams": rt.OPT_STREAMS, "static": rt.OPT_STATIC_SHARE, "min_batch": rt.OPT_MIN_BATCH}
    for kv in filter(None, args.opt.split(",")):
        k, v = kv.split("=")
        scene.set_option(opt_ids[k], int(v))
    W, H = cam_wh
    region = rt.Scene.make_region(W, H, tile=(TILE, TILE), tile_first=rank, tile_stride=world)
    n_px = rt.Scene.region_pixels(region)
    tiles = torch.zeros((n_px // (TILE * TILE), TILE, TILE, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        _, st = scene.render(max_depth=max_depth, shadow_rays=shadow_rays, antialiasing=aa, region=region, out_device=tiles)
        img = dist_tiles.gather_frame(tiles, W, H, TILE, TILE, rank, world)
        return st, img

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    rays = 0
    dev_ms = 0.0
    for _ in range(args.steps):
        st, img = step()
        rays += st["closest_rays"] + st["any_rays"]
        dev_ms += st["device_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    closest_per_step, any_per_step, paths_per_step = st["closest_rays"], st["any_rays"], st["paths"]

    if world > 1:
        t = torch.tensor([elapsed, float(rays), float(closest_per_step), float(any_per_step), float(paths_per_step)], dtype=torch.float64,
                         device=dev if backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays = float(t[1])
        closest_per_step, any_per_step, paths_per_step = float(t[2]), float(t[3]), float(t[4])

    # ---- rank 0, N = 1: roofline of the dominant kernel (HIP events on the kernels' own stream) + CPU baseline ----
    roofline = None
    cpu = None
    extra = {}
    if rank == 0:
        scene.set_option(rt.OPT_PROFILE, 1)
        scene.kernel_times(reset=True)
        _, st = scene.render(max_depth=max_depth, shadow_rays=shadow_rays, antialiasing=aa, region=region, out_device=tiles)
        kt = scene.kernel_times(reset=True)
        scene.set_option(rt.OPT_PROFILE, 0)
        cnt = None
        if not args.no_cpu_baseline and world == 1:
            cpu, cnt = cpu_baseline(sky, cam_wh, aa, max_depth, shadow_rays)
        # algorithmic bytes per ray of this workload: from the oracle's visit counters when the CPU leg ran (and kept
        # in profiles/byte_model.json), else the kept values (N > 1 ranks trace the same scene with the same mix of rays)
        mpath = os.path.join(ROOT, "profiles", "byte_model.json")
        model = None
        if cnt is not None:
            model = byte_model(cnt)
            try:
                kept = json.load(open(mpath)) if os.path.exists(mpath) else {}
                kept[args.workload] = {"closest": model[0], "any": model[1]}
                json.dump(kept, open(mpath, "w"), indent=1)
            except Exception:
                pass
        elif os.path.exists(mpath):
            try:
                m = json.load(open(mpath)).get(args.workload)
                model = (m["closest"], m["any"]) if m else None
            except Exception:
                model = None
        if model is not None and kt["closest_launches"]:
            # k_wave_trace carries the closest-hit rays of a bounce plus the shadow rays of the previous one; the
            # launch that ends a sample carries shadow rays only.  All its launches are pooled.
            b_closest, b_any = model
            n_launch = kt["closest_launches"] + kt["any_launches"]
            trace_ms = kt["closest_ms"] + kt["any_ms"]
            avg_ms = trace_ms / n_launch
            rays_per_launch = (st["closest_rays"] + st["any_rays"]) / n_launch
            bytes_per_launch = (st["closest_rays"] * b_closest + st["any_rays"] * b_any) / n_launch
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("workload") == args.workload:
                        traffic = tj.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "kernel": "k_wave_trace (persistent closest-hit + any-hit traversal of the wavefront queues)",
                        "avg_launch_ms": avg_ms, "launches": n_launch, "rays_per_launch": rays_per_launch,
                        "algorithmic_bytes_per_closest_ray": b_closest, "algorithmic_bytes_per_any_ray": b_any,
                        "stream_only_gbs": (st["closest_rays"] * 48.0 + st["any_rays"] * 36.0) / (trace_ms * 1e-3) / 1e9}
            extra["trace_kernel_mrays_per_s"] = (st["closest_rays"] + st["any_rays"]) / (trace_ms * 1e-3) / 1e6
            extra["stage_ms_per_frame"] = {"trace": trace_ms, "shade_resolve": st["shade_ms"], "device_total": st["device_ms"]}

    if rank == 0:
        value = rays / elapsed / 1e6
        line = {
            "metric": "Mrays/s (primary + diffuse bounce) on bunny BVH", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": W, "height": H, "spp": aa * aa, "max_depth": max_depth,
                       "shadow_rays": shadow_rays, "scatter_rays": 1, "tile": TILE,
                       "parallelism": f"tiles{world}" if world > 1 else "single",
                       "frame_per_gpu": f"{base_wh[0]}x{base_wh[1]}",
                       "scene": "reference main.rs scene: bundled bunny.ply x3 instances + floor cuboid + skybox.jpg"},
            "rays_per_step": {"closest": closest_per_step, "any": any_per_step, "paths": paths_per_step},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
