#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Default workload (BASELINE.json configs[2], the one the metric "primary + diffuse bounce" is quoted on): the reference's
bunny scene (3 instanced bunnies + glossy floor + the real skybox) at 1080x960, antialiasing 8 (64 spp), max_depth 5,
shadow_rays 1, scatter_rays 1.  One step = one full frame through the wavefront kernels (ray generation, closest-hit
trace, shade, shadow trace, compaction), scene and skybox resident in HBM.
Metric: Mrays/s = (closest-hit + any-hit rays traced) / wall time, whole job.

Other workloads: the 4K primary-ray batch (configs[1]), 10 000 spheres in an object tree at 3840x2160 / 64 spp
(configs[3]) and the 230 x 230 instanced-bunny grid (50.1 M triangles) at 7680x4320 / 256 spp (configs[4]).

N > 1: image tiles are the units that shard (40x40 pixels, dealt round-robin to the ranks, one rank per GPU, scene
replicated; no data-path collective besides the framebuffer gather to rank 0, one RCCL gather inside the timed region).
Default: weak scaling -- the frame grows with N (N x the pixels of the workload's frame at the same camera and aspect
ratio, sides rounded to whole tiles), so every rank renders as many pixels of the same picture as the single GPU does.
`--strong` cuts the workload's own frame into N shares instead (total work fixed).  `config` says which.
`python bench.py --gpus N` from a plain shell starts its own N ranks (torch.distributed.run, 127.0.0.1) before anything
touches the GPU; under torch.distributed.run it is one of the ranks.  On a box with fewer GPUs than ranks the ranks
share devices and gather through gloo (a rehearsal).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

# name: scene constructor, camera, antialiasing, max_depth, shadow_rays, BASELINE.json configs[] index
WORKLOADS = {
    "bunny_1080x960_64spp_depth5": dict(scene=("main",), cam=(1080, 960), aa=8, max_depth=5, shadow_rays=1, baseline_config=2,
                                        desc="reference main.rs scene: bundled bunny.ply x3 instances + floor cuboid + skybox.jpg"),
    "bunny_3840x2160_primary": dict(scene=("main",), cam=(3840, 2160), aa=1, max_depth=1, shadow_rays=1, baseline_config=1,
                                    desc="reference main.rs scene, primary rays and their shadow rays only"),
    "spheres_10k_3840x2160_64spp_depth5": dict(scene=("spheres", 10000, 1), cam=(3840, 2160), aa=8, max_depth=5, shadow_rays=1, baseline_config=3,
                                               desc="10 000 spheres (splitmix64 seed 1) in one AABBTree<Object> over a floor cuboid + skybox.jpg"),
    "bunny_grid_230x230_7680x4320_256spp_depth5": dict(scene=("grid", 230, 230), cam=(7680, 4320), aa=16, max_depth=5, shadow_rays=1, baseline_config=4,
                                                       desc="230 x 230 instances of the bundled bunny (50.1 M triangles) in one AABBTree<Object>, instanced (one mesh tree, "
                                                            "per-instance ray transform) + floor cuboid + skybox.jpg"),
    "bunny_grid_flat_230x230_7680x4320_256spp_depth5": dict(scene=("grid_flat", 230, 230), cam=(7680, 4320), aa=16, max_depth=5, shadow_rays=1, baseline_config=4,
                                                            desc="the same 230 x 230 grid flattened to world space: 52 900 separate meshes (50.1 M world-space triangles, "
                                                                 "one tree each) in one AABBTree<Object> + floor cuboid + skybox.jpg"),
    "bunny_grid_merged_230x230_7680x4320_256spp_depth5": dict(scene=("grid_merged", 230, 230), cam=(7680, 4320), aa=16, max_depth=5, shadow_rays=1, baseline_config=4,
                                                              desc="the same 230 x 230 grid flattened to world space and merged by material: three meshes of 16.7 M world-space "
                                                                   "triangles each (trees built on the GPU) + floor cuboid in a plain list + skybox.jpg"),
}
DEFAULT_WORKLOAD = "bunny_1080x960_64spp_depth5"
TILE = 40                        # divides every workload's frame (1080x960: 27 x 24 tiles, 81 per rank of 8; no partial tiles: 64-pixel tiles leave 0.7 % of the pixel slots empty)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak
N_SIMD = 1024                    # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4
VALU_PEAK_GINST = N_SIMD * CLOCK_GHZ / 2.0   # one wave64 VALU instruction per SIMD every 2 cycles: 1228.8 G wave-instructions/s
VALU_MEASURED_GINST = 856.0      # what this chip sustains on independent v_fma_f32 with >= 2 waves per SIMD (profiles/r02/valu_issue_probe.txt)
N_CU = 256
LDS_PEAK_TBS = 150.0             # every CU streaming ds_read_b64 / b128: 256 B per clock per CU
L2_PEAK_GBS = 34500.0            # aggregate L2 bandwidth
TRACE_KERNEL = "k_wave_trace"


def cpu_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env))
    return max(1, min(n, 64))


def make_scene(mod, spec, sky, cam, gpu_build=False):
    kind = spec[0]
    if kind == "main":
        return mod.main(sky=sky, cam=cam)
    if kind == "spheres":
        return mod.spheres(n=spec[1], seed=spec[2], sky=sky, cam=cam)
    if kind == "grid":
        return mod.grid(spec[1], spec[2], sky=sky, cam=cam)
    if kind == "grid_flat":
        return mod.grid(spec[1], spec[2], sky=sky, cam=cam, flat=1)
    if kind == "grid_merged":   # the product builds the three 16.7 M-triangle trees on the GPU, the oracle (bounded sample only) on the host
        return mod.grid(spec[1], spec[2], sky=sky, cam=cam, flat=3 if gpu_build else 2)
    raise SystemExit(f"unknown scene {spec}")


def cpu_baseline(wl, sky, cam, aa):
    """Times the CPU oracle (C++ restatement of the reference renderer) on a bounded sample of the workload and returns
    its throughput plus the reference-order visit counters used for the algorithmic byte model."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    scene = make_scene(ol.OracleScene, wl["scene"], sky, cam)
    threads = cpu_threads()
    # bounded sample: a centred rectangle of at most 1080 x 960 pixels at a sample count that gives about 50 M paths
    # (10 - 30 s on 16 host cores), never more than the workload's own
    rw, rh = min(cam[0], 1080), min(cam[1], 960)
    rect = ((cam[0] - rw) // 2, (cam[1] - rh) // 2, rw, rh)
    sample_aa = max(1, min(aa, int((50e6 / (rw * rh)) ** 0.5)))
    _, sec, cnt = scene.render(max_depth=wl["max_depth"], shadow_rays=wl["shadow_rays"], antialiasing=sample_aa, rect=rect, threads=threads)
    rays = cnt["closest_rays"] + cnt["any_rays"]
    base = {
        "value": rays / sec / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"centred {rw}x{rh} rectangle of the {cam[0]}x{cam[1]} frame at antialiasing {sample_aa} ({sample_aa * sample_aa} spp) instead of "
                  f"{aa * aa} spp, same scene, depth and lights: {rays} rays in {sec:.2f} s, 100-pixel chunks over {threads} threads",
    }
    return base, cnt


def byte_model(cnt):
    """Algorithmic bytes per ray from the oracle's reference-order visit counters:
    closest ray: 32 B in + 16 B out + 32 B per node visited + 36 B per triangle tested + 16 B per sphere + 36 B per cuboid
    test; any-hit ray: 32 B in + 4 B out + the same traversal terms."""
    nc = cnt["nodes_visited"] - cnt["nodes_visited_any"]
    tc = cnt["tris_tested"] - cnt["tris_tested_any"]
    sc = cnt["spheres_tested"] - cnt["spheres_tested_any"]
    cc = cnt["cuboids_tested"] - cnt["cuboids_tested_any"]
    closest = 48.0 + (32.0 * nc + 36.0 * tc + 16.0 * sc + 36.0 * cc) / max(1, cnt["closest_rays"])
    anyb = 36.0 + (32.0 * cnt["nodes_visited_any"] + 36.0 * cnt["tris_tested_any"] + 16.0 * cnt["spheres_tested_any"]
                   + 36.0 * cnt["cuboids_tested_any"]) / max(1, cnt["any_rays"])
    return closest, anyb


# ---- counter passes: this program again, as a child of rocprofv3, one --pmc pass per run ----------------------------
PMC_PASSES = (
    ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY"]),
    ("fetch", ["FETCH_SIZE", "TCC_HIT_sum"]),
    ("write", ["WRITE_SIZE", "TCC_MISS_sum", "TCC_REQ_sum"]),
    ("lds", ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS"]),
)


def run_pmc_passes(argv_inner, timeout_s=150):
    """Returns {counter: (sum over the trace kernel's dispatches, dispatches)} or None when rocprofv3 is unavailable."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = {}
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    for name, counters in PMC_PASSES:
        d = tempfile.mkdtemp(prefix=f"rtamd_pmc_{name}_", dir="/tmp")
        cmd = ["timeout", "-s", "KILL", str(timeout_s), exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__), "--inner"] + argv_inner
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s + 30)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                continue   # (this pass's figures stay null; the others are kept)
            seen = {}
            for r in csv.DictReader(open(files[0])):
                if TRACE_KERNEL in r["Kernel_Name"]:
                    s, disp = seen.setdefault(r["Counter_Name"], [0.0, set()])
                    seen[r["Counter_Name"]][0] = s + float(r["Counter_Value"])
                    disp.add(r["Dispatch_Id"])
            for k, (s, disp) in seen.items():
                out[k] = (s, len(disp))
        except Exception:
            continue
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out or None


def self_launch(n):
    """`python bench.py --gpus N` from a plain shell: start N ranks as children (this process never replaces itself
    with another program, so it does not matter whether counting the devices initialises the GPU runtime)."""
    import torch
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < n:
        env.setdefault("RTAMD_BENCH_BACKEND", "gloo")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout is relayed as one JSON line (rank 0's); anything else a library prints there (gloo's connection
    # messages) goes to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line_out = None
    for line in proc.stdout:
        t = line.strip()
        if t.startswith("{") and t.endswith("}"):
            try:
                json.loads(t)
                line_out = t
                continue
            except ValueError:
                pass
        sys.stderr.write(line)
    rc = proc.wait()
    if line_out is not None:
        print(line_out, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--aa", type=int, default=0, help="override the workload's antialiasing (spp = aa^2); recorded in config")
    ap.add_argument("--cam", default="", help="override the workload's frame, WxH; recorded in config")
    ap.add_argument("--weak", action="store_true", help="N > 1: grow the frame with N (fixed work per GPU); the default")
    ap.add_argument("--strong", action="store_true", help="N > 1: split the workload's own frame over the ranks (fixed total work) instead of growing it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic and the VALU count stay null)")
    ap.add_argument("--small-sky", action="store_true", help="use a strided skybox (debug)")
    ap.add_argument("--opt", default="", help="tuning overrides, e.g. chunk=128,blocks=3,lds=0,max_paths=2097152")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)   # child of a rocprofv3 counter pass: render only
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the render path)")
    # RTAMD_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # the gather goes through host memory); the driver's runs use the default, RCCL with one GPU per rank.
    # (also chosen by itself when the launcher starts more local ranks than the box has GPUs)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    backend = os.environ.get("RTAMD_BENCH_BACKEND") or ("nccl" if torch.cuda.device_count() >= local_world else "gloo")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # what the communication libraries print while they connect (gloo: "[Gloo] Rank ... is connected ...") goes to
        # stderr: stdout carries the one JSON line
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)

    rt = entry.load_package()
    from rust_raytracer_amd import dist_tiles

    wl = WORKLOADS[args.workload]
    aa = args.aa or wl["aa"]
    max_depth, shadow_rays = wl["max_depth"], wl["shadow_rays"]
    base_wh = tuple(int(v) for v in args.cam.split("x")) if args.cam else wl["cam"]
    cam_wh = base_wh
    weak = world > 1 and not args.strong
    if weak:  # world x the pixels, same camera and aspect ratio, whole tiles
        k = world ** 0.5
        cam_wh = (max(1, int(round(base_wh[0] * k / TILE))) * TILE, max(1, int(round(base_wh[1] * k / TILE))) * TILE)
    sky = rt.load_skybox(max_width=1024 if args.small_sky else None)
    t0 = time.perf_counter()
    scene = make_scene(rt.Scene, wl["scene"], sky, cam_wh, gpu_build=True)
    host_build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    scene.upload(dev_index)
    upload_s = time.perf_counter() - t0
    opt_ids = {"chunk": rt.OPT_CHUNK, "blocks": rt.OPT_BLOCKS_PER_CU, "lds": rt.OPT_LDS_SCENE, "max_paths": rt.OPT_MAX_PATHS, "streams": rt.OPT_STREAMS, "static": rt.OPT_STATIC_SHARE, "min_batch": rt.OPT_MIN_BATCH, "lean": rt.OPT_LEAN_KERNELS,
               "ordered": rt.OPT_ORDERED_WALK, "deepen": rt.OPT_TREE_DEEPENING, "ostack": rt.OPT_ORDERED_STACK,
               "frame": rt.OPT_FRAME_KERNEL, "rebalance": rt.OPT_FRAME_REBALANCE, "step_min": rt.OPT_FRAME_STEP_MIN}
    for kv in filter(None, args.opt.split(",")):
        k, v = kv.split("=")
        scene.set_option(opt_ids[k], int(v))
    W, H = cam_wh
    region = rt.Scene.make_region(W, H, tile=(TILE, TILE), tile_first=rank, tile_stride=world)
    n_px = rt.Scene.region_pixels(region)
    tiles = torch.zeros((n_px // (TILE * TILE), TILE, TILE, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    split = {"render_s": 0.0, "gather_s": 0.0}   # this rank's share of the timed region (N > 1: reported per rank)

    def step():
        t_a = time.perf_counter()
        _, st = scene.render(max_depth=max_depth, shadow_rays=shadow_rays, antialiasing=aa, region=region, out_device=tiles)   # returns when the frame is complete
        t_b = time.perf_counter()
        img = dist_tiles.gather_frame(tiles, W, H, TILE, TILE, rank, world)
        if world > 1:
            torch.cuda.synchronize()
        split["render_s"] += t_b - t_a
        split["gather_s"] += time.perf_counter() - t_b
        return st, img

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.inner:   # counter pass: the frame itself, nothing else
        for _ in range(max(1, args.steps)):
            step()
        torch.cuda.synchronize()
        return

    for _ in range(args.warmup):
        step()
    barrier()
    split["render_s"] = split["gather_s"] = 0.0
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

    per_rank = None
    if world > 1:
        mine = {"rank": rank, "device": dev_index, "owned_tiles": n_px // (TILE * TILE), "render_ms_per_step": split["render_s"] / args.steps * 1e3,
                "gather_ms_per_step": split["gather_s"] / args.steps * 1e3, "rays_per_step": float(st["closest_rays"] + st["any_rays"])}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([elapsed, float(rays), float(closest_per_step), float(any_per_step), float(paths_per_step)], dtype=torch.float64,
                         device=dev if backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays = float(t[1])
        closest_per_step, any_per_step, paths_per_step = float(t[2]), float(t[3]), float(t[4])

    # ---- rank 0: bound of the dominant kernel (HIP events on the kernels' own stream), counters, CPU baseline ----
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
            cpu, cnt = cpu_baseline(wl, sky, cam_wh, aa)
        pmc = None
        if not args.no_pmc and world == 1:
            inner = ["--workload", args.workload, "--steps", "1", "--warmup", "0", "--aa", str(aa), "--cam", f"{W}x{H}"]
            if args.opt:
                inner += ["--opt", args.opt]
            if args.small_sky:
                inner += ["--small-sky"]
            pmc = run_pmc_passes(inner)
        if kt["closest_launches"] + kt["any_launches"]:
            # k_wave_trace carries the closest-hit rays of a bounce plus the shadow rays of the previous one; the
            # launch that ends a sample carries shadow rays only.  All its launches are pooled.
            n_launch = kt["closest_launches"] + kt["any_launches"]
            trace_ms = kt["closest_ms"] + kt["any_ms"]
            avg_ms = trace_ms / n_launch
            n_rays = st["closest_rays"] + st["any_rays"]
            rays_per_launch = n_rays / n_launch
            rays_per_s = n_rays / (trace_ms * 1e-3)
            # The scene's nodes and triangles are served from LDS / L2, so the kernel is bound by instruction issue and
            # LDS / L2 latency, not by HBM: `achieved` counts the VALU wave-instructions it issues per second (rocprofv3
            # SQ_INSTS_VALU of one frame / its rays, times the live ray rate) against one instruction per SIMD per 2 cycles.
            valu_per_ray = hbm_bytes = lane_util = l2_hit = None
            if pmc and "SQ_INSTS_VALU" in pmc:
                valu_per_ray = pmc["SQ_INSTS_VALU"][0] / n_rays
                if pmc.get("SQ_ACTIVE_INST_VALU", (0, 0))[0]:
                    lane_util = pmc["SQ_THREAD_CYCLES_VALU"][0] / (pmc["SQ_ACTIVE_INST_VALU"][0] * 64.0)
            if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                # KiB; gfx950 tallies a 128-byte read request at 64 bytes: the read side is doubled (an upper bound for narrow reads)
                hbm_bytes = (2.0 * pmc["FETCH_SIZE"][0] * 1024.0 / pmc["FETCH_SIZE"][1]) + (pmc["WRITE_SIZE"][0] * 1024.0 / pmc["WRITE_SIZE"][1])
            if pmc and "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
                l2_hit = pmc["TCC_HIT_sum"][0] / max(1.0, pmc["TCC_HIT_sum"][0] + pmc["TCC_MISS_sum"][0])
            achieved = valu_per_ray * rays_per_s / 1e9 if valu_per_ray else None
            # the levels the nodes and triangles are actually served from: LDS-array cycles (every cycle moves up to 256 B per CU)
            # against the CUs' cycles of the launch, and L2 requests (at most 128 B each) against the L2's aggregate bandwidth
            lds = l2 = None
            if pmc and "SQ_LDS_IDX_ACTIVE" in pmc:
                cu_cycles = N_CU * CLOCK_GHZ * 1e9 * trace_ms * 1e-3
                idx = pmc["SQ_LDS_IDX_ACTIVE"][0]
                lds = {"array_busy_frac": idx / cu_cycles, "bank_conflict_share_of_array_cycles": pmc["SQ_LDS_BANK_CONFLICT"][0] / max(1.0, idx),
                       "array_cycles_per_ray": idx / n_rays, "lds_wave_insts_per_ray": pmc["SQ_INSTS_LDS"][0] / n_rays,
                       "bytes_upper_tbs": idx * 256.0 / (trace_ms * 1e-3) / 1e12, "peak_tbs": LDS_PEAK_TBS,
                       "note": "SQ_LDS_IDX_ACTIVE = LDS-array cycles, each serving up to 256 B; busy = those cycles / (256 CUs x 2.4 GHz x kernel time)"}
            if pmc and "TCC_REQ_sum" in pmc:
                req_gbs = pmc["TCC_REQ_sum"][0] * 128.0 / (trace_ms * 1e-3) / 1e9
                l2 = {"requests_per_ray": pmc["TCC_REQ_sum"][0] / n_rays, "gbs_upper": req_gbs, "peak_gbs": L2_PEAK_GBS, "frac_upper": req_gbs / L2_PEAK_GBS,
                      "note": "TCC_REQ_sum x 128 B (the most a request moves) / kernel time"}
            roofline = {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_GINST, "unit": "Ginst/s",
                        "frac": achieved / VALU_PEAK_GINST if achieved else None, "traffic": hbm_bytes,
                        "peak_measured": VALU_MEASURED_GINST, "frac_of_measured_issue_ceiling": achieved / VALU_MEASURED_GINST if achieved else None,
                        "lds": lds, "l2": l2,
                        "kernel": "k_wave_trace (persistent closest-hit + any-hit traversal of the wavefront queues)",
                        "avg_launch_ms": avg_ms, "launches": n_launch, "rays_per_launch": rays_per_launch,
                        "valu_wave_insts_per_ray": valu_per_ray, "valu_lane_utilisation": lane_util,
                        "note": "bound = vector-instruction issue + LDS/L2 latency (tree and triangles are LDS- or L2-resident); "
                                "peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (peak_measured: what independent v_fma_f32 "
                                "streams sustain on this chip); hbm_* = measured HBM traffic of the same kernel against the 8 TB/s peak; lds / l2 = "
                                "the levels the tree is served from; algorithmic_* = the reference-order byte model (bytes the reference's walk "
                                "would touch per ray; served from LDS / L2 here, so its rate is not bounded by the HBM peak)",
                        "hbm_gbs": hbm_bytes / (avg_ms * 1e-3) / 1e9 if hbm_bytes else None,
                        "hbm_frac": hbm_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if hbm_bytes else None,
                        "l2_hit_rate": l2_hit,
                        "stream_only_gbs": (st["closest_rays"] * 48.0 + st["any_rays"] * 36.0) / (trace_ms * 1e-3) / 1e9}
            if cnt is not None:
                b_closest, b_any = byte_model(cnt)
                alg = (st["closest_rays"] * b_closest + st["any_rays"] * b_any) / (trace_ms * 1e-3) / 1e9
                roofline.update({"algorithmic_bytes_per_closest_ray": b_closest, "algorithmic_bytes_per_any_ray": b_any, "algorithmic_gbs": alg})
            extra["trace_kernel_mrays_per_s"] = rays_per_s / 1e6
            extra["stage_ms_per_frame"] = {"trace": trace_ms, "shade_resolve": st["shade_ms"], "device_total": st["device_ms"]}

    if rank == 0:
        value = rays / elapsed / 1e6
        reduced = (aa != wl["aa"]) or (base_wh != wl["cam"])
        line = {
            "metric": "Mrays/s (primary + diffuse bounce) on bunny BVH", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "reference assets (bunny.ply, skybox.jpg) + generated instance / sphere placement" if wl["scene"][0] != "main"
                    else "reference assets (bunny.ply, skybox.jpg)",
            "config": {"workload": args.workload, "baseline_config": wl["baseline_config"], "width": W, "height": H, "spp": aa * aa, "max_depth": max_depth,
                       "shadow_rays": shadow_rays, "scatter_rays": 1, "tile": TILE,
                       "parallelism": (f"tiles{world} of one {W}x{H} frame" if not weak else f"tiles{world} of a {W}x{H} frame: {base_wh[0]}x{base_wh[1]} pixels' worth per GPU") if world > 1 else "single",
                       "backend": ("rccl" if backend == "nccl" else backend + " (ranks share devices: rehearsal)") if world > 1 else None,
                       "reduced_from_workload": reduced, "scene": wl["desc"], "host_build_s": host_build_s, "upload_s": upload_s},
            "rays_per_step": {"closest": closest_per_step, "any": any_per_step, "paths": paths_per_step},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if per_rank is not None:
            r_ms = [p["render_ms_per_step"] for p in per_rank]
            g_ms = [p["gather_ms_per_step"] for p in per_rank]
            line["multi_gpu"] = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "devices": sorted({p["device"] for p in per_rank}),
                                 "per_rank_render_ms": {"min": min(r_ms), "max": max(r_ms)}, "gather_ms": {"min": min(g_ms), "max": max(g_ms), "rank0": g_ms[0]},
                                 "owned_tiles": [p["owned_tiles"] for p in per_rank], "rays_per_rank": [p["rays_per_step"] for p in per_rank],
                                 "note": "render = this rank's tiles through the wavefront kernels (blocking); gather = the framebuffer gather to rank 0 "
                                         "(a rank that finishes early waits here for the slowest); ms_per_step is the slowest rank's total"}
        line["config"]["strong_scaling_workloads"] = ("with --strong: spheres_10k_3840x2160_64spp_depth5 and bunny_grid_230x230_7680x4320_256spp_depth5 (BASELINE configs[3], [4]) are the "
                                                      "fixed frames meant for 2 / 4 / 8 GPUs: a pixel's samples are sequential (one random stream per pixel, integrator.rs:36), so a "
                                                      "frame is spp x max_depth dependent launch steps whatever the tile count, and the 1080x960 frame cut in eight runs "
                                                      "every step at its latency floor")
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
