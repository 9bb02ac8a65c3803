#!/usr/bin/env python3
"""Per-phase breakdown of the persistent scheduler kernels (diagnostic build, `make -C rust-raytracer_amd/csrc diag`).

Prints, for FETCH / OBJ / TNODE / NODE / TRI: share of wave cycles, steps executed, mean active lanes per step.
Run on the GPU box:  RTAMD_LIB=rust-raytracer_amd/librtamd_diag.so python tools/phase_diag.py [--aa 2] [--depth 5]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTAMD_LIB", os.path.join(ROOT, "rust-raytracer_amd", "librtamd_diag.so"))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aa", type=int, default=2)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--cam", default="1080x960")
    ap.add_argument("--opt", default="")
    ap.add_argument("--scene", default="bunny", choices=["bunny", "spheres", "grid", "terrain", "merged", "flat"])
    ap.add_argument("--grid", type=int, default=230, help="instances per side of the grid scenes")
    ap.add_argument("--tile", type=int, default=0, help="render as square tiles of this size (slot order is tile-major)")
    ap.add_argument("--scatter", type=int, default=1)
    ap.add_argument("--frame", action="store_true", help="the frame kernel (RTAMD_OPT_FRAME_KERNEL) instead of the pipeline")
    args = ap.parse_args()
    rt = entry.load_package()
    w, h = map(int, args.cam.split("x"))
    sky = rt.load_skybox(max_width=2048)
    make = {"bunny": lambda: rt.Scene.main(sky=sky, cam=(w, h)), "spheres": lambda: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=(w, h)),
            "grid": lambda: rt.Scene.grid(args.grid, args.grid, sky=sky, cam=(w, h)),
            "merged": lambda: rt.Scene.grid(args.grid, args.grid, sky=sky, cam=(w, h), flat=3), "flat": lambda: rt.Scene.grid(args.grid, args.grid, sky=sky, cam=(w, h), flat=1), "terrain": lambda: rt.Scene.terrain(300, seed=1, with_bunny=True, sky=sky, cam=(w, h))}
    scene = make[args.scene]().upload(0)
    region = rt.Scene.make_region(w, h, tile=(args.tile, args.tile)) if args.tile else None
    opt_ids = {"chunk": rt.OPT_CHUNK, "blocks": rt.OPT_BLOCKS_PER_CU, "lds": rt.OPT_LDS_SCENE, "streams": rt.OPT_STREAMS, "static": rt.OPT_STATIC_SHARE, "min_batch": rt.OPT_MIN_BATCH,
               "ordered": rt.OPT_ORDERED_WALK, "deepen": rt.OPT_TREE_DEEPENING, "step_min": rt.OPT_FRAME_STEP_MIN, "rebalance": rt.OPT_FRAME_REBALANCE}
    if args.frame:
        scene.set_option(rt.OPT_FRAME_KERNEL, 1)
    for kv in filter(None, args.opt.split(",")):
        k, v = kv.split("=")
        scene.set_option(opt_ids[k], int(v))
    lib = rt.rtamd_lib()
    lib.rtamd_diag_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    buf = (C.c_uint64 * 48)()
    if args.frame:
        lib.rtamd_diag_read_frame.argtypes = lib.rtamd_diag_read.argtypes
        lib.rtamd_diag_read = lib.rtamd_diag_read_frame
    lib.rtamd_diag_read_shade.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    sbuf = (C.c_uint64 * 32)()
    for depth in sorted({1, args.depth}):
        scene.render(max_depth=depth, antialiasing=1)  # warm up
        lib.rtamd_diag_read(scene.ctx, buf, 1)
        lib.rtamd_diag_read_shade(scene.ctx, sbuf, 1)
        _, st = scene.render(max_depth=depth, shadow_rays=1, antialiasing=args.aa, region=region, scatter_rays=args.scatter)
        lib.rtamd_diag_read(scene.ctx, buf, 1)
        lib.rtamd_diag_read_shade(scene.ctx, sbuf, 1)
        d = np.array(list(buf), dtype=np.float64)[:15].reshape(5, 3)
        rays = st["closest_rays"] + st["any_rays"]
        total = d[:, 0].sum()
        print(f"max_depth {depth}: rays {rays} ({st['closest_rays']} closest, {st['any_rays']} any), device {st['device_ms']:.2f} ms")
        for name, (cyc, steps, lanes) in zip(("FETCH", "OBJ", "TNODE", "NODE", "TRI"), d):  # noqa: E501
            if steps:
                print(f"  {name:6s} cycles {100 * cyc / total:5.1f}%  steps {int(steps):>12d}  lanes/step {lanes / steps:5.1f}  "
                      f"cycles/step {cyc / steps:7.1f}  lane-steps/ray {lanes / rays:6.2f}")
        sub = np.array(list(buf), dtype=np.float64)[15:31].reshape(8, 2)
        for name, (cyc, cnt) in zip(("fetch: hand-out (wait + reads of the staged batch)", "fetch: queue-head atomic", "fetch: result stores", "obj: fold + next object", "obj: record + wrappers", "obj: mesh entry + root box", "fetch: decode + ray setup", "fetch: next batch (deal, claim, addresses; per request)"), sub):
            if cnt:
                print(f"  {name:30s} {100 * cyc / total:5.1f}% of cycles  n {int(cnt):>10d}  cycles each {cyc / cnt:8.1f}")
        if args.frame:
            waves, life_sum, life_max, steps, pix, step_ticks, fetches, handed, asked, unserved = (float(x) for x in list(buf)[32:42])
            print(f"  frame kernel: {int(waves)} waves, lifetime mean {life_sum / waves / 100:.0f} us, max {life_max / 100:.0f} us; integrator steps {int(steps)} "
                  f"({pix / max(steps, 1):.1f} pixels each, {step_ticks / max(steps, 1) / 100:.1f} us each, {100 * step_ticks / life_sum:.1f}% of the waves' time); "
                  f"FETCH steps {int(fetches)} ({handed / max(fetches, 1):.1f} rays handed out of {asked / max(fetches, 1):.1f} asked for, {unserved / max(fetches, 1):.1f} lanes left without)")
            t_scan, t_step, t_hand, t_load = (float(x) for x in list(buf)[42:46])
            print(f"  FETCH step, share of the waves' time: first scan {100 * t_scan / life_sum:.1f}%, integrator step with lists and second scan {100 * t_step / life_sum:.1f}%, "
                  f"hand-out {100 * t_hand / life_sum:.1f}%, ray loads + result stores + ray setup {100 * t_load / life_sum:.1f}%")
        sh = np.array(list(sbuf), dtype=np.float64)
        names = {0: "loads + hit reconstruction", 2: "lit-queue append", 3: "light samples", 5: "scatter", 7: "path-queue append"}
        if sh[16:].sum():
            print("  shade kernel, first thread of the first block, microseconds per launch:")
            for k, name in names.items():
                if sh[16 + k]:
                    print(f"    {name:28s} {sh[k] / sh[16 + k] / 100.0:7.2f} us  (launches {int(sh[16 + k])})")


if __name__ == "__main__":
    main()
