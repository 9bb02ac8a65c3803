#!/usr/bin/env python3
"""The frame kernel (RTAMD_OPT_FRAME_KERNEL) against the wavefront pipeline on the main scene: pixels and ray counts must be
identical; then the rate of both at the headline size.

    python tools/frame_kernel_check.py [--cam 256x192] [--aa 2] [--depth 5] [--time 1080x960:8] [--step-min 64]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cam", default="256x192")
    ap.add_argument("--aa", type=int, default=2)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--time", default="1080x960:8", help="WxH:aa of the timed frame ('' = none)")
    ap.add_argument("--step-min", default="64")
    ap.add_argument("--rebalance", default="1", help="comma list of RTAMD_OPT_FRAME_REBALANCE values for the timed frame")
    ap.add_argument("--blocks", type=int, default=0, help="RTAMD_OPT_BLOCKS_PER_CU for the timed frames of the frame kernel")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--ranks", type=int, default=1, help="render the tiles rank 0 of this many would own (needs --tile)")
    ap.add_argument("--scatter", type=int, default=1)
    ap.add_argument("--scene", default="main", choices=["main", "spheres", "grid", "textured"])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    rt = entry.load_package()
    sky = rt.load_skybox(max_width=512)
    w, h = map(int, args.cam.split("x"))
    make = {"main": lambda cam: rt.Scene.main(sky=sky, cam=cam), "spheres": lambda cam: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=cam),
            "grid": lambda cam: rt.Scene.grid(60, 60, sky=sky, cam=cam), "textured": lambda cam: rt.Scene.textured(sky=sky, cam=cam)}[args.scene]
    gpu = make((w, h)).upload(0)
    params = dict(max_depth=args.depth, shadow_rays=1, antialiasing=args.aa, scatter_rays=args.scatter)
    gpu.set_option(rt.OPT_FRAME_KERNEL, 0)
    a, sa = gpu.render(**params)
    print("pipeline rendered", flush=True)
    gpu.set_option(rt.OPT_FRAME_KERNEL, 1)
    for reb in (0, 1, 2):
        gpu.set_option(rt.OPT_FRAME_REBALANCE, reb)
        b, sb = gpu.render(**params)
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        print(f"{w}x{h} aa {args.aa} depth {args.depth} rebalance {reb}: identical {same}; max diff {d.max():.3e}, pixels differing {(d.max(axis=-1) > 0).sum()} of {w * h}; "
              f"rays pipeline {sa['closest_rays']} + {sa['any_rays']}, frame kernel {sb['closest_rays']} + {sb['any_rays']}; launches {sa['launches']} / {sb['launches']}", flush=True)
    if not args.time:
        return
    cam, aa = args.time.split(":")
    w, h = map(int, cam.split("x"))
    aa = int(aa)
    gpu = make((w, h)).upload(0)
    params = dict(max_depth=args.depth, shadow_rays=1, antialiasing=aa, scatter_rays=args.scatter)
    if args.tile:
        params["region"] = rt.Scene.make_region(w, h, tile=(args.tile, args.tile), tile_first=0, tile_stride=args.ranks)
    ref = None
    for mode in [0] + [1] * len(args.step_min.split(",")):
        pass
    runs = [(0, 64, 1)] + [(1, int(s), int(r)) for s in args.step_min.split(",") for r in args.rebalance.split(",")]
    for fk, smin, reb in runs:
        gpu.set_option(rt.OPT_FRAME_KERNEL, fk)
        gpu.set_option(rt.OPT_BLOCKS_PER_CU, args.blocks if (fk and args.blocks) else 2)
        gpu.set_option(rt.OPT_FRAME_STEP_MIN, smin)
        gpu.set_option(rt.OPT_FRAME_REBALANCE, reb)
        img, st = gpu.render(**params)
        best = 1e30
        for _ in range(args.reps):
            t0 = time.perf_counter()
            img, st = gpu.render(**params)
            best = min(best, time.perf_counter() - t0)
        rays = st["closest_rays"] + st["any_rays"]
        if ref is None:
            ref = img
        print(f"{w}x{h} {aa * aa} spp frame_kernel={fk} step_min={smin} rebalance={reb}: device {st['device_ms']:.2f} ms, {rays / st['device_ms'] / 1e3:.1f} Mrays/s (host {best * 1e3:.1f} ms), "
              f"rays {rays}, identical to pipeline {np.array_equal(ref.view(np.uint32), img.view(np.uint32))}", flush=True)


if __name__ == "__main__":
    main()
