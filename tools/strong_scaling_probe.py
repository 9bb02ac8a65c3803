#!/usr/bin/env python3
"""What one rank of an N-rank tile-sharded render takes on this GPU: renders the tile set of rank 0 of N (tiles 0, N, 2N, ...)
of a fixed frame for N = 1, 2, 4, 8 and prints the time next to the ideal 1/N.  No collective is involved (every rank's
render is independent; the gather of 12 B per pixel is the only exchange), so this is the compute side of strong scaling.

    python tools/strong_scaling_probe.py [--scene bunny|spheres|grid] [--cam WxH] [--aa 8] [--depth 5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bunny", choices=["bunny", "spheres", "grid"])
    ap.add_argument("--cam", default="1080x960")
    ap.add_argument("--aa", type=int, default=8)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--tile", type=int, default=40)
    ap.add_argument("--opt", default="", help="tuning overrides, e.g. static=0,chunk=128,streams=2,max_paths=1048576")
    args = ap.parse_args()
    rt = entry.load_package()
    w, h = map(int, args.cam.split("x"))
    sky = rt.load_skybox(max_width=2048)
    make = {"bunny": lambda: rt.Scene.main(sky=sky, cam=(w, h)), "spheres": lambda: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=(w, h)),
            "grid": lambda: rt.Scene.grid(230, 230, sky=sky, cam=(w, h))}
    scene = make[args.scene]().upload(0)
    opt_ids = {"chunk": rt.OPT_CHUNK, "blocks": rt.OPT_BLOCKS_PER_CU, "lds": rt.OPT_LDS_SCENE, "max_paths": rt.OPT_MAX_PATHS, "streams": rt.OPT_STREAMS,
               "static": rt.OPT_STATIC_SHARE, "min_batch": rt.OPT_MIN_BATCH}
    for kv in filter(None, args.opt.split(",")):
        k, v = kv.split("=")
        scene.set_option(opt_ids[k], int(v))
    base = None
    for n in [int(v) for v in args.ranks.split(',')]:
        region = rt.Scene.make_region(w, h, tile=(args.tile, args.tile), tile_first=0, tile_stride=n)
        scene.render(max_depth=args.depth, shadow_rays=1, antialiasing=1, region=region)
        best = None
        for _ in range(args.reps):
            _, st = scene.render(max_depth=args.depth, shadow_rays=1, antialiasing=args.aa, region=region)
            if best is None or st["device_ms"] < best["device_ms"]:
                best = st
        rays = best["closest_rays"] + best["any_rays"]
        scene.set_option(rt.OPT_PROFILE, 1)
        _, sp = scene.render(max_depth=args.depth, shadow_rays=1, antialiasing=args.aa, region=region)
        scene.set_option(rt.OPT_PROFILE, 0)
        stages = f"trace {sp['trace_closest_ms'] + sp['trace_any_ms']:.1f} ms, shade + resolve {sp['shade_ms']:.1f} ms, launches {sp['launches']}"
        if base is None:
            base = best["device_ms"]
        print(f"{args.scene} {w}x{h} aa {args.aa} {args.opt}: rank 0 of {n}: {best['device_ms']:9.2f} ms  {rays:>12d} rays  {rays / best['device_ms'] / 1e3:8.1f} Mrays/s per GPU"
              f"  speed-up {base / best['device_ms']:5.2f} of {n} (efficiency {100 * base / best['device_ms'] / n:5.1f} %)  [{stages}]", flush=True)


if __name__ == "__main__":
    main()
