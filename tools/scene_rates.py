#!/usr/bin/env python3
"""Whole-frame Mrays/s of the synthetic scenes next to the bunny scene (one GPU).

    python tools/scene_rates.py [--spp-root 4]

Scenes: bunny (reference main.rs), 10 000 spheres in an object tree, 230 x 230 instanced bunnies in an object tree
(50.1 M triangles), a 300 x 300 heightfield with two bunnies (mesh trees in global memory and LDS mixed).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp-root", type=int, default=4)
    ap.add_argument("--cam", default="1080x960")
    ap.add_argument("--scenes", default="", help="comma-separated subset of the scene names")
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--scatter", type=int, default=1, help="scatter_rays (other than 1: the depth-first kernel)")
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--df-rounds", type=int, default=1, help="scatter_rays != 1: 1 round pipeline, 0 one lane per pixel")
    args = ap.parse_args()
    rt = entry.load_package()
    w, h = map(int, args.cam.split("x"))
    sky = rt.load_skybox(max_width=2048)
    scenes = [
        ("bunny", lambda: rt.Scene.main(sky=sky, cam=(w, h))),
        ("spheres_10k", lambda: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=(w, h))),
        ("bunny_grid_230x230", lambda: rt.Scene.grid(230, 230, sky=sky, cam=(w, h))),
        ("terrain_300_two_bunnies", lambda: rt.Scene.terrain(300, seed=1, with_bunny=True, sky=sky, cam=(w, h))),
    ]
    want = set(filter(None, args.scenes.split(",")))
    for name, make in scenes:
        if want and name not in want:
            continue
        t0 = time.perf_counter()
        s = make()
        t_build = time.perf_counter() - t0
        s.upload(0)
        s.set_option(rt.OPT_DEPTH_FIRST_ROUNDS, args.df_rounds)
        s.render(max_depth=args.depth, shadow_rays=1, scatter_rays=args.scatter, antialiasing=1)
        best = None
        for _ in range(args.reps):
            _, st = s.render(max_depth=args.depth, shadow_rays=1, scatter_rays=args.scatter, antialiasing=args.spp_root)
            if best is None or st["device_ms"] < best["device_ms"]:
                best = st
        st = best
        rays = st["closest_rays"] + st["any_rays"]
        print(f"{name:26s} host build {t_build:6.2f} s  rays {rays:>11d}  device {st['device_ms']:8.2f} ms  {rays / st['device_ms'] / 1e3:8.1f} Mrays/s  launches {st['launches']}", flush=True)


if __name__ == "__main__":
    main()
