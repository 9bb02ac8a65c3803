#!/usr/bin/env python3
"""Where deep paths leave the oracle's: renders the main scene at a high depth limit on the GPU and with the CPU oracle,
lists the pixels that differ by more than 1e-3, and for each of them the smallest depth limit at which the two renders of
that pixel part, next to the oracle's path (one letter per closest-hit ray: T transparent, M metal, G glossy, L lambertian,
. miss).  Answers: is a large image RMSE at depth 64 a handful of chaotic paths, or a systematic difference?

    python tools/depth_divergence.py [--depth 64] [--aa 1] [--cam 150x110]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=64)
    ap.add_argument("--aa", type=int, default=1)
    ap.add_argument("--cam", default="150x110")
    ap.add_argument("--max-pixels", type=int, default=16)
    args = ap.parse_args()
    rt = entry.load_package()
    import oracle_lib as ol
    w, h = map(int, args.cam.split("x"))
    sky = rt.load_skybox(max_width=512)
    gpu, cpu = rt.Scene.main(sky=sky, cam=(w, h)).upload(0), ol.OracleScene.main(sky=sky, cam=(w, h))
    params = dict(max_depth=args.depth, shadow_rays=1, antialiasing=args.aa)
    img, st = gpu.render(**params)
    ref, _, cnt = cpu.render(**params)
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
    rmse = float(np.sqrt((d ** 2).mean()))
    off = np.argwhere(d.max(axis=2) > 1e-3)
    print(f"{w}x{h}, depth {args.depth}, {args.aa * args.aa} spp: rmse {rmse:.3e}, components off by > 1e-3: {(d > 1e-3).mean():.2e}, pixels off: {len(off)} of {w * h}; "
          f"closest rays gpu / oracle {st['closest_rays']} / {cnt['closest_rays']}")
    rest = d.copy()
    for y, x in off:
        rest[y, x] = 0.0
    print(f"rmse over the other pixels: {float(np.sqrt((rest ** 2).sum() / d.size)):.3e}")
    letters = {-1: ".", 0: "N", 1: "M", 2: "G", 3: "T", 4: "L"}
    for y, x in off[: args.max_pixels]:
        region = rt.Scene.make_region(w, h, rect=(int(x), int(y), 1, 1), tile=(1, 1))
        first = None
        for dep in range(0, args.depth + 1):
            g, _ = gpu.render(max_depth=dep, shadow_rays=1, antialiasing=args.aa, region=region)
            c, _, _ = cpu.render(max_depth=dep, shadow_rays=1, antialiasing=args.aa, rect=(int(x), int(y), 1, 1))
            if np.abs(g.reshape(-1) - c.reshape(-1)).max() > 1e-5:
                first = dep
                break
        rec, rgb = ol.pixel_path(cpu, int(x), int(y), max_depth=args.depth, antialiasing=args.aa)
        path = "".join(letters[int(r[3])] + ("|" if k + 1 < len(rec) and rec[k + 1][0] == 0 else "") for k, r in enumerate(rec))
        _, stp = gpu.render(region=region, **params)
        print(f"pixel ({x}, {y}): |gpu - oracle| {d[y, x].max():.3f}; renders part from depth limit {first}; oracle path ({len(rec)} rays, gpu {stp['closest_rays']}): {path[:160]}")


if __name__ == "__main__":
    main()
