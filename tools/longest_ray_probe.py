#!/usr/bin/env python3
"""How long the slowest rays of a scene keep one wave busy: the primary rays of a coarse camera grid are traced in launches
of 64 consecutive rays (one wave's batch) and the kernel time of every launch is kept.  A trace launch of the renderer
cannot end before its slowest batch, so the maximum here is the floor of a launch's duration whatever its size.

    python tools/longest_ray_probe.py [--scene grid|spheres|bunny] [--cam 7680x4320] [--step 16]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="grid", choices=["grid", "spheres", "bunny"])
    ap.add_argument("--cam", default="7680x4320")
    ap.add_argument("--step", type=int, default=16, help="every step-th pixel in x and y")
    args = ap.parse_args()
    rt = entry.load_package()
    W, H = map(int, args.cam.split("x"))
    sky = rt.load_skybox(max_width=1024)
    make = {"bunny": lambda: rt.Scene.main(sky=sky, cam=(W, H)), "spheres": lambda: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=(W, H)),
            "grid": lambda: rt.Scene.grid(230, 230, sky=sky, cam=(W, H))}
    scene = make[args.scene]().upload(0)
    c = scene.camera()
    pos = np.array(c.pos, np.float32); d0 = np.array(c.dir, np.float32)
    hor = np.array(c.horizontal, np.float32); ver = np.array(c.vertical, np.float32)
    ys, xs = np.mgrid[0:H:args.step, 0:W:args.step].astype(np.float32)
    u = (2.0 * (xs / W) - 1.0).reshape(-1, 1)
    v = (2.0 * (ys / H) - 1.0).reshape(-1, 1)
    d = d0[None, :] + u * hor[None, :] + v * ver[None, :]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    n_all = len(d) // 64 * 64
    dev = torch.device("cuda", 0)
    scene.set_option(rt.OPT_PROFILE, 1)
    times = []
    for b in range(0, n_all, 64):
        rays = [torch.full((64,), float(pos[k]), device=dev) for k in range(3)]
        rays += [torch.from_numpy(np.ascontiguousarray(d[b:b + 64, k])).to(dev) for k in range(3)]
        rays += [torch.full((64,), 1e12, device=dev)]
        hits = [torch.zeros(64, dtype=torch.int32, device=dev) for _ in range(2)] + [torch.zeros(64, device=dev) for _ in range(3)]
        scene.set_option(rt.OPT_MIN_BATCH, 64)   # keep the 64 rays in one wave
        scene.trace_closest_device(rays, hits, 64)
        torch.cuda.synchronize()
        kt = scene.kernel_times(reset=True)
        times.append(kt["closest_ms"] * 1e3)
    t = np.array(times)
    rows = W // args.step // 64 if W // args.step >= 64 else 1
    print(f"{args.scene} {W}x{H}, every {args.step}th pixel: {len(t)} batches of 64 primary rays")
    print(f"  us per batch: min {t.min():.0f}  median {np.median(t):.0f}  p90 {np.percentile(t, 90):.0f}  p99 {np.percentile(t, 99):.0f}  max {t.max():.0f}")
    worst = np.argsort(t)[-5:][::-1]
    for w in worst:
        y = (w * 64) // (W // args.step) * args.step
        print(f"  batch {w} (image row ~{y}): {t[w]:.0f} us")


if __name__ == "__main__":
    main()
