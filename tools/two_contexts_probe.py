#!/usr/bin/env python3
"""One frame rendered by K contexts of ONE GPU at once (each its own stream and workspace, one host thread per context,
tiles dealt round-robin) against the same frame through one context: do the launches of independent pixel groups fill
each other's tails when they come from separate submitters?

    python tools/two_contexts_probe.py [--scene main|spheres|grid] [--cam WxH] [--aa N] [--contexts 2] [--tile 40] [--reps 3]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="main", choices=["main", "spheres", "grid"])
    ap.add_argument("--cam", default="1080x960")
    ap.add_argument("--aa", type=int, default=8)
    ap.add_argument("--contexts", type=int, default=2)
    ap.add_argument("--tile", type=int, default=40)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    rt = entry.load_package()
    sky = rt.load_skybox(max_width=None)
    w, h = map(int, args.cam.split("x"))
    make = {"main": lambda: rt.Scene.main(sky=sky, cam=(w, h)), "spheres": lambda: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=(w, h)),
            "grid": lambda: rt.Scene.grid(230, 230, sky=sky, cam=(w, h))}[args.scene]
    params = dict(max_depth=5, shadow_rays=1, antialiasing=args.aa)
    K = args.contexts
    scenes = [make().upload(0) for _ in range(K)]
    full = rt.Scene.make_region(w, h, tile=(args.tile, args.tile))
    n_full = rt.Scene.region_pixels(full)
    out_full = torch.zeros(n_full * 3, dtype=torch.float32, device="cuda:0")
    regions = [rt.Scene.make_region(w, h, tile=(args.tile, args.tile), tile_first=k, tile_stride=K) for k in range(K)]
    outs = [torch.zeros(rt.Scene.region_pixels(r) * 3, dtype=torch.float32, device="cuda:0") for r in regions]
    torch.cuda.synchronize()

    def one():
        t0 = time.perf_counter()
        _, st = scenes[0].render(region=full, out_device=out_full, **params)
        return time.perf_counter() - t0, st["closest_rays"] + st["any_rays"]

    def many():
        rays = [0] * K

        def work(k):
            _, st = scenes[k].render(region=regions[k], out_device=outs[k], **params)
            rays[k] = st["closest_rays"] + st["any_rays"]
        th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0, sum(rays)

    one(); many()
    for name, f in (("one context", one), (f"{K} contexts", many), ("one context", one), (f"{K} contexts", many)):
        ts = []
        for _ in range(args.reps):
            t, rays = f()
            ts.append(t)
        t = min(ts)
        print(f"{args.scene} {w}x{h} {args.aa * args.aa} spp, {name}: {t * 1e3:.2f} ms, {rays / t / 1e6:.1f} Mrays/s (rays {rays})", flush=True)
    # the K shares assemble to the single context's frame
    a = out_full.cpu().numpy().reshape(-1, args.tile, args.tile, 3)
    ok = True
    for k in range(K):
        b = outs[k].cpu().numpy().reshape(-1, args.tile, args.tile, 3)
        ok = ok and np.array_equal(a[k::K].view(np.uint32), b.view(np.uint32))
    print("identical to the single context's frame:", ok)


if __name__ == "__main__":
    main()
