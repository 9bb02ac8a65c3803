#!/usr/bin/env python3
"""Where the wall time of one bench step goes on one GPU: the blocking render call, the de-interleave of the tiles into the frame
(dist_tiles.gather_frame, asynchronous), and what is left when the device is drained."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

import torch
rt = entry.load_package()
from rust_raytracer_amd import dist_tiles
W, H, TILE = 1080, 960, 40
scene = rt.Scene.main(sky=rt.load_skybox(), cam=(W, H)).upload(0)
region = rt.Scene.make_region(W, H, tile=(TILE, TILE))
n_px = rt.Scene.region_pixels(region)
tiles = torch.zeros((n_px // (TILE * TILE), TILE, TILE, 3), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
for rep in range(3):
    tr = tg = 0.0
    dev = 0.0
    t0 = time.perf_counter()
    for _ in range(8):
        a = time.perf_counter()
        _, st = scene.render(max_depth=5, shadow_rays=1, antialiasing=8, region=region, out_device=tiles)
        b = time.perf_counter()
        img = dist_tiles.gather_frame(tiles, W, H, TILE, TILE, 0, 1)
        c = time.perf_counter()
        tr += b - a; tg += c - b; dev += st["device_ms"]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"per step: wall {(t1 - t0) / 8 * 1e3:.2f} ms, render call {tr / 8 * 1e3:.2f} ms (device {dev / 8:.2f} ms), gather call {tg / 8 * 1e3:.3f} ms", flush=True)
