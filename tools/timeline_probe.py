#!/usr/bin/env python3
"""Where a trace launch spends its time: bulk (all waves busy) against tail (the queue is empty, the last rays finish).

Timeline library (`make -C rust-raytracer_amd/csrc timeline`).  Primary rays of the bunny scene in random order, traced
as batches of n rays; per launch: event time, span from the first wave's start to the last wave's end, the moment the
first wave found the queue empty, mean wave residency.  Run on the GPU box:  python tools/timeline_probe.py [--scene grid]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTAMD_LIB", os.path.join(ROOT, "rust-raytracer_amd", "librtamd_timeline.so"))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="bunny", choices=["bunny", "grid", "spheres"])
    ap.add_argument("--kind", default="closest", choices=["closest", "any"])
    ap.add_argument("--coherent", action="store_true", help="rays in image order instead of a random permutation")
    args = ap.parse_args()
    rt = entry.load_package()
    W, H = 2160, 1920
    sky = rt.load_skybox(max_width=1024)
    make = {"bunny": lambda: rt.Scene.main(sky=sky, cam=(W, H)), "grid": lambda: rt.Scene.grid(230, 230, sky=sky, cam=(W, H)),
            "spheres": lambda: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=(W, H))}
    scene = make[args.scene]().upload(0)
    lib = rt.rtamd_lib()
    lib.rtamd_timeline_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    buf = (C.c_uint64 * 8)()
    c = scene.camera()
    pos = np.array(c.pos, np.float32); d0 = np.array(c.dir, np.float32)
    hor = np.array(c.horizontal, np.float32); ver = np.array(c.vertical, np.float32)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    u = (2.0 * (xs / W) - 1.0).reshape(-1, 1)
    v = (2.0 * (ys / H) - 1.0).reshape(-1, 1)
    d = d0[None, :] + u * hor[None, :] + v * ver[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if not args.coherent:
        d = d[np.random.default_rng(1).permutation(len(d))]
    d = d.astype(np.float32)
    n_all = len(d)
    dev = torch.device("cuda", 0)
    rays = [torch.full((n_all,), float(pos[k]), device=dev) for k in range(3)]
    rays += [torch.from_numpy(np.ascontiguousarray(d[:, k])).to(dev) for k in range(3)]
    rays += [torch.full((n_all,), 1e12, device=dev)]
    hits = [torch.zeros(n_all, dtype=torch.int32, device=dev) for _ in range(2)] + [torch.zeros(n_all, device=dev) for _ in range(3)]
    occ = torch.zeros(n_all, dtype=torch.uint8, device=dev)
    scene.set_option(rt.OPT_PROFILE, 1)
    print("     n   event us | span us  bulk us  tail us  last start us | mean residency  waves  trips/wave")
    for n in (1, 64, 1 << 10, 1 << 14, 1 << 16, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22):
        if n > n_all:
            break
        reps = 5
        for rep in range(reps + 1):
            if rep == 1:
                torch.cuda.synchronize()
                scene.kernel_times(reset=True)
                lib.rtamd_timeline_read(scene.ctx, buf, 1)
            if rep >= 1:
                lib.rtamd_timeline_read(scene.ctx, buf, 1)
            if args.kind == "closest":
                scene.trace_closest_device(rays, hits, n)
            else:
                scene.trace_any_device(rays, occ, n)
        torch.cuda.synchronize()
        kt = scene.kernel_times(reset=True)
        ms = (kt["closest_ms"] + kt["any_ms"]) / max(1, kt["closest_launches"] + kt["any_launches"])
        lib.rtamd_timeline_read(scene.ctx, buf, 1)   # the last launch
        t = [int(x) for x in buf]
        tick = 1e-2   # s_memtime runs at 100 MHz
        span = (t[3] - t[0]) * tick
        dry = (t[2] - t[0]) * tick if t[2] < (1 << 63) else span
        print(f"{n:8d} {ms * 1e3:9.1f} | {span:7.1f} {dry:8.1f} {span - dry:8.1f} {(t[1] - t[0]) * tick:8.1f} | {t[4] * tick / max(span, 1e-9) / 4096:10.2f} of 4096 {t[5]:6d} {t[7] / max(1, t[5]):9.1f}")


if __name__ == "__main__":
    main()
