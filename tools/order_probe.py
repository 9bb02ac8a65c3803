#!/usr/bin/env python3
"""Does the order of a ray queue matter to the trace launch?  Primary rays of the bunny scene at 1080x960 through the batch
entry point in image order, with the rays that hit an object first (a stable partition by the previous trace's result: the
expensive rays start first, the shared tail of the queue is the cheap ones), hits last, and in random order.
Run on the GPU box:  python tools/order_probe.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    rt = entry.load_package()
    W, H = 1080, 960
    sky = rt.load_skybox(max_width=1024)
    scene = rt.Scene.main(sky=sky, cam=(W, H)).upload(0)
    c = scene.camera()
    pos = np.array(c.pos, np.float32); d0 = np.array(c.dir, np.float32)
    hor = np.array(c.horizontal, np.float32); ver = np.array(c.vertical, np.float32)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    u = (2.0 * (xs / W) - 1.0).reshape(-1, 1)
    v = (2.0 * (ys / H) - 1.0).reshape(-1, 1)
    d = d0[None, :] + u * hor[None, :] + v * ver[None, :]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    n = len(d)
    dev = torch.device("cuda", 0)

    def run(dirs, label):
        rays = [torch.full((n,), float(pos[k]), device=dev) for k in range(3)]
        rays += [torch.from_numpy(np.ascontiguousarray(dirs[:, k])).to(dev) for k in range(3)]
        rays += [torch.full((n,), 1e12, device=dev)]
        hits = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)] + [torch.zeros(n, device=dev) for _ in range(3)]
        scene.set_option(rt.OPT_PROFILE, 1)
        for rep in range(6):
            if rep == 1:
                torch.cuda.synchronize()
                scene.kernel_times(reset=True)
            scene.trace_closest_device(rays, hits, n)
        torch.cuda.synchronize()
        kt = scene.kernel_times(reset=True)
        ms = kt["closest_ms"] / max(1, kt["closest_launches"])
        print(f"{label:28s} {ms * 1e3:8.1f} us  {n / ms / 1e3:8.1f} Mrays/s")
        return hits[0].cpu().numpy()

    obj = run(d, "image order")
    hit = obj != -1 if obj.dtype == np.int32 else obj != rt.MISS
    hit = (obj.astype(np.int64) & 0xFFFFFFFF) != rt.MISS
    print(f"  rays that hit something: {hit.mean() * 100:.1f} %; a mesh object (0-2): {(np.isin(obj, [0, 1, 2])).mean() * 100:.1f} %")
    mesh = np.isin(obj, [0, 1, 2])
    run(np.concatenate([d[mesh], d[~mesh]]), "mesh hits first")
    run(np.concatenate([d[hit], d[~hit]]), "all hits first")
    run(np.concatenate([d[~mesh], d[mesh]]), "mesh hits last")
    run(d[np.random.default_rng(1).permutation(n)], "random order")


if __name__ == "__main__":
    main()
