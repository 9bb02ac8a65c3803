#!/usr/bin/env python3
"""Cost of a ray that does no traversal work: 1 Mi rays that miss every object box of the bunny scene, through the
batched closest-hit and any-hit entry points.  What remains is fetch + per-object tests + result store.
Run on the GPU box:  python tools/overhead_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    rt = entry.load_package()
    scene = rt.Scene.main(sky=rt.load_skybox(max_width=256), cam=(64, 64)).upload(0)
    n = 1 << 20
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 2] = np.abs(d[:, 2]) + 0.2           # upwards, away from everything (the scene lies below z = 3 ... the camera's up is -z)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = np.array([-6.0, -6.0, 30.0], np.float32)
    rays = [torch.full((n,), float(pos[k]), device=dev) for k in range(3)] + [torch.from_numpy(np.ascontiguousarray(d[:, k])).to(dev) for k in range(3)]
    rays += [torch.full((n,), 1e12, device=dev)]
    hits = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)] + [torch.zeros(n, device=dev) for _ in range(3)]
    occ = torch.zeros(n, dtype=torch.uint8, device=dev)
    scene.set_option(rt.OPT_PROFILE, 1)
    for kind in ("closest", "any"):
        for rep in range(4):
            if kind == "closest":
                scene.trace_closest_device(rays, hits, n)
            else:
                scene.trace_any_device(rays, occ, n)
            if rep == 0:
                torch.cuda.synchronize(); scene.kernel_times(reset=True)
        torch.cuda.synchronize()
        kt = scene.kernel_times(reset=True)
        ms = (kt["closest_ms"] + kt["any_ms"]) / max(1, kt["closest_launches"] + kt["any_launches"])
        miss = int((hits[0] == -1).sum()) if kind == "closest" else int((occ == 0).sum())
        print(f"{kind:7s} n {n}  {ms * 1e3:8.1f} us/launch  {n / ms / 1e3:9.1f} Mrays/s  misses {miss}")


if __name__ == "__main__":
    main()
