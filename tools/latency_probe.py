#!/usr/bin/env python3
"""Kernel time of the batched closest-hit / any-hit entry points against the number of rays.

Primary rays of the bunny scene in random order (so every batch mixes cheap and expensive rays), traced as
batches of n = 1 Ki ... 4 Mi rays; prints microseconds per launch and Mrays/s.  Shows the fixed cost of a
launch (scene staging, slowest wave) next to the steady-state rate.  Run on the GPU box:  python tools/latency_probe.py
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
    W, H = 2160, 1920
    scene = rt.Scene.main(sky=rt.load_skybox(max_width=1024), cam=(W, H)).upload(0)
    c = scene.camera()
    pos = np.array(c.pos, np.float32); d0 = np.array(c.dir, np.float32)
    hor = np.array(c.horizontal, np.float32); ver = np.array(c.vertical, np.float32)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    u = (2.0 * (xs / W) - 1.0).reshape(-1, 1)
    v = (2.0 * (ys / H) - 1.0).reshape(-1, 1)
    d = d0[None, :] + u * hor[None, :] + v * ver[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rng = np.random.default_rng(1)
    d = d[rng.permutation(len(d))].astype(np.float32)
    n_all = len(d)
    dev = torch.device("cuda", 0)
    rays = [torch.full((n_all,), float(pos[k]), device=dev) for k in range(3)]
    rays += [torch.from_numpy(np.ascontiguousarray(d[:, k])).to(dev) for k in range(3)]
    rays += [torch.full((n_all,), 1e12, device=dev)]
    hits = [torch.zeros(n_all, dtype=torch.int32, device=dev) for _ in range(2)] + [torch.zeros(n_all, device=dev) for _ in range(3)]
    occ = torch.zeros(n_all, dtype=torch.uint8, device=dev)
    scene.set_option(rt.OPT_PROFILE, 1)
    for kind in ("closest", "any"):
        for n in (1, 64, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
            if n > n_all:
                break
            for rep in range(3):
                if kind == "closest":
                    scene.trace_closest_device(rays, hits, n)
                else:
                    scene.trace_any_device(rays, occ, n)
                if rep == 0:
                    torch.cuda.synchronize()
                    scene.kernel_times(reset=True)
            torch.cuda.synchronize()
            kt = scene.kernel_times(reset=True)
            ms = (kt["closest_ms"] + kt["any_ms"]) / max(1, kt["closest_launches"] + kt["any_launches"])
            print(f"{kind:7s} n {n:8d}  {ms * 1e3:9.1f} us/launch  {n / ms / 1e3:9.1f} Mrays/s")


if __name__ == "__main__":
    main()
