#!/usr/bin/env python3
"""Kernel time of the batched closest-hit trace on the main scene for small and large batches of incoherent rays: the fixed
cost of a trace launch (launch, staging the scene into LDS, the walk of one ray) and the rate it reaches.

    python tools/trace_floor_probe.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from test_gpu_trace import random_rays  # noqa: E402

rt = entry.load_package()
gpu = rt.Scene.main(sky=rt.load_skybox(max_width=512), cam=(1080, 960)).upload(0)
gpu.set_option(rt.OPT_PROFILE, 1)
for n in (64, 16384, 131072, 262144, 393216, 524288, 600000, 700000, 786432, 1048576, 2097152):
    rays = random_rays(n, seed=5, center=(0, 0, 1), radius=6.0)
    best = 1e9
    for _ in range(5):
        gpu.kernel_times(reset=True)
        gpu.trace_closest(rays)
        best = min(best, gpu.kernel_times()["closest_ms"])
    print(n, "rays:", f"{best * 1e3:.1f} us", f"{n / best / 1e3:.1f} Mrays/s", flush=True)
print("static share (eighths of the queue dealt without atomics) x batch size:")
for share in (0, 3, 4, 6, 8):
    gpu.set_option(rt.OPT_STATIC_SHARE, share)
    for n in (262144, 393216, 524288, 786432, 1048576):
        rays = random_rays(n, seed=5, center=(0, 0, 1), radius=6.0)
        best = 1e9
        for _ in range(4):
            gpu.kernel_times(reset=True)
            gpu.trace_closest(rays)
            best = min(best, gpu.kernel_times()["closest_ms"])
        print("share", share, n, "rays:", f"{best * 1e3:.1f} us", f"{n / best / 1e3:.1f} Mrays/s", flush=True)
