#!/usr/bin/env python3
"""Per-launch picture of the trace launches of one sample pass: when the waves run dry and when they end.

Timeline library (`make -C rust-raytracer_amd/csrc timeline`).  Renders the frame at 1 spp (one pass: one trace launch per
bounce, each with its own queue-head word) and prints, per launch, percentiles of the waves' dry and end times relative to
the first wave's start, and how full the wave slots were.  Run on the GPU box:  python tools/launch_timeline.py
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTAMD_LIB", os.path.join(ROOT, "rust-raytracer_amd", "librtamd_timeline.so"))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bunny", choices=["bunny", "spheres", "grid"])
    ap.add_argument("--cam", default="1080x960")
    ap.add_argument("--opt", default="")
    args = ap.parse_args()
    rt = entry.load_package()
    W, H = (int(v) for v in args.cam.split("x"))
    sky = rt.load_skybox(max_width=1024)
    make = {"bunny": lambda: rt.Scene.main(sky=sky, cam=(W, H)), "grid": lambda: rt.Scene.grid(230, 230, sky=sky, cam=(W, H)),
            "spheres": lambda: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=(W, H))}
    scene = make[args.workload]().upload(0)
    opt_ids = {"chunk": rt.OPT_CHUNK, "blocks": rt.OPT_BLOCKS_PER_CU, "static": rt.OPT_STATIC_SHARE, "min_batch": rt.OPT_MIN_BATCH}
    for kv in filter(None, args.opt.split(",")):
        k, v = kv.split("=")
        scene.set_option(opt_ids[k], int(v))
    lib = rt.rtamd_lib()
    SLOTS, WAVES = 16, 8192
    n_words = SLOTS * WAVES * 4
    lib.rtamd_timeline_dump.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_int]
    buf = (C.c_uint64 * n_words)()
    scene.render(max_depth=5, shadow_rays=1, antialiasing=1)
    lib.rtamd_timeline_dump(scene.ctx, buf, n_words, 1)
    scene.set_option(rt.OPT_PROFILE, 1)
    scene.kernel_times(reset=True)
    _, st = scene.render(max_depth=5, shadow_rays=1, antialiasing=1)
    kt = scene.kernel_times(reset=True)
    lib.rtamd_timeline_dump(scene.ctx, buf, n_words, 1)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(SLOTS, WAVES, 4).astype(np.int64)
    print(f"{args.workload} {W}x{H} 1 spp: rays {st['closest_rays'] + st['any_rays']}, trace launches {kt['closest_launches'] + kt['any_launches']}, "
          f"trace {kt['closest_ms'] + kt['any_ms']:.3f} ms")
    print("slot waves | span us | start p50 max | dry p1 p50 p99 | end p1 p10 p50 p90 p99 max | residency | trips p50 max")
    order = sorted((int(t[s][t[s, :, 2] > 0][:, 0].min()), s) for s in range(SLOTS) if (t[s, :, 2] > 0).any())
    for _, s in order:
        w = t[s][t[s, :, 2] > 0]
        t0 = w[:, 0].min()
        start, end = (w[:, 0] - t0) * 1e-2, (w[:, 2] - t0) * 1e-2
        dry = (np.where(w[:, 1] > 0, w[:, 1], w[:, 2]) - t0) * 1e-2
        span = end.max()
        pc = lambda a, q: np.percentile(a, q)
        print(f"{s:4d} {len(w):5d} | {span:7.1f} | {pc(start, 50):5.1f} {start.max():5.1f} | {pc(dry, 1):6.1f} {pc(dry, 50):6.1f} {pc(dry, 99):6.1f} | "
              f"{pc(end, 1):6.1f} {pc(end, 10):6.1f} {pc(end, 50):6.1f} {pc(end, 90):6.1f} {pc(end, 99):6.1f} {end.max():6.1f} | "
              f"{(end - start).sum() / (span * 4096):5.2f} | {pc(w[:, 3], 50):5.0f} {w[:, 3].max():5d}")


if __name__ == "__main__":
    main()
