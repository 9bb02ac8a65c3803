#!/usr/bin/env python3
"""How much of a frame's trace launches is drain: wave residency and the share of it spent after the wave's queue ran dry.

Timeline library (`make -C rust-raytracer_amd/csrc timeline`); renders the headline frame and reads the accumulated wave
stamps (100 MHz real-time counter).  Run on the GPU box:  python tools/tail_probe.py [--workload spheres|grid] [--aa 8]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTAMD_LIB", os.path.join(ROOT, "rust-raytracer_amd", "librtamd_timeline.so"))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bunny", choices=["bunny", "spheres", "grid"])
    ap.add_argument("--aa", type=int, default=8)
    ap.add_argument("--cam", default="1080x960")
    args = ap.parse_args()
    rt = entry.load_package()
    W, H = (int(v) for v in args.cam.split("x"))
    sky = rt.load_skybox()
    make = {"bunny": lambda: rt.Scene.main(sky=sky, cam=(W, H)), "grid": lambda: rt.Scene.grid(230, 230, sky=sky, cam=(W, H)),
            "spheres": lambda: rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=(W, H))}
    scene = make[args.workload]().upload(0)
    lib = rt.rtamd_lib()
    lib.rtamd_timeline_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    buf = (C.c_uint64 * 8)()
    scene.render(max_depth=5, shadow_rays=1, antialiasing=2)
    scene.set_option(rt.OPT_PROFILE, 1)
    scene.kernel_times(reset=True)
    lib.rtamd_timeline_read(scene.ctx, buf, 1)
    _, st = scene.render(max_depth=5, shadow_rays=1, antialiasing=args.aa)
    kt = scene.kernel_times(reset=True)
    lib.rtamd_timeline_read(scene.ctx, buf, 1)
    t = [int(x) for x in buf]
    tick = 1e-5   # ms per tick of the 100 MHz counter
    n_launch = kt["closest_launches"] + kt["any_launches"]
    trace_ms = kt["closest_ms"] + kt["any_ms"]
    life_ms, dry_ms, waves, trips = t[4] * tick, t[6] * tick, t[5], t[7]
    slots = 4096
    print(f"{args.workload} {W}x{H} aa {args.aa}: rays {st['closest_rays'] + st['any_rays']}, trace launches {n_launch}, trace {trace_ms:.2f} ms ({trace_ms / n_launch * 1e3:.1f} us each)")
    print(f"  waves per launch {waves / n_launch:.0f}, mean wave lifetime {life_ms / waves * 1e3:.1f} us, of which after its queue ran dry {dry_ms / waves * 1e3:.1f} us ({100 * dry_ms / life_ms:.1f} %)")
    print(f"  wave residency: {100 * life_ms / (trace_ms * slots):.1f} % of {slots} wave slots x launch time; loop trips per wave {trips / waves:.1f}")


if __name__ == "__main__":
    main()
