#!/usr/bin/env python3
"""Renders the frames of tests/golden/gpu_frames.npz on the GPU with the current build (run on the GPU box, then copy the
file from gpurun_out/ into tests/golden/).  The frames pin the wavefront pipeline's pixels bit for bit across builds:
restructuring launches must not change the order in which a pixel's radiance is summed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402


def frames(rt, sky):
    yield "main_d5_sr2_aa3", rt.Scene.main(sky=sky, cam=(160, 128)), dict(max_depth=5, shadow_rays=2, antialiasing=3)
    yield "main_d2_sr1_aa2", rt.Scene.main(sky=sky, cam=(160, 128)), dict(max_depth=2, shadow_rays=1, antialiasing=2)
    yield "main_d1_sr3_aa2", rt.Scene.main(sky=sky, cam=(160, 128)), dict(max_depth=1, shadow_rays=3, antialiasing=2)
    yield "terrain_d4_sr1_aa2", rt.Scene.terrain(40, seed=2, sky=sky, cam=(160, 128)), dict(max_depth=4, shadow_rays=1, antialiasing=2)
    yield "textured_d4_sr1_aa2", rt.Scene.textured(sky=sky, cam=(160, 128)), dict(max_depth=4, shadow_rays=1, antialiasing=2)
    yield "spheres_d6_sr1_aa2", rt.Scene.spheres(n=2000, sky=sky, cam=(160, 128)), dict(max_depth=6, shadow_rays=1, antialiasing=2)


def main():
    import oracle_lib as ol
    rt = entry.load_package()
    sky = ol.procedural_sky()
    out = {}
    for name, scene, params in frames(rt, sky):
        img, _ = scene.upload(0).render(**params)
        out[name] = img
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gpu_frames.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
