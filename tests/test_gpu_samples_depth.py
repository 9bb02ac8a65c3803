"""The sample counts and depth limits of the benchmark configurations against the oracle: 64 and 256 samples per pixel (the
pixel's one random stream carried over every sample, integrator.rs:36-53), max_depth = 64, and the count of the one
licensed difference of the object-tree walk in passes (a spurious t = 0 hit, triangle.rs:26-28) on millions of rays.
"""
import numpy as np
import pytest

from test_gpu_configs import grazing_rays
from test_gpu_render import compare
from test_gpu_trace import random_rays

pytestmark = pytest.mark.gpu


def test_main_scene_64_samples_per_pixel(rt, ol, sky):
    """antialiasing = 8, max_depth = 5 on a 96 x 64 rectangle of the 1080 x 960 frame that holds bunny, floor and sky: what
    bench.py's headline workload renders (64 sequential sample passes over a primary-ray queue that is kept and re-aimed)."""
    cam = (1080, 960)
    gpu, cpu = rt.Scene.main(sky=sky, cam=cam).upload(0), ol.OracleScene.main(sky=sky, cam=cam)
    for rect in ((492, 440, 96, 64), (300, 600, 96, 64)):
        region = rt.Scene.make_region(cam[0], cam[1], rect=rect)
        params = dict(max_depth=5, shadow_rays=1, antialiasing=8)
        img, st = gpu.render(region=region, **params)
        ref, _, cnt = cpu.render(rect=rect, **params)
        rmse, frac = compare(img, ref)
        assert rmse <= 2e-4, rmse                      # 64 samples average single flipped decisions away: measured ~1e-5
        assert st["paths"] == rect[2] * rect[3] * 64
        assert abs(st["closest_rays"] - cnt["closest_rays"]) <= 2e-3 * cnt["closest_rays"]
        assert abs(st["any_rays"] - cnt["any_rays"]) <= 2e-3 * max(1, cnt["any_rays"])
        assert cnt["closest_rays"] > 1.2 * st["paths"]   # the rectangle is not all sky


def test_grid_8k_256_samples_per_pixel(rt, ol, sky):
    """antialiasing = 16, max_depth = 5 on a 32 x 24 rectangle of the 7680 x 4320 frame of the 230 x 230 instanced grid."""
    cam = (7680, 4320)
    gpu, cpu = rt.Scene.grid(230, 230, sky=sky, cam=cam).upload(0), ol.OracleScene.grid(230, 230, sky=sky, cam=cam)
    rect = (3800, 2400, 32, 24)
    region = rt.Scene.make_region(cam[0], cam[1], rect=rect)
    params = dict(max_depth=5, shadow_rays=1, antialiasing=16)
    img, st = gpu.render(region=region, **params)
    ref, _, cnt = cpu.render(rect=rect, **params)
    rmse, _ = compare(img, ref)
    assert rmse <= 2e-4, rmse
    assert st["paths"] == rect[2] * rect[3] * 256
    assert abs(st["closest_rays"] - cnt["closest_rays"]) <= 2e-3 * cnt["closest_rays"]
    assert cnt["closest_rays"] > 1.2 * st["paths"]


def test_depth_64_differs_only_on_chaotic_mirror_paths(rt, ol, sky):
    """max_depth = 64, one sample per pixel.  The GPU and the oracle draw the same random numbers and make the same
    decisions; what differs is the last bit of powf / sinf / cosf in the glossy lobe and the light samples (device maths
    library against glibc).  A direction that is off by one rounding step stays harmless on a short path; on a path that goes on for
    dozens of mirror bounces over the curved metal bunny the difference grows with every bounce until the two paths part.
    So: outside a handful of pixels the images agree to ~1e-5; every pixel that is off holds a glossy bounce followed by a
    long specular chain; their number is bounded.  (The image RMSE itself is then set by how many such pixels a frame holds:
    1.7e-3 here, under the 2e-3 bar of the shallow tests by luck, which is why it is not what this test asserts.)"""
    cam = (150, 110)
    gpu, cpu = rt.Scene.main(sky=sky, cam=cam).upload(0), ol.OracleScene.main(sky=sky, cam=cam)
    params = dict(max_depth=64, shadow_rays=1, antialiasing=1)
    img, st = gpu.render(**params)
    ref, _, cnt = cpu.render(**params)
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
    off = np.argwhere(d.max(axis=2) > 1e-3)
    assert len(off) <= 0.002 * cam[0] * cam[1], len(off)          # measured: 7 of 16 500
    rest = d.copy()
    for y, x in off:
        rest[y, x] = 0.0
    assert float(np.sqrt((rest ** 2).sum() / d.size)) <= 1e-4     # measured: 9e-6
    assert float(np.sqrt((d ** 2).mean())) <= 6e-3                # measured: 1.7e-3
    assert abs(st["closest_rays"] - cnt["closest_rays"]) <= 5e-3 * cnt["closest_rays"]
    GLOSSY, SPECULAR = 2, (1, 3)
    for y, x in off:
        rec, _ = ol.pixel_path(cpu, int(x), int(y), **params)
        kinds = [int(r[3]) for r in rec]
        assert GLOSSY in kinds, (x, y, kinds)
        after = kinds[kinds.index(GLOSSY) + 1:]
        assert sum(k in SPECULAR for k in after) >= 6, (x, y, kinds)


def test_walk_in_passes_differs_only_by_spurious_hits(rt, sky):
    """The object-tree walk in passes of growing reach (render entry points) against the single walk in the reference's order
    on 4 million rays through the 230 x 230 grid, most of them grazing: bit-identical except where the single walk reports
    the spurious t = 0 hit of a triangle the ray is parallel to (triangle.rs:26-28 + mesh.rs:130) that lies beyond the ray's
    real nearest hit.  Measured: 1 - 2 per 4 million (5 per 16 million, tools/deepening_stress.py)."""
    gpu = rt.Scene.grid(230, 230, sky=sky, cam=(640, 480)).upload(0)
    total = 0
    for r in range(2):
        n = 2_000_000
        rays = np.ascontiguousarray(np.concatenate([grazing_rays(n // 2, 100 + r, 430.0, (-1.0, 4.0), (0.002, 0.03)[r]),
                                                    random_rays(n // 4, seed=200 + r, center=(0, 0, 2), radius=430.0),
                                                    random_rays(n // 4, seed=300 + r, center=(0, 0, 2), radius=430.0 * (3.0, 12.0)[r])], axis=1))
        any_rays = rays.copy()
        any_rays[6] = np.random.default_rng(400 + r).choice(np.array([1.0, 25.0, 300.0, 1e12], np.float32), rays.shape[1])
        single, occ_single = gpu.trace_closest(rays), gpu.trace_any(any_rays)
        try:
            gpu.set_option(rt.OPT_TREE_DEEPENING, 2)
            passes, occ_passes = gpu.trace_closest(rays), gpu.trace_any(any_rays)
        finally:
            gpu.set_option(rt.OPT_TREE_DEEPENING, 1)
        differ = np.zeros(rays.shape[1], bool)
        for k in single:
            differ |= single[k].view(np.uint32) != passes[k].view(np.uint32)
        spurious = (single["obj"] != rt.MISS) & (single["t"] == 0.0) & (single["u"] == 0.0) & (single["v"] == 0.0)
        assert not np.any(differ & ~spurious)
        occ_differ = occ_single != occ_passes
        assert not np.any(occ_differ & ~((occ_single == 1) & (occ_passes == 0)))   # a spurious hit occludes at any t_max
        total += int(differ.sum()) + int(occ_differ.sum())
    assert total <= 12, total


def test_walk_in_passes_with_ray_origins_inside_cuboids(rt, ol, sky):
    """3 000 spheres and cuboids in one top-level tree (more than 256 nodes, walked in passes when asked to), rays that start
    anywhere in the scene's volume, thousands of them inside a cuboid: a cuboid that holds the origin answers its far side
    even beyond t_max (cuboid.rs:22-28), the one primitive whose answer depends on the order of the walk.  The walk in
    passes keeps the reference's order inside a pass, and this tree is not walked nearer child first (it holds cuboids):
    bit for bit the oracle's hits, with the passes on and off."""
    cam = (240, 180)
    gpu = rt.Scene.scatter(3000, seed=5, accelerate=True, sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.scatter(3000, seed=5, accelerate=True, sky=sky, cam=cam)
    # (3 000 objects: the list's top-level tree has thousands of nodes, far beyond the 257 from which a tree is walked in passes)
    rng = np.random.default_rng(91)
    n = 300_000
    rays = random_rays(n, seed=92, center=(0, 0, 1), radius=9.0)
    rays[0:3] = np.stack([rng.uniform(-9, 9, n), rng.uniform(-9, 9, n), rng.uniform(0.0, 2.5, n)]).astype(np.float32)   # inside the objects' slab
    rays = np.ascontiguousarray(rays)
    ref = cpu.trace_closest(rays)
    from test_gpu_trace import check_closest
    for mode in (0, 2):
        try:
            gpu.set_option(rt.OPT_TREE_DEEPENING, mode)
            gpu.set_option(rt.OPT_ORDERED_WALK, 2)
            check_closest(rt, gpu.trace_closest(rays), ref, exact_uv=False)
        finally:
            gpu.set_option(rt.OPT_TREE_DEEPENING, 1)
            gpu.set_option(rt.OPT_ORDERED_WALK, 1)
    # rays that leave a cuboid through its far side: the hit lies behind the origin's own cuboid wall
    inside = (ref["obj"] >= 0) & (ref["t"] > 0)
    assert inside.sum() > 10_000
