"""The frame kernel (RTAMD_OPT_FRAME_KERNEL: every wave owns a set of pixels and runs them from their first camera ray to their
last sample, tracing and shading in turn) against the pipelines of launches: per pixel the same operations in the same order
on the same random stream (integrate_recur, src/integrator.rs:56-104), so pixels and ray counts must be identical, bit for
bit -- with one scatter ray per hit against the wavefront pipeline, with branching scatter rays against the rounds of launches.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both(rt, gpu, **params):
    gpu.set_option(rt.OPT_FRAME_KERNEL, 0)
    a, sa = gpu.render(**params)
    gpu.set_option(rt.OPT_FRAME_KERNEL, 1)
    b, sb = gpu.render(**params)
    gpu.set_option(rt.OPT_FRAME_KERNEL, 2)
    return a, sa, b, sb


def same(a, sa, b, sb):
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())
    assert (sa["closest_rays"], sa["any_rays"], sa["paths"]) == (sb["closest_rays"], sb["any_rays"], sb["paths"])


@pytest.mark.parametrize("aa,depth,shadow,scatter", [(1, 5, 1, 1), (2, 5, 1, 1), (3, 1, 1, 1), (2, 2, 3, 1), (4, 8, 1, 1),
                                                     (2, 5, 1, 2), (1, 4, 2, 3), (3, 1, 1, 2), (2, 3, 1, 0)])
def test_main_scene_is_identical(rt, sky, aa, depth, shadow, scatter):
    gpu = rt.Scene.main(sky=sky, cam=(200, 150)).upload(0)
    a, sa, b, sb = both(rt, gpu, max_depth=depth, shadow_rays=shadow, antialiasing=aa, scatter_rays=scatter)
    same(a, sa, b, sb)
    # one seeding launch, then per window of samples a launch and (between windows) one that prices the pixels for the next deal
    assert sb["launches"] <= 8 and sa["launches"] > sb["launches"]


@pytest.mark.parametrize("rebalance", [0, 1, 3])
def test_sample_windows_do_not_change_pixels(rt, sky, rebalance):
    """16 samples per pixel in 1, 2 or 4 launches ([0, 1) [1, 4) [4, 16)), the waves' pixels dealt anew before each."""
    gpu = rt.Scene.main(sky=sky, cam=(240, 180)).upload(0)
    gpu.set_option(rt.OPT_FRAME_REBALANCE, rebalance)
    for scatter in (1, 2):
        a, sa, b, sb = both(rt, gpu, max_depth=4, shadow_rays=1, antialiasing=4, scatter_rays=scatter)
        same(a, sa, b, sb)
        assert sb["launches"] == 2 + 2 * min(rebalance, 2)     # (16 samples: windows end at 1, 4 and 16)


def test_textured_scene_and_tile_sharding(rt, sky):
    """Textures, a point light, a reflected mesh; rendered as the tiles rank 1 of 3 owns (ragged edge tiles included)."""
    cam = (190, 130)
    gpu = rt.Scene.textured(sky=sky, cam=cam).upload(0)
    region = rt.Scene.make_region(cam[0], cam[1], tile=(32, 32), tile_first=1, tile_stride=3)
    for scatter in (1, 2):
        a, sa, b, sb = both(rt, gpu, max_depth=4, shadow_rays=2, antialiasing=2, scatter_rays=scatter, region=region)
        same(a, sa, b, sb)


def test_more_pixels_than_one_launch_holds(rt, sky):
    """A frame of more pixel slots than 256 per wave of the grid: several launches."""
    cam = (1280, 900)
    gpu = rt.Scene.main(sky=sky, cam=cam).upload(0)
    a, sa, b, sb = both(rt, gpu, max_depth=3, shadow_rays=1, antialiasing=1)
    same(a, sa, b, sb)
    assert sb["launches"] == 4     # two chunks: a seeding launch and a frame launch each


def test_step_threshold_does_not_change_pixels(rt, sky):
    gpu = rt.Scene.main(sky=sky, cam=(160, 120)).upload(0)
    gpu.set_option(rt.OPT_FRAME_KERNEL, 1)
    ref = None
    for step_min in (64, 16, 1):
        gpu.set_option(rt.OPT_FRAME_STEP_MIN, step_min)
        img, st = gpu.render(max_depth=5, shadow_rays=1, antialiasing=2)
        if ref is None:
            ref = img
        assert np.array_equal(ref.view(np.uint32), img.view(np.uint32))


def test_other_kernel_instances(rt, sky):
    """Scenes without meshes (spheres in an object tree), with instanced meshes under an object tree, and with a mesh in global
    memory beside the one in LDS: each runs another instance of the scheduler kernel."""
    cam = (160, 120)
    for scene in (rt.Scene.spheres(n=3000, sky=sky, cam=cam), rt.Scene.grid(12, 12, sky=sky, cam=cam), rt.Scene.terrain(30, seed=2, sky=sky, cam=cam)):
        gpu = scene.upload(0)
        for scatter in (1, 2):
            a, sa, b, sb = both(rt, gpu, max_depth=4, shadow_rays=1, antialiasing=2, scatter_rays=scatter)
            same(a, sa, b, sb)


def test_default_mode_takes_the_frame_kernel_for_branching_scatter_rays_only(rt, sky):
    gpu = rt.Scene.main(sky=sky, cam=(160, 120)).upload(0)
    _, st = gpu.render(max_depth=3, shadow_rays=1, antialiasing=1, scatter_rays=2)
    assert st["launches"] == 2
    _, st = gpu.render(max_depth=3, shadow_rays=1, antialiasing=1, scatter_rays=1)
    assert st["launches"] > 2
