"""The frame kernel (RTAMD_OPT_FRAME_KERNEL: the whole integrator in one launch per chunk, every wave owning its pixels from
the first camera ray to the last sample) against the wavefront pipeline: per pixel the same operations in the same order on
the same random stream (integrate_recur, src/integrator.rs:56-104), so pixels and ray counts must be identical, bit for bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both(rt, gpu, **params):
    gpu.set_option(rt.OPT_FRAME_KERNEL, 0)
    a, sa = gpu.render(**params)
    gpu.set_option(rt.OPT_FRAME_KERNEL, 1)
    b, sb = gpu.render(**params)
    gpu.set_option(rt.OPT_FRAME_KERNEL, 0)
    return a, sa, b, sb


def same(a, sa, b, sb):
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())
    assert (sa["closest_rays"], sa["any_rays"], sa["paths"]) == (sb["closest_rays"], sb["any_rays"], sb["paths"])


@pytest.mark.parametrize("aa,depth,shadow", [(1, 5, 1), (2, 5, 1), (3, 1, 1), (2, 2, 3), (4, 8, 1)])
def test_main_scene_is_identical(rt, sky, aa, depth, shadow):
    gpu = rt.Scene.main(sky=sky, cam=(200, 150)).upload(0)
    a, sa, b, sb = both(rt, gpu, max_depth=depth, shadow_rays=shadow, antialiasing=aa)
    same(a, sa, b, sb)
    assert sb["launches"] == 2 and sa["launches"] > 2     # one seeding launch, one frame launch


def test_textured_scene_and_tile_sharding(rt, sky):
    """Textures, a point light, a reflected mesh; rendered as the tiles rank 1 of 3 owns (ragged edge tiles included)."""
    cam = (190, 130)
    gpu = rt.Scene.textured(sky=sky, cam=cam).upload(0)
    region = rt.Scene.make_region(cam[0], cam[1], tile=(32, 32), tile_first=1, tile_stride=3)
    a, sa, b, sb = both(rt, gpu, max_depth=4, shadow_rays=2, antialiasing=2, region=region)
    same(a, sa, b, sb)


def test_more_pixels_than_one_launch_holds(rt, sky):
    """A frame of more pixel slots than 256 per wave of the grid: several launches."""
    cam = (1280, 900)
    gpu = rt.Scene.main(sky=sky, cam=cam).upload(0)
    a, sa, b, sb = both(rt, gpu, max_depth=3, shadow_rays=1, antialiasing=1)
    same(a, sa, b, sb)
    assert sb["launches"] == 4


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


def test_scenes_the_mode_is_not_built_for_take_the_pipeline(rt, sky):
    gpu = rt.Scene.spheres(n=300, sky=sky, cam=(120, 90)).upload(0)
    a, sa, b, sb = both(rt, gpu, max_depth=3, shadow_rays=1, antialiasing=1)
    same(a, sa, b, sb)
