"""GPU parity of the wavefront integrator (rtamd_render) against the CPU oracle's calculate_pixel loop.

The GPU consumes each pixel's random stream in the reference's order, so images agree with the oracle up to
(a) float summation order of the path radiance (forward accumulation instead of the nested recursion) and
(b) last-bit differences of sin/cos/pow/atan2/acos between the device maths library and glibc, which can flip
a shadow-ray or Fresnel decision for single samples.  Stated tolerance: image RMSE <= 2e-3 and at most 0.5 %
of the pixel components off by more than 1e-3; in practice almost every pixel agrees to ~1e-6.
"""
import numpy as np
import pytest

from test_gpu_trace import check_closest

pytestmark = pytest.mark.gpu

RMSE_TOL = 2e-3
OUTLIER_TOL = 1e-3
OUTLIER_FRACTION = 5e-3


def compare(img, ref):
    assert img.shape == ref.shape
    assert not np.isnan(img).any()
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
    rmse = float(np.sqrt((d ** 2).mean()))
    frac = float((d > OUTLIER_TOL).mean())
    assert rmse <= RMSE_TOL, rmse
    assert frac <= OUTLIER_FRACTION, frac
    return rmse, frac


@pytest.fixture(scope="module")
def pair(rt, ol, sky):
    cam = (240, 200)
    return rt.Scene.main(sky=sky, cam=cam).upload(0), ol.OracleScene.main(sky=sky, cam=cam), cam


@pytest.mark.parametrize("params", [
    dict(max_depth=5, shadow_rays=1, antialiasing=1),
    dict(max_depth=5, shadow_rays=1, antialiasing=3),
    dict(max_depth=2, shadow_rays=3, antialiasing=2),
    dict(max_depth=1, shadow_rays=1, antialiasing=1),
    dict(max_depth=0, shadow_rays=1, antialiasing=2),
    dict(max_depth=10, shadow_rays=5, antialiasing=1),   # WhittedIntegrator::new() defaults
])
def test_render_matches_oracle(pair, params):
    gpu, cpu, _ = pair
    img, st = gpu.render(**params)
    ref, _, cnt = cpu.render(**params)
    rmse, frac = compare(img, ref)
    # ray counts follow the same paths: allow the rare flipped decision
    assert abs(st["closest_rays"] - cnt["closest_rays"]) <= 2e-3 * max(1, cnt["closest_rays"])
    assert abs(st["any_rays"] - cnt["any_rays"]) <= 2e-3 * max(1, cnt["any_rays"])
    assert st["paths"] == img.shape[0] * img.shape[1] * params["antialiasing"] ** 2


@pytest.mark.parametrize("params", [
    dict(max_depth=3, shadow_rays=1, scatter_rays=2, antialiasing=2),
    dict(max_depth=4, shadow_rays=2, scatter_rays=3, antialiasing=1),
    dict(max_depth=3, shadow_rays=1, scatter_rays=0, antialiasing=2),   # direct lighting only
    dict(max_depth=5, shadow_rays=1, scatter_rays=1, antialiasing=2),   # wavefront path, for the cross-check below
])
def test_branching_scatter_rays_match_oracle(rt, pair, params):
    """scatter_rays != 1 runs the depth-first kernel (the recursion's random-stream order and nested sums)."""
    gpu, cpu, cam = pair
    img, st = gpu.render(**params)
    ref, _, cnt = cpu.render(**params)
    compare(img, ref)
    assert abs(st["closest_rays"] - cnt["closest_rays"]) <= 2e-3 * max(1, cnt["closest_rays"])
    assert abs(st["any_rays"] - cnt["any_rays"]) <= 2e-3 * max(1, cnt["any_rays"])
    if params["scatter_rays"] != 1:
        # tiles and chunking leave the pixels bit-identical here too
        W, H = cam
        image = np.zeros((H, W, 3), np.float32)
        gpu.set_option(rt.OPT_MAX_PATHS, 5000)
        for rank in range(2):
            region = rt.Scene.make_region(W, H, tile=(64, 48), tile_first=rank, tile_stride=2)
            tiles, _ = gpu.render(region=region, **params)
            rt.assemble_tiles(tiles, region, rank, 2, image)
        gpu.set_option(rt.OPT_MAX_PATHS, 1 << 22)
        assert np.array_equal(image.view(np.uint32), img.view(np.uint32))


@pytest.mark.parametrize("params", [
    dict(max_depth=5, shadow_rays=1, scatter_rays=2, antialiasing=2),
    dict(max_depth=3, shadow_rays=2, scatter_rays=3, antialiasing=1),
    dict(max_depth=1, shadow_rays=1, scatter_rays=2, antialiasing=2),   # every scatter ray ends at the depth limit
    dict(max_depth=4, shadow_rays=1, scatter_rays=0, antialiasing=1),
])
def test_depth_first_rounds_equal_the_one_lane_kernel(rt, pair, params):
    """scatter_rays != 1 has two forms: rounds of one scheduler launch over every pixel's next ray (default) and one lane
    per pixel with nested loops.  Same operations per pixel in the same order: same bits, same ray counts."""
    gpu, _, _ = pair
    gpu.set_option(rt.OPT_FRAME_KERNEL, 0)    # (a frame this small would otherwise take the frame kernel: test_gpu_frame_kernel.py)
    img, st = gpu.render(**params)
    gpu.set_option(rt.OPT_DEPTH_FIRST_ROUNDS, 0)
    try:
        ref, st0 = gpu.render(**params)
    finally:
        gpu.set_option(rt.OPT_DEPTH_FIRST_ROUNDS, 1)
        gpu.set_option(rt.OPT_FRAME_KERNEL, 2)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert (st["closest_rays"], st["any_rays"]) == (st0["closest_rays"], st0["any_rays"])
    assert st["launches"] > st0["launches"]


def test_tiles_equal_full_frame_bitwise(rt, pair):
    """Any tile decomposition yields identical pixels: the seed depends only on (cx, cy).  Tiles whose sides are multiples of
    8 are walked in squares of 8 x 8 pixels, the others row by row: the image must not show which."""
    gpu, _, cam = pair
    full, _ = gpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
    W, H = cam
    for world, tile in ((2, (64, 48)), (3, (64, 48)), (2, (36, 20)), (3, (40, 40)), (2, (8, 8)), (1, (W, H - 1))):
        image = np.zeros((H, W, 3), np.float32)
        for rank in range(world):
            region = rt.Scene.make_region(W, H, tile=tile, tile_first=rank, tile_stride=world)
            tiles, _ = gpu.render(max_depth=4, shadow_rays=1, antialiasing=2, region=region)
            rt.assemble_tiles(tiles, region, rank, world, image)
        assert np.array_equal(image.view(np.uint32), full.view(np.uint32)), (world, tile)


def test_sub_rectangle_and_chunking(rt, pair):
    gpu, cpu, cam = pair
    rect = (37, 21, 130, 90)
    region = rt.Scene.make_region(cam[0], cam[1], rect=rect)
    img, _ = gpu.render(max_depth=3, antialiasing=2, region=region)
    ref, _, _ = cpu.render(max_depth=3, antialiasing=2, rect=rect)
    compare(img, ref)
    # several wavefront chunks must give the same pixels as one
    gpu.set_option(rt.OPT_MAX_PATHS, 4096)
    img2, _ = gpu.render(max_depth=3, antialiasing=2, region=region)
    gpu.set_option(rt.OPT_MAX_PATHS, 1 << 22)
    assert np.array_equal(img.view(np.uint32), img2.view(np.uint32))


def test_sphere_and_grid_scenes_render(rt, ol, sky):
    cam = (200, 150)
    for make_gpu, make_cpu in ((lambda: rt.Scene.spheres(n=2000, sky=sky, cam=cam), lambda: ol.OracleScene.spheres(n=2000, sky=sky, cam=cam)),
                               (lambda: rt.Scene.grid(5, 4, sky=sky, cam=cam), lambda: ol.OracleScene.grid(5, 4, sky=sky, cam=cam))):
        gpu, cpu = make_gpu().upload(0), make_cpu()
        img, _ = gpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
        ref, _, _ = cpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
        compare(img, ref)


def test_terrain_scene_with_point_lights_renders(rt, ol, sky):
    """Two meshes (one in LDS, one in global memory) and the PointLight sampler (ball and true point)."""
    cam = (200, 150)
    gpu, cpu = rt.Scene.terrain(40, seed=2, sky=sky, cam=cam).upload(0), ol.OracleScene.terrain(40, seed=2, sky=sky, cam=cam)
    for params in (dict(max_depth=4, shadow_rays=1, antialiasing=2), dict(max_depth=2, shadow_rays=3, antialiasing=1)):
        img, st = gpu.render(**params)
        ref, _, cnt = cpu.render(**params)
        compare(img, ref)
        assert abs(st["any_rays"] - cnt["any_rays"]) <= 2e-3 * max(1, cnt["any_rays"])


def test_textured_scene_renders(rt, ol, sky):
    """UVTexture, Checkerboard and Image textures on Lambartian / Glossy, Glass, a standalone Triangle, BoundingBox prefilters
    and a Mesh::from_triangles mesh; both integrator forms."""
    cam = (200, 160)
    tex = (np.arange(37 * 53 * 3) * 7 % 256).astype(np.uint8).reshape(37, 53, 3)
    for t in (tex, None):
        gpu, cpu = rt.Scene.textured(sky=sky, tex=t, cam=cam).upload(0), ol.OracleScene.textured(sky=sky, tex=t, cam=cam)
        rays = cpu.primary_rays(0, 0, cam[0], cam[1])
        c = cpu.trace_closest(rays)
        check_closest(rt, gpu.trace_closest(rays), c, exact_uv=False)   # spheres: u, v through atan2 / acos
        assert set(np.unique(c["obj"])) >= set(range(9))   # every object is in view, the BoundingBox-wrapped ones included
        assert np.array_equal(gpu.trace_any(rays), cpu.trace_any(rays)[0])
        for params in (dict(max_depth=4, shadow_rays=1, antialiasing=2), dict(max_depth=3, shadow_rays=1, scatter_rays=2, antialiasing=1)):
            img, _ = gpu.render(**params)
            ref, _, _ = cpu.render(**params)
            compare(img, ref)


def test_tuning_options_do_not_change_pixels(rt, sky):
    """Work distribution knobs (streams, static share, batch size, chunk, blocks, LDS staging) leave every pixel bit-identical."""
    gpu = rt.Scene.main(sky=sky, cam=(320, 256)).upload(0)
    params = dict(max_depth=4, shadow_rays=2, antialiasing=2)
    ref, st0 = gpu.render(**params)
    for opt, val, back in ((rt.OPT_STREAMS, 3, 1), (rt.OPT_STATIC_SHARE, 0, 6), (rt.OPT_STATIC_SHARE, 8, 6), (rt.OPT_MIN_BATCH, 64, 4),
                           (rt.OPT_CHUNK, 256, 64), (rt.OPT_BLOCKS_PER_CU, 1, 2), (rt.OPT_MAX_PATHS, 20000, 1 << 22), (rt.OPT_LDS_SCENE, 0, 1)):
        gpu.set_option(opt, val)
        img, st = gpu.render(**params)
        gpu.set_option(opt, back)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (opt, val)
        assert (st["closest_rays"], st["any_rays"]) == (st0["closest_rays"], st0["any_rays"])


def test_depth_first_rounds_in_chunks_smaller_than_an_earlier_workspace(rt, sky):
    """scatter_rays != 1 rendered in chunks (RTAMD_OPT_MAX_PATHS) on a context whose wavefront workspace an earlier, larger
    render has left behind: the frame stack is laid out for the chunk, not for the workspace's capacity.  Same pixels and
    ray counts as the frame rendered in one piece."""
    params = dict(max_depth=3, shadow_rays=1, scatter_rays=2, antialiasing=2)
    ref, st0 = rt.Scene.main(sky=sky, cam=(128, 96)).upload(0).render(**params)   # one chunk, a context of its own
    gpu = rt.Scene.main(sky=sky, cam=(128, 96)).upload(0)
    gpu.set_option(rt.OPT_FRAME_KERNEL, 0)                                        # (the rounds of launches are what is tested)
    gpu.render(max_depth=2, shadow_rays=1, scatter_rays=1, antialiasing=1)        # wavefront render: a workspace for the whole frame
    gpu.set_option(rt.OPT_MAX_PATHS, 2500)
    img, st = gpu.render(**params)                                                # five chunks in that workspace
    gpu.set_option(rt.OPT_MAX_PATHS, 1 << 22)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert (st["closest_rays"], st["any_rays"]) == (st0["closest_rays"], st0["any_rays"])


def test_lean_and_general_kernel_forms_agree(rt, sky):
    """Scenes of meshes + cuboids, of instanced meshes in an object tree and of spheres + cuboids run the lean form of the traversal
    kernels by default (compiled without the other object kinds and the op-by-op wrapper walk): the general form gives the same
    hits, occlusion flags and pixels, bit for bit."""
    rng = np.random.default_rng(5)
    for scene in (rt.Scene.main(sky=sky, cam=(256, 192)), rt.Scene.grid(12, 9, sky=sky, cam=(256, 192)), rt.Scene.spheres(n=600, seed=3, sky=sky, cam=(256, 192))):
        gpu = scene.upload(0)
        n = 60000
        pos = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
        pos[:, 2] = np.abs(pos[:, 2]) * 0.6 + 0.05
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = np.ascontiguousarray(np.concatenate([pos, d, np.full((n, 1), 1e12, np.float32)], axis=1).astype(np.float32).T)
        lean_hit, lean_any = gpu.trace_closest(rays), gpu.trace_any(rays)
        lean_img, lean_st = gpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
        gpu.set_option(rt.OPT_LEAN_KERNELS, 0)
        full_hit, full_any = gpu.trace_closest(rays), gpu.trace_any(rays)
        full_img, full_st = gpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
        gpu.set_option(rt.OPT_LEAN_KERNELS, 1)
        for k in ("obj", "prim", "t", "u", "v"):
            assert np.array_equal(lean_hit[k].view(np.uint32), full_hit[k].view(np.uint32)), k
        assert np.array_equal(lean_any, full_any)
        assert np.array_equal(lean_img.view(np.uint32), full_img.view(np.uint32))
        assert (lean_st["closest_rays"], lean_st["any_rays"]) == (full_st["closest_rays"], full_st["any_rays"])


def test_quantizer_matches(rt, ol, pair):
    import torch
    gpu, _, _ = pair
    img, _ = gpu.render(max_depth=2, antialiasing=1)
    ref = ol.quantize(img)
    d = torch.from_numpy(img).cuda()
    out = torch.zeros(img.shape, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rc = rt.rtamd_lib().rtamd_quantize_device(gpu.ctx, d.data_ptr(), d.numel(), out.data_ptr())
    assert rc == 0
    assert np.array_equal(out.cpu().numpy(), ref)


def test_pixels_are_those_of_the_pinned_gpu_frames(rt, ol, sky):
    """tests/golden/gpu_frames.npz holds frames rendered by an earlier build of the wavefront pipeline (tools/make_gpu_golden.py):
    changes to the launch structure must leave every pixel bit-identical (same order of every radiance sum)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_gpu_golden
    pinned = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpu_frames.npz"))
    for name, scene, params in make_gpu_golden.frames(rt, sky):
        img, _ = scene.upload(0).render(**params)
        assert np.array_equal(img.view(np.uint32), pinned[name].view(np.uint32)), name
