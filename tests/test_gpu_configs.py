"""GPU parity at the sizes the benchmark configurations name: the 230 x 230 instanced grid (50.1 M triangles) under the
7680x4320 camera, 10 000 spheres, the real skybox image, the flattened (world-space) grid, instances seen from far
away, the multi-context render and the top-level tree extension.  Same bar as test_gpu_trace / test_gpu_render.
"""
import numpy as np
import pytest

from test_gpu_render import compare
from test_gpu_trace import check_closest, random_rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def grid_pair(rt, ol, sky):
    cam = (7680, 4320)
    return rt.Scene.grid(230, 230, sky=sky, cam=cam).upload(0), ol.OracleScene.grid(230, 230, sky=sky, cam=cam), cam


def test_grid_230x230_sampled_rays_bit_exact(rt, grid_pair):
    """52 900 instances in one object tree: about 100 k primary rays of the 8K camera, 100 k incoherent rays, any-hit."""
    gpu, cpu, cam = grid_pair
    rays = np.concatenate([cpu.primary_rays(0, y, cam[0], 4) for y in (700, 2160, 3300)] + [cpu.primary_rays(3000, 2500, 128, 64)], axis=1)
    rays = np.ascontiguousarray(rays)
    assert rays.shape[1] > 100_000
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    assert hits > 0.3 * rays.shape[1]
    rays = random_rays(100_000, seed=51, center=(0, 0, 2), radius=420.0)
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    assert hits > 1000
    rays[6] = np.random.default_rng(9).uniform(0.0, 600.0, rays.shape[1]).astype(np.float32)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)
    assert 0 < ref.sum() < ref.size


def test_grid_230x230_scheduler_equals_plain_kernel(rt, grid_pair):
    """A larger batch through both kernels (no oracle): the persistent scheduler and the plain nested loops agree bit for bit."""
    gpu, cpu, cam = grid_pair
    rays = np.ascontiguousarray(np.concatenate([cpu.primary_rays(0, 1800, cam[0], 96), random_rays(300_000, seed=53, center=(0, 0, 2), radius=420.0)], axis=1))
    a = gpu.trace_closest(rays)
    occ_a = gpu.trace_any(rays)
    try:
        gpu.set_option(rt.OPT_SCHEDULER, 0)
        b = gpu.trace_closest(rays)
        occ_b = gpu.trace_any(rays)
    finally:
        gpu.set_option(rt.OPT_SCHEDULER, 1)
    for k in a:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert np.array_equal(occ_a, occ_b)


def grazing_rays(n, seed, half_extent, z_range, dz):
    """Rays that start inside the grid's slab and run nearly parallel to it: they cross hundreds of instance boxes."""
    rng = np.random.default_rng(seed)
    rays = np.zeros((7, n), np.float32)
    rays[0] = rng.uniform(-half_extent, half_extent, n)
    rays[1] = rng.uniform(-half_extent, half_extent, n)
    rays[2] = rng.uniform(z_range[0], z_range[1], n)
    a = rng.uniform(0.0, 2.0 * np.pi, n)
    a[: n // 4] = np.round(a[: n // 4] / (np.pi / 4)) * (np.pi / 4)      # a quarter exactly along the axes and diagonals
    d = np.stack([np.cos(a), np.sin(a), rng.uniform(-dz, dz, n)])
    d[:, : n // 8][np.abs(d[:, : n // 8]) < 1e-6] = 0.0                   # with components that are exactly zero
    rays[3:6] = (d / np.linalg.norm(d, axis=0)).astype(np.float32)
    rays[6] = 1e12
    return rays


def test_grid_230x230_grazing_rays_and_the_walk_in_passes(rt, grid_pair):
    """Grazing rays (they cross hundreds of instance boxes), rays aimed from far outside, closest and any-hit, against the
    oracle, bit for bit.  Then the object-tree walk in passes of growing reach (what the render entry points use; opt-in
    for the batch entry points) against the single walk: identical on these rays.  It may differ only where the reference
    reports a spurious t = 0 hit of a near-parallel triangle (triangle.rs:26-28) that lies beyond the ray's real nearest
    hit -- about one grazing ray in ten million (tools/deepening_stress.py counts them)."""
    gpu, cpu, cam = grid_pair
    graze = grazing_rays(6000, 71, 430.0, (-1.0, 4.0), 0.03)
    far = random_rays(4000, seed=72, center=(0, 0, 2), radius=420.0)
    far[0:3] = far[0:3] * 40.0                                            # origins up to 17 000 away: parameters beyond the limit's range
    mixed = random_rays(6000, seed=73, center=(0, 0, 2), radius=430.0)
    # aimed at the grid from 1000 to 3000 away, low over the horizon: ray parameters on both sides of the limit's range
    rng = np.random.default_rng(75)
    n = 4000
    aimed = np.zeros((7, n), np.float32)
    ang = rng.uniform(0.0, 2.0 * np.pi, n)
    dist = rng.uniform(1000.0, 3000.0, n)
    aimed[0], aimed[1], aimed[2] = dist * np.cos(ang), dist * np.sin(ang), rng.uniform(1.0, 60.0, n)
    target = np.stack([rng.uniform(-400.0, 400.0, n), rng.uniform(-400.0, 400.0, n), rng.uniform(0.0, 3.0, n)])
    d = target - aimed[0:3].astype(np.float64)
    aimed[3:6] = (d / np.linalg.norm(d, axis=0)).astype(np.float32)
    aimed[6] = 1e12
    rays = np.ascontiguousarray(np.concatenate([graze, far, mixed, aimed], axis=1))
    a = gpu.trace_closest(rays)
    ref = cpu.trace_closest(rays)
    hits = check_closest(rt, a, ref)
    assert hits > 4000
    assert not np.any((ref["obj"] >= 0) & (ref["t"] == 0.0))              # none of these rays is one of the spurious hits
    any_rays = rays.copy()
    any_rays[6] = np.random.default_rng(74).choice(np.array([3.0, 40.0, 700.0, 1e12], np.float32), rays.shape[1])
    occ = gpu.trace_any(any_rays)
    assert np.array_equal(occ, cpu.trace_any(any_rays)[0])
    try:
        gpu.set_option(rt.OPT_TREE_DEEPENING, 2)
        b = gpu.trace_closest(rays)
        occ_b = gpu.trace_any(any_rays)
    finally:
        gpu.set_option(rt.OPT_TREE_DEEPENING, 1)
    for k in a:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert np.array_equal(occ, occ_b)


def test_grid_8k_sub_rectangle_render_in_chunks(rt, grid_pair):
    """A sub-rectangle of the 7680x4320 frame against the oracle, max_paths forcing several wavefront chunks."""
    gpu, cpu, cam = grid_pair
    rect = (3712, 2304, 224, 144)
    region = rt.Scene.make_region(cam[0], cam[1], rect=rect)
    params = dict(max_depth=5, shadow_rays=1, antialiasing=2)
    img, st = gpu.render(region=region, **params)
    ref, _, cnt = cpu.render(rect=rect, **params)
    compare(img, ref)
    assert abs(st["closest_rays"] - cnt["closest_rays"]) <= 2e-3 * cnt["closest_rays"]
    gpu.set_option(rt.OPT_MAX_PATHS, 8192)
    try:
        img2, _ = gpu.render(region=region, **params)
    finally:
        gpu.set_option(rt.OPT_MAX_PATHS, 1 << 22)
    assert np.array_equal(img.view(np.uint32), img2.view(np.uint32))


def test_spheres_10000(rt, ol, sky):
    cam = (320, 240)
    gpu = rt.Scene.spheres(n=10000, seed=1, sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.spheres(n=10000, seed=1, sky=sky, cam=cam)
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays), exact_uv=False)
    assert hits > 0.5 * rays.shape[1]
    rays = random_rays(100_000, seed=57, center=(0, 0, 2), radius=14.0)
    check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays), exact_uv=False)
    rays[6] = np.random.default_rng(4).uniform(0.0, 20.0, rays.shape[1]).astype(np.float32)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)
    img, _ = gpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
    ref_img, _, _ = cpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
    compare(img, ref_img)


def aimed_rays(n, seed, dist, scale):
    """Rays from `dist` away (all around, above the floor) aimed into the box of the sphere scene scaled by `scale`."""
    rng = np.random.default_rng(seed)
    a, el = rng.uniform(0.0, 2.0 * np.pi, n), rng.uniform(0.05, 1.2, n)
    o = dist * np.stack([np.cos(a) * np.cos(el), np.sin(a) * np.cos(el), np.sin(el)])
    target = np.stack([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), rng.uniform(0, 4, n)]) * scale
    rays = np.zeros((7, n), np.float32)
    rays[0:3] = o
    d = target - rays[0:3].astype(np.float64)
    rays[3:6] = (d / np.linalg.norm(d, axis=0)).astype(np.float32)
    dn = np.sqrt((rays[3:6].astype(np.float64) ** 2).sum(axis=0))         # unit directions in f32 (sphere.rs:22-33 assumes them)
    rays[3:6] = (rays[3:6] / dn).astype(np.float32)
    rays[6] = 1e12
    return rays


@pytest.mark.parametrize("scale", [1e-4, 1e-5])
def test_tiny_sphere_tree_seen_from_far_walk_in_passes(rt, ol, sky, scale):
    """A tree of 2000 spheres (more than 257 nodes: the walk in passes applies) scaled to 1e-4 / 1e-5 and met by rays from
    0.01 to 2000 units away: the spacing of the objects is near or below the resolution of the ray parameter at the tree's
    entry point, so a pass limit may have no room to grow.  Such a walk must be dropped to the single walk, not repeated
    for ever, and every ray must get the single walk's answer (and the oracle's)."""
    cam = (64, 48)
    host = rt.Scene.spheres(n=2000, seed=3, sky=sky, cam=cam, scale=scale)
    assert host.tree(1)[0].shape[0] >= 257
    gpu = host.upload(0)
    cpu = ol.OracleScene.spheres(n=2000, seed=3, sky=sky, cam=cam, scale=scale)
    rays = np.ascontiguousarray(np.concatenate([aimed_rays(8000, 90 + k, dist, scale) for k, dist in
                                                enumerate((0.01, 0.3, 1.0, 30.0, 300.0, 1000.0, 1500.0, 2000.0, 2500.0))], axis=1))
    ref = cpu.trace_closest(rays)
    a = gpu.trace_closest(rays)
    hits = check_closest(rt, a, ref, exact_uv=False)
    assert hits > 0.3 * rays.shape[1]
    any_rays = rays.copy()
    any_rays[6] = np.random.default_rng(7).choice(np.array([0.5, 1000.0, 1e12], np.float32), rays.shape[1])
    occ = gpu.trace_any(any_rays)
    assert np.array_equal(occ, cpu.trace_any(any_rays)[0])
    try:
        gpu.set_option(rt.OPT_TREE_DEEPENING, 2)                           # the walk in passes, in the batch entry points too
        b = gpu.trace_closest(rays)
        occ_b = gpu.trace_any(any_rays)
    finally:
        gpu.set_option(rt.OPT_TREE_DEEPENING, 1)
    for k in a:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert np.array_equal(occ, occ_b)
    img, _ = gpu.render(max_depth=3, shadow_rays=1, antialiasing=1)       # the render entry points walk in passes by default
    ref_img, _, _ = cpu.render(max_depth=3, shadow_rays=1, antialiasing=1)
    compare(img, ref_img)


def test_real_skybox_render(rt, ol):
    """assets/skybox.jpg (6900 x 3450, decoded by PIL on both sides) through the miss shader: the first + y * w + x texel path."""
    real = ol.load_skybox()
    assert real.shape == (3450, 6900, 3)
    cam = (240, 200)
    gpu, cpu = rt.Scene.main(sky=real, cam=cam).upload(0), ol.OracleScene.main(sky=real, cam=cam)
    for params in (dict(max_depth=5, shadow_rays=1, antialiasing=2), dict(max_depth=0, shadow_rays=1, antialiasing=1)):
        img, _ = gpu.render(**params)
        ref, _, _ = cpu.render(**params)
        compare(img, ref)
    assert len(np.unique((ref * 255).astype(np.uint8).reshape(-1, 3), axis=0)) > 500   # a real image, not a flat colour


def test_flat_grid_matches_oracle(rt, ol, sky):
    """The world-space form of the grid: every instance its own mesh and tree, the oracle built on the same arrays."""
    cam = (320, 240)
    gpu = rt.Scene.grid(6, 5, sky=sky, cam=cam, flat=True).upload(0)
    cpu = ol.OracleScene.grid(6, 5, sky=sky, cam=cam, flat=True)
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    assert hits > 0.3 * rays.shape[1]
    rays = random_rays(100_000, seed=61, center=(0, 0, 1), radius=16.0)
    check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    rays[6] = np.random.default_rng(6).uniform(0.0, 20.0, rays.shape[1]).astype(np.float32)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)
    img, _ = gpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
    ref_img, _, _ = cpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
    compare(img, ref_img)


def test_instances_seen_from_far_away(rt, ol, sky):
    """Rotated instances viewed from 1e4 object sizes away: the object-box prefilter must not drop a ray that the
    reference's own root-box test accepts once rounding at that distance exceeds the box's fixed pad."""
    cam = (64, 48)
    gpu = rt.Scene.grid(6, 5, sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.grid(6, 5, sky=sky, cam=cam)
    rng = np.random.default_rng(71)
    n = 200_000
    target = np.stack([rng.uniform(-13, 13, n), rng.uniform(-11, 11, n), rng.uniform(0.5, 3.5, n)])
    d = rng.normal(size=(3, n))
    d /= np.linalg.norm(d, axis=0)
    dist = 10.0 ** rng.uniform(2.0, 4.6, n)          # up to 4e4 units: the meshes are about 3.5 units wide
    rays = np.empty((7, n), np.float32)
    rays[0:3] = target - d * dist
    dd = (target - rays[0:3].astype(np.float64))
    dd /= np.linalg.norm(dd, axis=0)
    rays[3:6] = dd.astype(np.float32)
    dn = np.sqrt((rays[3:6].astype(np.float64) ** 2).sum(axis=0))
    rays[3:6] = (rays[3:6] / dn).astype(np.float32)
    rays[6] = 1e12
    rays = np.ascontiguousarray(rays)
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    assert hits > 0.2 * n
    rays[6] = (dist * rng.uniform(0.9, 1.1, n)).astype(np.float32)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)


def test_render_over_several_contexts_is_bitwise_the_single_context_frame(rt, sky):
    """render::parallel_render over a list of devices (rtamd_render_multi): two and three contexts on GPU 0."""
    cam = (200, 150)
    params = dict(max_depth=4, shadow_rays=2, antialiasing=2)
    single, st1 = rt.Scene.main(sky=sky, cam=cam).upload(0).render(**params)
    for ids in ([0, 0], [0, 0, 0]):
        scene = rt.Scene.main(sky=sky, cam=cam)
        img, st = rt.render_multi(scene, ids, **params)
        assert np.array_equal(img.view(np.uint32), single.view(np.uint32)), ids
        assert (st["closest_rays"], st["any_rays"], st["paths"]) == (st1["closest_rays"], st1["any_rays"], st1["paths"])


@pytest.mark.parametrize("accelerate", [False, True])
def test_plain_list_and_its_top_level_tree(rt, ol, sky, accelerate):
    """300 spheres / cuboids in a plain GeometryList, walked linearly as the reference does and through the top-level
    tree extension; the oracle is built the same way in both cases."""
    cam = (240, 180)
    gpu = rt.Scene.scatter(300, seed=5, accelerate=accelerate, sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.scatter(300, seed=5, accelerate=accelerate, sky=sky, cam=cam)
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays), exact_uv=False)
    assert hits > 0.5 * rays.shape[1]
    rays = random_rays(60_000, seed=67, center=(0, 0, 1), radius=9.0)
    check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays), exact_uv=False)
    rays[6] = np.random.default_rng(2).uniform(0.0, 12.0, rays.shape[1]).astype(np.float32)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)
    img, _ = gpu.render(max_depth=3, shadow_rays=1, antialiasing=2)
    ref_img, _, _ = cpu.render(max_depth=3, shadow_rays=1, antialiasing=2)
    compare(img, ref_img)


@pytest.mark.parametrize("flat", [2, 3])
def test_merged_flat_grid_matches_oracle(rt, ol, sky, flat):
    """The grid flattened and merged by material: three world-space meshes in a plain list, their trees built on the host
    (2) or by the GPU builder (3); the oracle builds the same three meshes with the reference's builder."""
    cam = (320, 240)
    gpu = rt.Scene.grid(6, 5, sky=sky, cam=cam, flat=flat).upload(0)
    cpu = ol.OracleScene.grid(6, 5, sky=sky, cam=cam, flat=2)
    gn, go, gs = gpu.tree(0)
    cn, co, cs = cpu.bvh(0)
    assert np.array_equal(gn, cn) and np.array_equal(go, co) and gs[0] == 10 * 948   # the first merged mesh: 10 of the 30 instances
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    assert hits > 0.3 * rays.shape[1]
    rays = random_rays(100_000, seed=73, center=(0, 0, 1), radius=16.0)
    check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    rays[6] = np.random.default_rng(6).uniform(0.0, 20.0, rays.shape[1]).astype(np.float32)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)
    img, _ = gpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
    ref_img, _, _ = cpu.render(max_depth=4, shadow_rays=1, antialiasing=2)
    compare(img, ref_img)
