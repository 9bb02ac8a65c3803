"""GPU parity of the batched Geometry::hit / is_hit entry points against the CPU oracle.

Bar: object id, primitive id (tree order), t, u, v bit-exact for triangle meshes and cuboids; for
spheres obj / t bit-exact and u, v (atan2 / acos, library-dependent last bits) within 1e-6 absolute.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_rays(n, seed, center=(0.0, 0.0, 1.0), radius=9.0, tmax=1e12):
    """Seeded incoherent rays: origins in a ball around the scene, directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(3, n))
    o = o / np.linalg.norm(o, axis=0) * (rng.random(n) ** (1 / 3)) * radius + np.array(center)[:, None]
    d = rng.normal(size=(3, n))
    d = d / np.linalg.norm(d, axis=0)
    rays = np.empty((7, n), np.float32)
    rays[0:3] = o
    rays[3:6] = d.astype(np.float32)
    # the oracle's sphere test assumes unit directions like the reference; renormalise in f32
    dn = np.sqrt((rays[3:6].astype(np.float64) ** 2).sum(axis=0))
    rays[3:6] = (rays[3:6] / dn).astype(np.float32)
    rays[6] = tmax
    return np.ascontiguousarray(rays)


def rays_in_box_planes(scene, n, seed):
    """Axis-aligned rays whose origin lies exactly in a plane of a box of the scene's object tree, mixed with such rays from
    random origins: a slab product of the tree's boxes is 0 x inf = NaN, and the reference's min! / max! rules decide."""
    words, _, _ = scene.tree(1)
    box = words[:-1, :6].copy().view(np.float32)            # min xyz, max xyz of every node but the sentinel
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, box.shape[0], n)
    lo, hi = box[pick, 0:3], box[pick, 3:6]
    o = (lo + (hi - lo) * rng.random((n, 3)).astype(np.float32)).astype(np.float32)
    plane = rng.integers(0, 3, n)                              # the axis whose plane the origin lies in ...
    side = rng.random(n) < 0.5
    o[np.arange(n), plane] = np.where(side, lo[np.arange(n), plane], hi[np.arange(n), plane])
    along = (plane + 1 + rng.integers(0, 2, n)) % 3            # ... and the axis the ray runs along (another one)
    d = np.zeros((n, 3), np.float32)
    d[np.arange(n), along] = np.where(rng.random(n) < 0.5, 1.0, -1.0)
    back = rng.random(n) < 0.5                                 # half of them start outside the scene and run through it
    o[back, along[back]] = -d[back, along[back]] * 40.0
    rays = np.empty((7, n), np.float32)
    rays[0:3], rays[3:6], rays[6] = o.T, d.T, 1e12
    loose = np.arange(n) % 4 == 3                              # origins off the planes
    rays[0:3, loose] += rng.normal(size=(3, int(loose.sum()))).astype(np.float32)
    return np.ascontiguousarray(rays)


def check_closest(rt, got, ref, exact_uv=True):
    ref_obj = ref["obj"].astype(np.int64)
    ref_obj[ref_obj < 0] = rt.MISS
    assert np.array_equal(got["obj"].astype(np.int64), ref_obj)
    hit = ref["obj"] >= 0
    assert np.array_equal(got["prim"][hit], ref["prim"][hit])
    assert np.array_equal(got["t"][hit].view(np.uint32), ref["t"][hit].view(np.uint32))
    if exact_uv:
        assert np.array_equal(got["u"][hit].view(np.uint32), ref["u"][hit].view(np.uint32))
        assert np.array_equal(got["v"][hit].view(np.uint32), ref["v"][hit].view(np.uint32))
    else:
        # acos of a normal component that rounds above 1 is NaN on both sides
        for k in ("u", "v"):
            g, r = got[k][hit], ref[k][hit]
            assert np.array_equal(np.isnan(g), np.isnan(r)), k
            ok = ~np.isnan(r)
            assert np.abs(g[ok] - r[ok]).max(initial=0.0) <= 1e-6, (k, float(np.abs(g[ok] - r[ok]).max()))
    assert np.all(got["prim"][~hit] == rt.MISS)
    return int(hit.sum())


@pytest.fixture(scope="module")
def main_pair(rt, ol, sky):
    cam = (480, 270)
    return rt.Scene.main(sky=sky, cam=cam).upload(0), ol.OracleScene.main(sky=sky, cam=cam), cam


def test_primary_rays_bit_exact(rt, main_pair):
    gpu, cpu, cam = main_pair
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    assert hits > 0.3 * rays.shape[1]


def test_incoherent_rays_bit_exact(rt, main_pair):
    gpu, cpu, _ = main_pair
    rays = random_rays(200_000, seed=7)
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    assert hits > 1000


def test_any_hit_matches(rt, main_pair):
    gpu, cpu, _ = main_pair
    rays = random_rays(200_000, seed=11)
    rays[6] = np.random.default_rng(3).uniform(0.0, 12.0, rays.shape[1]).astype(np.float32)
    got = gpu.trace_any(rays)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(got, ref)
    assert 0 < ref.sum() < ref.size


def test_kernel_variants_agree(rt, main_pair):
    """LDS-staged vs L2-read mesh, persistent scheduler vs plain nested loops: all four give the same bits."""
    gpu, cpu, _ = main_pair
    rays = random_rays(100_000, seed=13)
    ref = cpu.trace_closest(rays)
    ref_any, _ = cpu.trace_any(rays)
    try:
        for sched in (0, 1):
            for lds in (0, 1):
                gpu.set_option(rt.OPT_SCHEDULER, sched)
                gpu.set_option(rt.OPT_LDS_SCENE, lds)
                check_closest(rt, gpu.trace_closest(rays), ref)
                assert np.array_equal(gpu.trace_any(rays), ref_any)
    finally:
        gpu.set_option(rt.OPT_SCHEDULER, 1)
        gpu.set_option(rt.OPT_LDS_SCENE, 1)


def test_axis_aligned_and_degenerate_rays(rt, main_pair):
    """Rays with zero direction components (inf / NaN slabs), origins on box planes, tiny t_max."""
    gpu, cpu, _ = main_pair
    nodes, _, _ = cpu.bvh(0)
    rng = np.random.default_rng(5)
    n = 30_000
    rays = random_rays(n, seed=17)
    axis = rng.integers(0, 3, n)
    for a in range(3):  # axis-aligned directions
        sel = (axis == a) & (np.arange(n) % 2 == 0)
        rays[3:6, sel] = 0.0
        rays[3 + a, sel] = np.where(rng.random(sel.sum()) < 0.5, 1.0, -1.0)
    rays[3, ::7] = -0.0  # negative zero component
    rays[6, ::5] = rng.uniform(0, 1e-3, rays[6, ::5].shape)  # tiny t_max
    rays[6, 1::11] = 0.0
    check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)


def test_empty_and_ragged_batches(rt, main_pair):
    gpu, cpu, _ = main_pair
    empty = np.zeros((7, 0), np.float32)
    assert gpu.trace_closest(empty)["obj"].size == 0
    assert gpu.trace_any(empty).size == 0
    for n in (1, 63, 65, 511, 513, 4097):
        rays = random_rays(n, seed=100 + n)
        check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))


def test_sphere_scene(rt, ol, sky):
    cam = (320, 240)
    gpu = rt.Scene.spheres(n=3000, seed=1, sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.spheres(n=3000, seed=1, sky=sky, cam=cam)
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays), exact_uv=False)
    assert hits > 0.5 * rays.shape[1]
    rays = random_rays(100_000, seed=23, center=(0, 0, 2), radius=14.0)
    check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays), exact_uv=False)
    rays[6] = np.random.default_rng(4).uniform(0.0, 20.0, rays.shape[1]).astype(np.float32)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)


def test_instanced_grid_scene(rt, ol, sky):
    cam = (320, 240)
    gpu = rt.Scene.grid(6, 5, sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.grid(6, 5, sky=sky, cam=cam)
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    assert hits > 0.3 * rays.shape[1]
    rays = random_rays(100_000, seed=29, center=(0, 0, 1), radius=16.0)
    check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    rays[6] = np.random.default_rng(6).uniform(0.0, 20.0, rays.shape[1]).astype(np.float32)
    ref, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), ref)


@pytest.mark.parametrize("kind", ["spheres", "grid"])
def test_rays_in_the_planes_of_object_tree_boxes(rt, ol, sky, kind):
    """Object trees of scenes too large for LDS tables (3 000 spheres; 12 x 10 bunny instances): rays that turn slab products
    of the tree's boxes into NaNs take the reference's box test, the others a plain one; both kinds share waves here."""
    cam = (64, 64)
    if kind == "spheres":
        gpu, cpu = rt.Scene.spheres(n=3000, seed=1, sky=sky, cam=cam), ol.OracleScene.spheres(n=3000, seed=1, sky=sky, cam=cam)
    else:
        gpu, cpu = rt.Scene.grid(12, 10, sky=sky, cam=cam), ol.OracleScene.grid(12, 10, sky=sky, cam=cam)
    gpu.upload(0)
    rays = rays_in_box_planes(gpu, 40_000, seed=41)
    mixed = np.ascontiguousarray(np.concatenate([rays, random_rays(40_000, seed=43, center=(0, 0, 1), radius=14.0)], axis=1)[:, np.random.default_rng(2).permutation(80_000)])
    hits = check_closest(rt, gpu.trace_closest(mixed), cpu.trace_closest(mixed), exact_uv=kind != "spheres")
    assert hits > 2000
    mixed[6] = np.random.default_rng(7).uniform(0.0, 30.0, mixed.shape[1]).astype(np.float32)
    ref, _ = cpu.trace_any(mixed)
    assert np.array_equal(gpu.trace_any(mixed), ref)
    assert 0 < ref.sum() < ref.size


@pytest.mark.parametrize("n,with_bunny", [(6, True), (70, True), (70, False)])
def test_two_mesh_scene_with_coplanar_duplicates(rt, ol, sky, n, with_bunny):
    """Heightfield + bunny: n = 6 keeps the heightfield in LDS and the bunny in global memory, n = 70 the other way
    round, without the bunny nothing fits LDS.  One cell in seven lists a triangle twice: the later copy must win."""
    cam = (320, 240)
    gpu = rt.Scene.terrain(n, seed=3, with_bunny=with_bunny, sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.terrain(n, seed=3, with_bunny=with_bunny, sky=sky, cam=cam)
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    ref = cpu.trace_closest(rays)
    hits = check_closest(rt, gpu.trace_closest(rays), ref)
    assert hits > 0.4 * rays.shape[1]
    # straight down onto the heightfield: every cell centre, duplicated cells included
    xs = (np.arange(n) + 0.3) * (12.0 / n) - 6.0
    gx, gy = np.meshgrid(xs, xs + 0.1 * (12.0 / n))
    down = np.zeros((7, gx.size), np.float32)
    down[0], down[1], down[2] = gx.ravel(), gy.ravel(), 3.0
    down[5] = -1.0
    down[6] = 1e12
    ref = cpu.trace_closest(down)
    check_closest(rt, gpu.trace_closest(down), ref)
    on_field = ref["obj"] == 0
    assert on_field.mean() > 0.6
    # straight down along the grid lines: the origin lies in the planes of the tree's boxes, where a slab product is
    # 0 x inf = NaN and the reference's min! / max! rules decide (the vertex columns are reproduced in f32 as the scene makes them)
    cell = np.float32(12.0) / np.float32(n)
    lines = np.float32(-6.0) + cell * np.arange(n + 1, dtype=np.float32)
    ex, ey = np.meshgrid(lines, xs.astype(np.float32))
    cx, cy = np.meshgrid(lines, lines)
    edge = np.zeros((7, 2 * ex.size + cx.size), np.float32)
    edge[0] = np.concatenate([ex.ravel(), ey.ravel(), cx.ravel()])
    edge[1] = np.concatenate([ey.ravel(), ex.ravel(), cy.ravel()])
    edge[2], edge[5], edge[6] = 3.0, -1.0, 1e12
    ref_edge = cpu.trace_closest(edge)
    check_closest(rt, gpu.trace_closest(edge), ref_edge)
    occ_edge, _ = cpu.trace_any(edge)
    assert np.array_equal(gpu.trace_any(edge), occ_edge)
    rays = random_rays(100_000, seed=31, center=(0, 0, 0.5), radius=8.0)
    check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays))
    rays[6] = np.random.default_rng(8).uniform(0.0, 12.0, rays.shape[1]).astype(np.float32)
    occ, _ = cpu.trace_any(rays)
    assert np.array_equal(gpu.trace_any(rays), occ)
    gpu.set_option(rt.OPT_LDS_SCENE, 0)
    check_closest(rt, gpu.trace_closest(down), ref)
    check_closest(rt, gpu.trace_closest(edge), ref_edge)
    assert np.array_equal(gpu.trace_any(edge), occ_edge)


def test_full_4k_primary_batch(rt, ol, sky):
    """The 3840x2160 primary-ray configuration at full size: all 8 294 400 rays against the oracle."""
    cam = (3840, 2160)
    gpu = rt.Scene.main(sky=sky, cam=cam).upload(0)
    cpu = ol.OracleScene.main(sky=sky, cam=cam)
    rays = cpu.primary_rays(0, 0, cam[0], cam[1])
    assert rays.shape[1] == 8_294_400
    hits = check_closest(rt, gpu.trace_closest(rays), cpu.trace_closest(rays, threads=16))
    assert 0.2 * rays.shape[1] < hits < 0.9 * rays.shape[1]


def test_errors_are_reported_not_fatal(rt, sky):
    s = rt.Scene.main(sky=sky, cam=(64, 64))
    with pytest.raises(rt.RtamdError):
        s.trace_closest(np.zeros((7, 4), np.float32))  # not uploaded
    s.upload(0)
    with pytest.raises(rt.RtamdError):
        s.render(antialiasing=0)  # invalid parameter: error code, no abort
    with pytest.raises(rt.RtamdError):
        s.render(region=rt.Scene.make_region(64, 64, rect=(0, 0, 128, 64)))  # outside the camera image
