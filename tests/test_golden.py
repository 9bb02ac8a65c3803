"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the CPU oracle).

CPU part: the oracle and the product's host-side BVH builder still reproduce the frozen bits.
GPU part (-m gpu): the HIP path reproduces them through the C ABI.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CAM = (160, 120)


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def scene_for(mod, name, sky):
    if name == "bunny":
        return mod.OracleScene.main(sky=sky, cam=CAM) if hasattr(mod, "OracleScene") else mod.Scene.main(sky=sky, cam=CAM)
    if name == "spheres":
        return mod.OracleScene.spheres(n=1500, seed=1, sky=sky, cam=CAM) if hasattr(mod, "OracleScene") else mod.Scene.spheres(n=1500, seed=1, sky=sky, cam=CAM)
    return mod.OracleScene.grid(5, 4, sky=sky, cam=CAM) if hasattr(mod, "OracleScene") else mod.Scene.grid(5, 4, sky=sky, cam=CAM)


def check_hits(got, g, exact_uv):
    obj = g["obj"].astype(np.int64)
    obj[obj < 0] = 0xFFFFFFFF
    assert np.array_equal(np.asarray(got["obj"]).astype(np.int64) & 0xFFFFFFFF, obj)
    hit = g["obj"] >= 0
    assert np.array_equal(got["prim"][hit], g["prim"][hit])
    assert np.array_equal(got["t"][hit].view(np.uint32), g["t"][hit].view(np.uint32))
    for k in ("u", "v"):
        a, b = got[k][hit], g[k][hit]
        if exact_uv:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
        else:
            assert np.array_equal(np.isnan(a), np.isnan(b))
            ok = ~np.isnan(b)
            assert np.abs(a[ok] - b[ok]).max(initial=0.0) <= 1e-6


# ---------------------------------------------------------------------------------------- CPU
def test_oracle_reproduces_bvh_fixture(ol, sky):
    g = load("bunny_bvh.npz")
    nodes, order, stats = ol.OracleScene.main(sky=sky, cam=CAM).bvh(0)
    assert np.array_equal(nodes, g["nodes"]) and np.array_equal(order, g["order"]) and np.array_equal(stats, g["stats"])
    assert stats.tolist() == [948, 132, 66, 7, 30]  # `BVH statistics` line of bvh.rs:195-203 for the bundled bunny


@pytest.mark.parametrize("name", ["bunny", "spheres", "grid"])
def test_oracle_reproduces_ray_fixtures(ol, sky, name):
    g = load(f"{name}_rays.npz")
    s = scene_for(ol, name, sky)
    got = s.trace_closest(g["rays"], threads=2)
    check_hits(got, g, exact_uv=True)  # same machine code, same libm: everything is bit-identical
    occ, _ = s.trace_any(g["rays"], threads=2)
    assert np.array_equal(occ, g["any"])
    assert 0.1 < (g["obj"] >= 0).mean() < 0.95 and 0 < g["any"].sum() < g["any"].size


def test_oracle_reproduces_render_fixture(ol, sky):
    g = load("bunny_render_160x120_aa2.npz")
    img, _, cnt = ol.OracleScene.main(sky=sky, cam=CAM).render(max_depth=5, shadow_rays=1, antialiasing=2, threads=3)
    assert np.array_equal(img.view(np.uint32), g["image"].view(np.uint32))
    assert cnt["closest_rays"] == int(g["closest_rays"]) and cnt["any_rays"] == int(g["any_rays"])


def test_host_builder_reproduces_tree_fixtures(rt, sky):
    g = load("bunny_bvh.npz")
    nodes, order, stats = rt.Scene.main(sky=sky, cam=CAM).tree(0)
    assert np.array_equal(nodes, g["nodes"]) and np.array_equal(order, g["order"]) and np.array_equal(stats, g["stats"])
    for name in ("spheres", "grid"):
        g = load(f"{name}_rays.npz")
        nodes, _, stats = scene_for(rt, name, sky).tree(1)
        assert np.array_equal(nodes, g["tree_nodes"]) and np.array_equal(stats, g["tree_stats"])


# ---------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bunny", "spheres", "grid"])
def test_gpu_reproduces_ray_fixtures(rt, sky, name):
    g = load(f"{name}_rays.npz")
    s = scene_for(rt, name, sky).upload(0)
    check_hits(s.trace_closest(g["rays"]), g, exact_uv=(name != "spheres"))
    assert np.array_equal(s.trace_any(g["rays"]), g["any"])


@pytest.mark.gpu
def test_gpu_reproduces_render_fixture(rt, sky):
    g = load("bunny_render_160x120_aa2.npz")
    img, st = rt.Scene.main(sky=sky, cam=CAM).upload(0).render(max_depth=5, shadow_rays=1, antialiasing=2)
    d = np.abs(img.astype(np.float64) - g["image"])
    assert np.sqrt((d ** 2).mean()) <= 2e-3 and (d > 1e-3).mean() <= 5e-3
    assert abs(st["closest_rays"] - int(g["closest_rays"])) <= 50 and abs(st["any_rays"] - int(g["any_rays"])) <= 50
