#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the CPU oracle (tests/oracle_lib.py).

The reference repository holds no fixtures and cannot be built here, so these vectors are outputs of the
C++ restatement (oracle/), pinned to the reference by tests/test_oracle_kat.py.  They freeze the oracle's behaviour:
a later change to the oracle, the host BVH builder or the GPU kernels that alters any of these bits fails
tests/test_golden.py.   Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def seeded_rays(n, seed, center, radius):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(3, n))
    o = o / np.linalg.norm(o, axis=0) * (rng.random(n) ** (1 / 3)) * radius + np.array(center)[:, None]
    d = rng.normal(size=(3, n))
    d = d / np.linalg.norm(d, axis=0)
    rays = np.empty((7, n), np.float32)
    rays[0:3] = o
    rays[3:6] = d
    dn = np.sqrt((rays[3:6].astype(np.float64) ** 2).sum(axis=0))
    rays[3:6] = (rays[3:6] / dn).astype(np.float32)
    rays[6] = rng.choice([1e12, 4.0, 9.0], n).astype(np.float32)
    return np.ascontiguousarray(rays)


def ray_fixture(scene, cam, seed, center, radius, n_random=3072, stride=97):
    prim = scene.primary_rays(0, 0, cam[0], cam[1])[:, ::stride]
    rays = np.ascontiguousarray(np.concatenate([prim, seeded_rays(n_random, seed, center, radius)], axis=1))
    hit = scene.trace_closest(rays, threads=4)
    occ, _ = scene.trace_any(rays, threads=4)
    return dict(rays=rays, obj=hit["obj"], prim=hit["prim"], t=hit["t"], u=hit["u"], v=hit["v"], any=occ)


def main():
    sky = ol.procedural_sky()
    cam = (160, 120)
    s = ol.OracleScene.main(sky=sky, cam=cam)
    nodes, order, stats = s.bvh(0)
    np.savez_compressed(os.path.join(HERE, "bunny_bvh.npz"), nodes=nodes, order=order, stats=stats)
    np.savez_compressed(os.path.join(HERE, "bunny_rays.npz"), **ray_fixture(s, cam, 11, (0, 0, 1), 9.0))
    img, _, cnt = s.render(max_depth=5, shadow_rays=1, antialiasing=2, threads=4)
    np.savez_compressed(os.path.join(HERE, "bunny_render_160x120_aa2.npz"), image=img,
                        closest_rays=cnt["closest_rays"], any_rays=cnt["any_rays"])
    sp = ol.OracleScene.spheres(n=1500, seed=1, sky=sky, cam=cam)
    n2, o2, st2 = sp.bvh(1)
    np.savez_compressed(os.path.join(HERE, "spheres_rays.npz"), tree_nodes=n2, tree_stats=st2, **ray_fixture(sp, cam, 12, (0, 0, 2), 14.0))
    gr = ol.OracleScene.grid(5, 4, sky=sky, cam=cam)
    n3, o3, st3 = gr.bvh(1)
    np.savez_compressed(os.path.join(HERE, "grid_rays.npz"), tree_nodes=n3, tree_stats=st3, **ray_fixture(gr, cam, 13, (0, 0, 1), 14.0))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
