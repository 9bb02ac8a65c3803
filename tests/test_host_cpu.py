"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol, the host library builds
the same trees / camera as the oracle, scene flattening is consistent, and nothing silently falls back to the CPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_exports_every_declared_symbol(rt):
    lib = rt.rtamd_lib()
    header = open(os.path.join(ROOT, "include", "rtamd.h")).read()
    declared = set(re.findall(r"\b(rtamd_[a-z_]+)\s*\(", header))
    assert declared == set(rt.ABI_SYMBOLS), declared ^ set(rt.ABI_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_abi_struct_sizes_match_header(rt):
    # the ctypes mirrors must have the layout of the C structs (checked against sizes computed from the header types)
    assert C.sizeof(rt.Rays) == 7 * 8 and C.sizeof(rt.Hits) == 5 * 8
    assert C.sizeof(rt.Camera) == 12 * 4 + 8 and C.sizeof(rt.Integrator) == 16 and C.sizeof(rt.Region) == 32
    assert C.sizeof(rt.Stats) == 4 * 8 + 4 * 8


def test_region_pixels(rt):
    r = rt.Scene.make_region(1080, 960, tile=(64, 64), tile_first=0, tile_stride=1)
    assert rt.Scene.region_pixels(r) == 17 * 15 * 64 * 64
    total = sum(rt.Scene.region_pixels(rt.Scene.make_region(1080, 960, tile=(64, 64), tile_first=k, tile_stride=8)) for k in range(8))
    assert total == 17 * 15 * 64 * 64
    assert rt.Scene.region_pixels(rt.Region(0, 0, 0, 10, 8, 8, 0, 1)) == 0  # empty rectangle is rejected
    assert rt.Scene.region_pixels(rt.Region(0, 0, 10, 10, 8, 8, 3, 2)) == 0  # tile_first must be < tile_stride


def test_host_trees_and_camera_match_oracle(rt, ol, sky):
    g, c = rt.Scene.main(sky=sky, cam=(1080, 960)), ol.OracleScene.main(sky=sky, cam=(1080, 960))
    gn, go, gs = g.tree(0)
    cn, co, cs = c.bvh(0)
    assert np.array_equal(gn, cn) and np.array_equal(go, co) and np.array_equal(gs, cs)
    gc = g.camera()
    cc, wh = c.camera()
    flat = np.array(list(gc.pos) + list(gc.dir) + list(gc.horizontal) + list(gc.vertical), np.float32)
    assert np.array_equal(flat.view(np.uint32), cc.view(np.uint32)) and (gc.width, gc.height) == wh
    for make_g, make_c in ((lambda: rt.Scene.spheres(n=700, seed=3, sky=sky), lambda: ol.OracleScene.spheres(n=700, seed=3, sky=sky)),
                           (lambda: rt.Scene.grid(7, 3, sky=sky), lambda: ol.OracleScene.grid(7, 3, sky=sky))):
        gn, _, gs = make_g().tree(1)
        cn, _, cs = make_c().bvh(1)
        assert np.array_equal(gn, cn) and np.array_equal(gs, cs)


def test_terrain_scene_host_matches_oracle(rt, ol, sky):
    for n, wb in ((6, True), (40, False)):
        g, c = rt.Scene.terrain(n, seed=3, with_bunny=wb, sky=sky, cam=(64, 48)), ol.OracleScene.terrain(n, seed=3, with_bunny=wb, sky=sky, cam=(64, 48))
        gn, go, gs = g.tree(0)
        cn, co, cs = c.bvh(0)
        assert np.array_equal(gn, cn) and np.array_equal(go, co) and np.array_equal(gs, cs)
        assert gs[0] == 2 * n * n + sum(1 for iy in range(n) for ix in range(n) if (ix + iy) % 7 == 0)  # duplicates are kept


def test_textured_scene_host_matches_oracle(rt, ol, sky):
    """Same mesh tree and camera on both sides of the textured scene (materials are compared through the GPU render)."""
    tex = (np.arange(5 * 7 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
    g, c = rt.Scene.textured(sky=sky, tex=tex, cam=(64, 48)), ol.OracleScene.textured(sky=sky, tex=tex, cam=(64, 48))
    gn, go, gs = g.tree(0)
    cn, co, cs = c.bvh(0)
    assert np.array_equal(gn, cn) and np.array_equal(go, co) and np.array_equal(gs, cs)
    gc = g.camera()
    cc, wh = c.camera()
    flat = np.array(list(gc.pos) + list(gc.dir) + list(gc.horizontal) + list(gc.vertical), np.float32)
    assert np.array_equal(flat.view(np.uint32), cc.view(np.uint32)) and (gc.width, gc.height) == wh


def test_no_cpu_fallback(rt, sky):
    """Without an uploaded GPU context every trace / render call must raise, never compute on the host."""
    s = rt.Scene.main(sky=sky, cam=(32, 32))
    rays = np.zeros((7, 8), np.float32)
    rays[5] = 1.0
    for call in (lambda: s.trace_closest(rays), lambda: s.trace_any(rays), lambda: s.render()):
        with pytest.raises(rt.RtamdError):
            call()
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(rt.RtamdError):  # rtamd_create reports "no usable GPU" instead of falling back
            s.upload(0)


def test_product_does_not_reference_the_oracle():
    """Only tests, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "rust-raytracer_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle_lib" not in text and "librt_oracle" not in text and "rt_oracle" not in text, os.path.join(base, f)


def test_assemble_tiles_round_trip(rt):
    rng = np.random.default_rng(0)
    W, H, tw, th = 70, 50, 16, 16
    img = rng.random((H, W, 3)).astype(np.float32)
    full = rt.Scene.make_region(W, H, tile=(tw, th))
    tiles_x, tiles_y = (W + tw - 1) // tw, (H + th - 1) // th
    out = np.zeros_like(img)
    for rank in range(3):
        ids = list(range(rank, tiles_x * tiles_y, 3))
        tiles = np.zeros((len(ids), th, tw, 3), np.float32)
        for q, t in enumerate(ids):
            ty, tx = divmod(t, tiles_x)
            blk = img[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            tiles[q, :blk.shape[0], :blk.shape[1]] = blk
        rt.assemble_tiles(tiles, full, rank, 3, out)
    assert np.array_equal(out, img)
