"""CPU-side checks of the host library's loader / writer / builder additions: binary PLY, PNG output, reflections,
the flattened grid, the top-level tree of a plain list.  No GPU."""
import ctypes as C
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNNY = os.path.join(ROOT, "assets", "bunny.ply")


def _ply_info(rt, path, n_first=8):
    counts = np.zeros(2, np.uint32)
    v = np.zeros((n_first, 3), np.float32)
    f = np.zeros((n_first, 3), np.uint32)
    rc = rt.host_lib().rth_ply_info(path.encode(), counts.ctypes.data_as(C.POINTER(C.c_uint32)), v.ctypes.data_as(C.POINTER(C.c_float)),
                                    f.ctypes.data_as(C.POINTER(C.c_uint32)), n_first)
    return rc, counts, v, f


def _write_binary_ply(src, dst, big_endian, index_type="int"):
    """Re-encodes the ASCII bunny (x y z confidence intensity; list uchar int faces) as a binary PLY."""
    lines = open(src).read().split("\n")
    end = lines.index("end_header")
    nv = int([l for l in lines if l.startswith("element vertex")][0].split()[2])
    nf = int([l for l in lines if l.startswith("element face")][0].split()[2])
    e = ">" if big_endian else "<"
    header = "\n".join(l.replace("format ascii 1.0", "format binary_%s_endian 1.0" % ("big" if big_endian else "little"))
                        .replace("list uchar int", "list uchar " + index_type) for l in lines[:end + 1]) + "\n"
    body = b""
    for l in lines[end + 1:end + 1 + nv]:
        body += struct.pack(e + "5f", *[float(x) for x in l.split()])
    for l in lines[end + 1 + nv:end + 1 + nv + nf]:
        t = [int(x) for x in l.split()]
        body += struct.pack(e + "B", t[0]) + struct.pack(e + ("%di" if index_type == "int" else "%dI") % t[0], *t[1:1 + t[0]])
    open(dst, "wb").write(header.encode() + body)


def test_binary_ply_equals_ascii(rt, tmp_path, sky):
    rc, counts, v, f = _ply_info(rt, BUNNY)
    assert rc == 0 and tuple(counts) == (453, 948)
    for big, itype in ((False, "int"), (True, "int"), (False, "uint")):
        path = str(tmp_path / f"bunny_{'be' if big else 'le'}_{itype}.ply")
        _write_binary_ply(BUNNY, path, big, itype)
        rc2, counts2, v2, f2 = _ply_info(rt, path)
        assert rc2 == 0 and np.array_equal(counts2, counts)
        assert np.array_equal(v2.view(np.uint32), v.view(np.uint32)) and np.array_equal(f2, f)
        # the whole mesh: same tree, same triangle order
        a, b = rt.Scene.main(sky=sky, cam=(64, 48)), rt.Scene.main(sky=sky, cam=(64, 48), ply=path)
        for x, y in zip(a.tree(0), b.tree(0)):
            assert np.array_equal(x, y)
    # a truncated body is an error, not a crash
    path = str(tmp_path / "short.ply")
    data = open(str(tmp_path / "bunny_le_int.ply"), "rb").read()
    open(path, "wb").write(data[:len(data) // 2])
    assert _ply_info(rt, path)[0] != 0


def test_png_writer_round_trip(rt, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    for w, h in ((37, 21), (300, 260)):   # the second spans several stored deflate blocks
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = str(tmp_path / f"t_{w}.png")
        assert rt.host_lib().rth_save_png(path.encode(), img.ctypes.data, w, h) == 0
        back = np.asarray(Image.open(path).convert("RGB"))
        assert back.shape == img.shape and np.array_equal(back, img)


def test_reflection_matrix_as_the_reference_writes_it(rt):
    f = np.float32
    for axis in ((1.0, 0.0, 0.0), (0.3, 0.2, 1.0), (0.0, -2.0, 0.5)):
        a = np.array(axis, np.float32)
        out = np.zeros(9, np.float32)
        rt.host_lib().rth_kat_reflection(a.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
        n = a / f(np.sqrt(f(f(f(a[0] * a[0]) + f(a[1] * a[1])) + f(a[2] * a[2]))))
        x, y, z = n
        two = f(2.0)
        exp = np.array([f(1) - f(f(two * x) * x), f(0) - f(f(two * x) * y), f(0) - f(f(two * x) * z),
                        f(0) - f(f(two * x) * y), f(1) - f(f(two * y) * y), f(0) - f(f(two * y) * z),
                        f(0) - f(f(two * x) * z), f(f(f(f(0) * two) * y) * z), f(1) - f(f(two * z) * z)], np.float32)   # [7]: `0.0 * 2.0 * b * c`
        assert np.array_equal(out.view(np.uint32), exp.view(np.uint32)), axis


def test_flat_grid_and_scatter_scenes_match_the_oracle_on_the_host(rt, ol, sky):
    # object tree of the flattened grid: boxes come from the instances' world-space meshes
    g, c = rt.Scene.grid(5, 4, sky=sky, cam=(64, 48), flat=True), ol.OracleScene.grid(5, 4, sky=sky, cam=(64, 48), flat=True)
    gn, go, gs = g.tree(1)
    cn, co, cs = c.bvh(1)
    assert np.array_equal(gn, cn) and np.array_equal(gs, cs)
    # the instanced form has different (looser, transformed) object boxes
    gi = rt.Scene.grid(5, 4, sky=sky, cam=(64, 48)).tree(1)[0]
    assert not np.array_equal(gi, gn)
    # plain list: no tree; accelerated: one tree over the run, the same as the oracle's
    assert rt.Scene.scatter(40, seed=2, accelerate=False, sky=sky, cam=(64, 48)).tree(1) is None
    gn, _, gs = rt.Scene.scatter(40, seed=2, accelerate=True, sky=sky, cam=(64, 48)).tree(1)
    cn, _, cs = ol.OracleScene.scatter(40, seed=2, accelerate=True, sky=sky, cam=(64, 48)).bvh(1)
    assert np.array_equal(gn, cn) and np.array_equal(gs, cs) and gs[0] == 40


def test_tree_builder_prints_the_reference_statistics_line(rt, capfd):
    """AABBTree::new prints "BVH statistics: ..." (bvh.rs:173-204); the C++ host's builder does when asked to (rtamd_main does)."""
    rng = np.random.default_rng(3)
    lo = rng.uniform(-5, 5, (500, 3)).astype(np.float32)
    boxes = np.concatenate([lo, lo + rng.uniform(0.01, 0.5, (500, 3)).astype(np.float32)], axis=1)
    rt.host_lib().rth_set_print_stats(1)
    try:
        nodes, order, _ = rt.build_tree(boxes, hit_cost=1.0)
    finally:
        rt.host_lib().rth_set_print_stats(0)
    out = capfd.readouterr().out
    sizes = nodes[1:, 7].astype(np.int64) - nodes[:-1, 7].astype(np.int64)
    leaf = sizes[sizes > 0]
    assert f"BVH statistics: 500 objs, {len(nodes)} nodes, {len(leaf)} leafs, min/max/avg leaf: {leaf.min()}/{leaf.max()}/" in out
    rt.build_tree(boxes, hit_cost=1.0)
    assert "BVH statistics" not in capfd.readouterr().out
