"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE).

Only tests, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
The oracle is the scalar C++ restatement of the reference renderer under ``oracle/``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "librt_oracle.so")
BUNNY_PLY = os.path.join(ROOT, "assets", "bunny.ply")
SKYBOX_JPG = os.path.join(ROOT, "assets", "skybox.jpg")

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("rt_oracle.cpp", "rt_oracle_capi.cpp", "rt_oracle.hpp")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_scene_main.restype = C.c_void_p
        _lib.orc_scene_main.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_scene_spheres.restype = C.c_void_p
        _lib.orc_scene_spheres.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_scene_spheres_scaled.restype = C.c_void_p
        _lib.orc_scene_spheres_scaled.argtypes = [C.c_int, C.c_uint64, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_scene_textured.restype = C.c_void_p
        _lib.orc_scene_textured.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_scene_terrain.restype = C.c_void_p
        _lib.orc_scene_terrain.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_scene_grid.restype = C.c_void_p
        _lib.orc_scene_grid.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_scene_grid2.restype = C.c_void_p
        _lib.orc_scene_grid2.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_scene_scatter.restype = C.c_void_p
        _lib.orc_scene_scatter.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_scene_free.argtypes = [C.c_void_p]
        _lib.orc_scene_set_camera_size.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _lib.orc_scene_camera.argtypes = [C.c_void_p, _f32p, _u32p]
        _lib.orc_bvh_size.argtypes = [C.c_void_p, C.c_int, _u32p]
        _lib.orc_bvh_dump.argtypes = [C.c_void_p, C.c_int, _u32p, _u32p, _u32p]
        _lib.orc_mesh_data.argtypes = [C.c_void_p, _f32p, C.c_int]
        _lib.orc_trace_closest.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_f32p), _i32p, _u32p, _f32p, _f32p, _f32p,
                                           _f32p, _f32p, _f32p, C.c_int, _u64p]
        _lib.orc_trace_any.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_f32p), _u8p, C.c_int, _u64p]
        _lib.orc_primary_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_f32p)]
        _lib.orc_render.restype = C.c_double
        _lib.orc_render.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _u64p]
        _lib.orc_quantize.argtypes = [_f32p, C.c_size_t, _u8p]
        _lib.orc_pixel_path.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, _f32p, C.c_int, _f32p]
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


COUNTER_NAMES = ("nodes_visited", "tris_tested", "spheres_tested", "cuboids_tested", "closest_rays", "any_rays",
                 "degenerate_tris", "_", "nodes_visited_any", "tris_tested_any", "spheres_tested_any", "cuboids_tested_any")


def load_skybox(max_width=None):
    """Decode assets/skybox.jpg with PIL (RGB8, row-major).  Optionally downscale by integer striding."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    img = np.asarray(Image.open(SKYBOX_JPG).convert("RGB"))
    if max_width is not None and img.shape[1] > max_width:
        step = int(np.ceil(img.shape[1] / max_width))
        img = img[::step, ::step]
    return np.ascontiguousarray(img)


def procedural_sky(w=512, h=256):
    """Deterministic synthetic equirectangular sky (no file needed)."""
    y, x = np.mgrid[0:h, 0:w]
    r = (128 + 100 * np.sin(x * 2 * np.pi / w)).astype(np.uint8)
    g = (60 + 180 * y / h).astype(np.uint8)
    b = (200 - 120 * y / h + 30 * np.cos(x * 6 * np.pi / w)).astype(np.uint8)
    return np.ascontiguousarray(np.stack([r, g, b], axis=-1))


class OracleScene:
    def __init__(self, handle, sky):
        if not handle:
            raise RuntimeError("oracle scene construction failed")
        self.h = C.c_void_p(handle)
        self._sky = sky

    @classmethod
    def main(cls, sky=None, cam=(1080, 960), ply=BUNNY_PLY):
        sky = procedural_sky() if sky is None else np.ascontiguousarray(sky, dtype=np.uint8)
        hnd = lib().orc_scene_main(ply.encode(), sky.ctypes.data, sky.shape[1], sky.shape[0], cam[0], cam[1])
        return cls(hnd, sky)

    @classmethod
    def spheres(cls, n=10000, seed=1, sky=None, cam=(1080, 960), scale=None):
        sky = procedural_sky() if sky is None else np.ascontiguousarray(sky, dtype=np.uint8)
        if scale is not None:
            hnd = lib().orc_scene_spheres_scaled(n, seed, scale, sky.ctypes.data, sky.shape[1], sky.shape[0], cam[0], cam[1])
        else:
            hnd = lib().orc_scene_spheres(n, seed, sky.ctypes.data, sky.shape[1], sky.shape[0], cam[0], cam[1])
        return cls(hnd, sky)

    @classmethod
    def grid(cls, nx, ny, sky=None, cam=(1080, 960), ply=BUNNY_PLY, flat=False):
        sky = procedural_sky() if sky is None else np.ascontiguousarray(sky, dtype=np.uint8)
        hnd = lib().orc_scene_grid2(ply.encode(), nx, ny, int(flat), sky.ctypes.data, sky.shape[1], sky.shape[0], cam[0], cam[1])
        return cls(hnd, sky)

    @classmethod
    def scatter(cls, n, seed=1, accelerate=False, sky=None, cam=(1080, 960)):
        sky = procedural_sky() if sky is None else np.ascontiguousarray(sky, dtype=np.uint8)
        hnd = lib().orc_scene_scatter(n, seed, int(accelerate), sky.ctypes.data, sky.shape[1], sky.shape[0], cam[0], cam[1])
        return cls(hnd, sky)

    @classmethod
    def terrain(cls, n, seed=1, with_bunny=True, sky=None, cam=(1080, 960), ply=BUNNY_PLY):
        sky = procedural_sky() if sky is None else np.ascontiguousarray(sky, dtype=np.uint8)
        hnd = lib().orc_scene_terrain(ply.encode(), n, seed, int(with_bunny), sky.ctypes.data, sky.shape[1], sky.shape[0], cam[0], cam[1])
        return cls(hnd, sky)

    @classmethod
    def textured(cls, sky=None, tex=None, cam=(1080, 960), ply=BUNNY_PLY):
        sky = procedural_sky() if sky is None else np.ascontiguousarray(sky, dtype=np.uint8)
        if tex is None:
            hnd = lib().orc_scene_textured(ply.encode(), sky.ctypes.data, sky.shape[1], sky.shape[0], None, 0, 0, cam[0], cam[1])
        else:
            tex = np.ascontiguousarray(tex, dtype=np.uint8)
            hnd = lib().orc_scene_textured(ply.encode(), sky.ctypes.data, sky.shape[1], sky.shape[0], tex.ctypes.data, tex.shape[1], tex.shape[0], cam[0], cam[1])
        return cls(hnd, sky)

    def __del__(self):
        try:
            lib().orc_scene_free(self.h)
        except Exception:
            pass

    def set_camera_size(self, w, h):
        lib().orc_scene_set_camera_size(self.h, w, h)

    def camera(self):
        out = np.zeros(12, np.float32)
        wh = np.zeros(2, np.uint32)
        lib().orc_scene_camera(self.h, _fp(out), wh.ctypes.data_as(_u32p))
        return out, (int(wh[0]), int(wh[1]))

    def bvh(self, which=0):
        sz = np.zeros(2, np.uint32)
        if lib().orc_bvh_size(self.h, which, sz.ctypes.data_as(_u32p)) != 0:
            return None
        nodes = np.zeros((int(sz[0]), 8), np.uint32)
        leaf = np.zeros(int(sz[1]), np.uint32)
        stats = np.zeros(5, np.uint32)
        lib().orc_bvh_dump(self.h, which, nodes.ctypes.data_as(_u32p), leaf.ctypes.data_as(_u32p), stats.ctypes.data_as(_u32p))
        return nodes, leaf, stats

    def mesh_data(self):
        n = lib().orc_mesh_data(self.h, None, 0)
        out = np.zeros((2 * n, 3), np.float32)
        lib().orc_mesh_data(self.h, _fp(out), out.size)
        return out[:n], out[n:]

    @staticmethod
    def _ray_ptrs(rays):
        assert rays.dtype == np.float32 and rays.shape[0] == 7 and rays.flags.c_contiguous
        return (_f32p * 7)(*[_fp(rays[i]) for i in range(7)])

    def primary_rays(self, x0, y0, w, h):
        rays = np.zeros((7, w * h), np.float32)
        lib().orc_primary_rays(self.h, x0, y0, w, h, self._ray_ptrs(rays))
        return rays

    def trace_closest(self, rays, threads=None, full=False):
        n = rays.shape[1]
        threads = threads or os.cpu_count() or 1
        obj = np.zeros(n, np.int32); prim = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        tl = np.zeros(n, np.float32); pos = np.zeros((n, 3), np.float32); nrm = np.zeros((n, 3), np.float32)
        cnt = np.zeros(12, np.uint64)
        lib().orc_trace_closest(self.h, n, self._ray_ptrs(rays), obj.ctypes.data_as(_i32p), prim.ctypes.data_as(_u32p),
                                _fp(t), _fp(u), _fp(v), _fp(tl), _fp(pos), _fp(nrm), threads, cnt.ctypes.data_as(_u64p))
        res = dict(obj=obj, prim=prim, t=t, u=u, v=v, counters=dict(zip(COUNTER_NAMES, cnt.tolist())))
        if full:
            res.update(t_local=tl, pos=pos, norm=nrm)
        return res

    def trace_any(self, rays, threads=None):
        n = rays.shape[1]
        threads = threads or os.cpu_count() or 1
        out = np.zeros(n, np.uint8)
        cnt = np.zeros(12, np.uint64)
        lib().orc_trace_any(self.h, n, self._ray_ptrs(rays), out.ctypes.data_as(_u8p), threads, cnt.ctypes.data_as(_u64p))
        return out, dict(zip(COUNTER_NAMES, cnt.tolist()))

    def render(self, max_depth=5, shadow_rays=1, scatter_rays=1, antialiasing=1, rect=None, threads=None):
        _, (W, H) = self.camera()
        x0, y0, w, h = rect if rect is not None else (0, 0, W, H)
        threads = threads or os.cpu_count() or 1
        out = np.zeros((h, w, 3), np.float32)
        params = (C.c_int * 4)(max_depth, shadow_rays, scatter_rays, antialiasing)
        cnt = np.zeros(12, np.uint64)
        sec = lib().orc_render(self.h, params, x0, y0, w, h, _fp(out), threads, cnt.ctypes.data_as(_u64p))
        return out, sec, dict(zip(COUNTER_NAMES, cnt.tolist()))


def pixel_path(scene, cx, cy, max_depth=5, shadow_rays=1, scatter_rays=1, antialiasing=1, max_records=4096):
    """Diagnostic: the closest-hit rays of one pixel in the order the recursion traces them, as rows of
    (depth, object or -1, primitive, material kind or -1, t, origin xyz, direction xyz), and the pixel's colour."""
    rec = np.zeros((max_records, 11), np.float32)
    rgb = np.zeros(3, np.float32)
    params = (C.c_int * 4)(max_depth, shadow_rays, scatter_rays, antialiasing)
    n = lib().orc_pixel_path(scene.h, params, cx, cy, _fp(rec), max_records, _fp(rgb))
    return rec[:min(n, max_records)], rgb


def quantize(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    out = np.zeros(rgb.shape, np.uint8)
    lib().orc_quantize(_fp(rgb), rgb.size, out.ctypes.data_as(_u8p))
    return out


# ---- small known-answer probes -----------------------------------------------------------------------
def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def kat_moller_trumbore(tri, pos, d):
    out = (C.c_float * 3)()
    lib().orc_kat_moller_trumbore((C.c_float * 9)(*[float(x) for x in np.ravel(tri)]), _f3(pos), _f3(d), out)
    return tuple(out)


def kat_triangle_hit(tri, pos, d, t_max=1e12):
    out = (C.c_float * 3)()
    lib().orc_kat_triangle_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    hit = lib().orc_kat_triangle_hit((C.c_float * 9)(*[float(x) for x in np.ravel(tri)]), _f3(pos), _f3(d), t_max, out)
    return bool(hit), tuple(out)


def kat_aabb(mn, mx, pos, d):
    out = (C.c_float * 2)()
    hit = lib().orc_kat_aabb(_f3(mn), _f3(mx), _f3(pos), _f3(d), out)
    return bool(hit), tuple(out)


def kat_sphere(center, radius, pos, d, t_max=1e12):
    out = (C.c_float * 5)()
    lib().orc_kat_sphere.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    hit = lib().orc_kat_sphere(_f3(center), radius, _f3(pos), _f3(d), t_max, out)
    return bool(hit), dict(t=out[0], pos=(out[1], out[2], out[3]), any=bool(out[4]))


def kat_cuboid(a, b, pos, d, t_max=1e12):
    out = (C.c_float * 8)()
    lib().orc_kat_cuboid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    hit = lib().orc_kat_cuboid(_f3(a), _f3(b), _f3(pos), _f3(d), t_max, out)
    return bool(hit), dict(t=out[0], pos=tuple(out[1:4]), norm=tuple(out[4:7]), any=bool(out[7]))


def kat_mesh_hit(verts, faces, pos, d, t_max=1e12):
    verts = np.ascontiguousarray(verts, np.float32)
    faces = np.ascontiguousarray(faces, np.uint32)
    prim, face = C.c_uint32(), C.c_uint32()
    tuv = (C.c_float * 3)()
    lib().orc_kat_mesh_hit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    hit = lib().orc_kat_mesh_hit(verts.ctypes.data, len(verts), faces.ctypes.data, len(faces), _f3(pos), _f3(d), t_max,
                                 C.byref(prim), C.byref(face), tuv)
    return bool(hit), dict(prim=prim.value, face=face.value, t=tuv[0], u=tuv[1], v=tuv[2])


def kat_rng(seed, n):
    u = np.zeros(n, np.uint32)
    f = np.zeros(n, np.float32)
    lib().orc_kat_rng.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    lib().orc_kat_rng(seed, n, u.ctypes.data, f.ctypes.data)
    return u, f


def kat_ply(path, n_first=4):
    counts = np.zeros(2, np.uint32)
    faces = np.zeros((n_first, 3), np.uint32)
    lib().orc_kat_ply.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]
    rc = lib().orc_kat_ply(path.encode(), counts.ctypes.data, faces.ctypes.data, n_first)
    return rc, counts, faces


def kat_rotation(axis, angle):
    out = np.zeros(9, np.float32)
    lib().orc_kat_rotation.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
    lib().orc_kat_rotation(_f3(axis), angle, out.ctypes.data)
    return out.reshape(3, 3)


def kat_texture(kind, u, v, param=0, image=None):
    """Lambartian(texture).sample_at(u, v); kind 0 colour, 1 UV, 2 checkerboard(param), 3 image (H x W x 3 uint8)."""
    out = np.zeros(3, np.float32)
    lib().orc_kat_texture.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
    if image is None:
        lib().orc_kat_texture(kind, param, None, 0, 0, u, v, out.ctypes.data)
    else:
        image = np.ascontiguousarray(image, dtype=np.uint8)
        lib().orc_kat_texture(kind, param, image.ctypes.data, image.shape[1], image.shape[0], u, v, out.ctypes.data)
    return out
