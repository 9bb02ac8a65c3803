"""rust-raytracer_amd: MI355X (gfx950) implementation of the rust-raytracer render hot path.

Python here is only a thin ctypes binding over two native libraries that live next to this file:

* ``librtamd.so``       HIP kernels + the C ABI declared in ``include/rtamd.h`` (the drop-in boundary)
* ``librtamd_host.so``  the C++ host side: the reference's scene-building API, its BVH builder and the
                         flattening into the ABI's arrays (``host/rt_host.hpp``)

There is no CPU fallback: every trace / render call goes to the GPU library and raises ``RtamdError``
if the library is missing or reports an error.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_RTAMD = os.environ.get("RTAMD_LIB", os.path.join(PKG_DIR, "librtamd.so"))  # RTAMD_LIB: diagnostic build override
LIB_HOST = os.path.join(PKG_DIR, "librtamd_host.so")
BUNNY_PLY = os.path.join(ROOT, "assets", "bunny.ply")
SKYBOX_JPG = os.path.join(ROOT, "assets", "skybox.jpg")

MISS = 0xFFFFFFFF
OPT_PROFILE, OPT_MAX_PATHS, OPT_LDS_SCENE, OPT_SCHEDULER, OPT_CHUNK, OPT_BLOCKS_PER_CU, OPT_STREAMS, OPT_STATIC_SHARE, OPT_MIN_BATCH = 0, 1, 2, 3, 4, 5, 6, 7, 8
OPT_DEPTH_FIRST_ROUNDS = 9
OPT_TREE_DEEPENING = 10
OPT_LEAN_KERNELS = 11
OPT_ORDERED_WALK = 12
OPT_STRICT_ORDER = 13
OPT_ORDERED_STACK = 14
OPT_FRAME_KERNEL = 15
OPT_FRAME_STEP_MIN = 16
OPT_FRAME_REBALANCE = 17

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


class RtamdError(RuntimeError):
    pass


class Rays(C.Structure):  # rtamd_rays
    _fields_ = [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "tmax")]


class Hits(C.Structure):  # rtamd_hits
    _fields_ = [(n, C.c_void_p) for n in ("obj", "prim", "t", "u", "v")]


class Camera(C.Structure):  # rtamd_camera
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("width", C.c_uint32), ("height", C.c_uint32)]


class Integrator(C.Structure):  # rtamd_integrator
    _fields_ = [("max_depth", C.c_int32), ("shadow_rays", C.c_int32), ("scatter_rays", C.c_int32), ("antialiasing", C.c_int32)]


class Region(C.Structure):  # rtamd_region
    _fields_ = [(n, C.c_uint32) for n in ("x0", "y0", "w", "h", "tile_w", "tile_h", "tile_first", "tile_stride")]


class Stats(C.Structure):  # rtamd_stats
    _fields_ = [("closest_rays", C.c_uint64), ("any_rays", C.c_uint64), ("paths", C.c_uint64), ("launches", C.c_uint64),
                ("device_ms", C.c_double), ("trace_closest_ms", C.c_double), ("trace_any_ms", C.c_double), ("shade_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/rtamd.h declares
ABI_SYMBOLS = (
    "rtamd_create", "rtamd_destroy", "rtamd_last_error", "rtamd_upload_scene", "rtamd_trace_closest", "rtamd_trace_any",
    "rtamd_trace_closest_device", "rtamd_trace_any_device", "rtamd_region_pixels", "rtamd_render", "rtamd_render_device",
    "rtamd_quantize_device", "rtamd_set_option", "rtamd_kernel_times", "rtamd_trace_closest_static", "rtamd_render_multi", "rtamd_build_bvh",
)


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 and the host library (in-tree, next to this file)."""
    out = None if verbose else subprocess.DEVNULL
    args = ["-B"] if force else []
    subprocess.check_call(["make", "-C", os.path.join(PKG_DIR, "csrc"), "-s"] + args, stdout=out)
    subprocess.check_call(["make", "-C", os.path.join(PKG_DIR, "host"), "-s"] + args, stdout=out)


_rt = None
_host = None


def rtamd_lib():
    global _rt
    if _rt is None:
        if not os.path.exists(LIB_RTAMD):
            raise RtamdError(f"{LIB_RTAMD} is missing: build it with rust-raytracer_amd.build() (hipcc --offload-arch=gfx950)")
        _rt = C.CDLL(LIB_RTAMD, mode=C.RTLD_GLOBAL)
        _rt.rtamd_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        _rt.rtamd_destroy.argtypes = [C.c_void_p]
        _rt.rtamd_last_error.restype = C.c_char_p
        _rt.rtamd_last_error.argtypes = [C.c_void_p]
        _rt.rtamd_trace_closest.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.POINTER(Hits)]
        _rt.rtamd_trace_any.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.c_void_p]
        _rt.rtamd_trace_closest_device.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.POINTER(Hits), C.c_void_p]
        _rt.rtamd_trace_any_device.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.c_void_p, C.c_void_p]
        _rt.rtamd_trace_closest_static.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.POINTER(Hits), C.c_void_p]
        _rt.rtamd_region_pixels.restype = C.c_size_t
        _rt.rtamd_region_pixels.argtypes = [C.POINTER(Region)]
        _rt.rtamd_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Integrator), C.POINTER(Region), C.c_void_p, C.POINTER(Stats)]
        _rt.rtamd_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Integrator), C.POINTER(Region), C.c_void_p, C.POINTER(Stats)]
        _rt.rtamd_quantize_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _rt.rtamd_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        _rt.rtamd_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                           C.POINTER(C.c_uint64), C.c_int]
    return _rt


def host_lib():
    global _host
    if _host is None:
        rtamd_lib()
        if not os.path.exists(LIB_HOST):
            raise RtamdError(f"{LIB_HOST} is missing: build it with rust-raytracer_amd.build()")
        _host = C.CDLL(LIB_HOST)
        _host.rth_last_error.restype = C.c_char_p
        _host.rth_scene_main.restype = C.c_void_p
        _host.rth_scene_main.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _host.rth_scene_spheres.restype = C.c_void_p
        _host.rth_scene_spheres.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _host.rth_scene_spheres_scaled.restype = C.c_void_p
        _host.rth_scene_spheres_scaled.argtypes = [C.c_int, C.c_uint64, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _host.rth_scene_terrain.restype = C.c_void_p
        _host.rth_scene_terrain.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _host.rth_scene_textured.restype = C.c_void_p
        _host.rth_scene_textured.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _host.rth_scene_grid.restype = C.c_void_p
        _host.rth_scene_grid.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _host.rth_scene_grid2.restype = C.c_void_p
        _host.rth_scene_grid2.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _host.rth_scene_scatter.restype = C.c_void_p
        _host.rth_scene_scatter.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _host.rth_render_multi.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.POINTER(Stats)]
        _host.rth_ply_info.argtypes = [C.c_char_p, _u32p, _f32p, _u32p, C.c_int]
        _host.rth_save_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        _host.rth_kat_reflection.argtypes = [_f32p, _f32p]
        _host.rth_build_tree.argtypes = [C.c_int, _f32p, C.c_uint32, C.c_float, C.c_void_p, _u32p, _u32p, C.POINTER(C.c_double)]
        _host.rth_scene_free.argtypes = [C.c_void_p]
        _host.rth_scene_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
        _host.rth_scene_upload.restype = C.c_void_p
        _host.rth_scene_upload.argtypes = [C.c_void_p, C.c_int]
        _host.rth_scene_tree_size.argtypes = [C.c_void_p, C.c_int, _u32p]
        _host.rth_scene_tree_dump.argtypes = [C.c_void_p, C.c_int, _u32p, _u32p, _u32p]
        _host.rth_render.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(Stats)]
    return _host


def load_skybox(max_width=None):
    """Decode assets/skybox.jpg (PIL) into an RGB8 array; the reference decodes it with the `image` crate."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    img = np.asarray(Image.open(SKYBOX_JPG).convert("RGB"))
    if max_width is not None and img.shape[1] > max_width:
        step = int(np.ceil(img.shape[1] / max_width))
        img = img[::step, ::step]
    return np.ascontiguousarray(img)


def _ptr(a):
    """Device or host address of a numpy array / torch tensor."""
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


class Scene:
    """One scene script instance (host side) bound to one GPU context after upload()."""

    def __init__(self, handle, sky):
        if not handle:
            raise RtamdError("scene construction failed: " + host_lib().rth_last_error().decode())
        self.h = C.c_void_p(handle)
        self._sky = sky
        self.ctx = None

    @staticmethod
    def _sky_args(sky):
        if sky is None:
            return None, None, 0, 0
        sky = np.ascontiguousarray(sky, dtype=np.uint8)
        return sky, sky.ctypes.data, sky.shape[1], sky.shape[0]

    @classmethod
    def main(cls, sky=None, cam=(1080, 960), ply=BUNNY_PLY):
        """The reference's hard-coded scene (src/main.rs:22-83)."""
        sky, p, w, h = cls._sky_args(sky)
        return cls(host_lib().rth_scene_main(ply.encode(), p, w, h, cam[0], cam[1]), sky)

    @classmethod
    def spheres(cls, n=10000, seed=1, sky=None, cam=(1080, 960), scale=None):
        """n spheres in one object tree over a floor; scale: every length of the scene multiplied by it (f32)."""
        sky, p, w, h = cls._sky_args(sky)
        if scale is not None:
            return cls(host_lib().rth_scene_spheres_scaled(n, seed, scale, p, w, h, cam[0], cam[1]), sky)
        return cls(host_lib().rth_scene_spheres(n, seed, p, w, h, cam[0], cam[1]), sky)

    @classmethod
    def grid(cls, nx, ny, sky=None, cam=(1080, 960), ply=BUNNY_PLY, flat=False):
        """nx x ny instanced bunnies in an object tree; flat = 1 / True: every instance a world-space mesh of its own; flat = 2:
        the instances of each material merged into one world-space mesh (3 meshes, plain root list), trees built on the
        host; flat = 3: the same, trees built by the GPU builder on GPU 0."""
        sky, p, w, h = cls._sky_args(sky)
        return cls(host_lib().rth_scene_grid2(ply.encode(), nx, ny, int(flat), p, w, h, cam[0], cam[1]), sky)

    @classmethod
    def scatter(cls, n, seed=1, accelerate=False, sky=None, cam=(1080, 960)):
        """n spheres / cuboids in a plain GeometryList; accelerate: the list's top-level tree extension."""
        sky, p, w, h = cls._sky_args(sky)
        return cls(host_lib().rth_scene_scatter(n, seed, int(accelerate), p, w, h, cam[0], cam[1]), sky)

    @classmethod
    def terrain(cls, n, seed=1, with_bunny=True, sky=None, cam=(1080, 960), ply=BUNNY_PLY):
        """Heightfield mesh of n x n cells (with coplanar duplicate triangles), optionally with two bunny instances."""
        sky, p, w, h = cls._sky_args(sky)
        return cls(host_lib().rth_scene_terrain(ply.encode(), n, seed, int(with_bunny), p, w, h, cam[0], cam[1]), sky)

    @classmethod
    def textured(cls, sky=None, tex=None, cam=(1080, 960), ply=BUNNY_PLY):
        """UV / checkerboard / image textures, Glass and a standalone triangle in one list (tex: RGB8 array for the
        image-textured sphere; None = the skybox image)."""
        sky, p, w, h = cls._sky_args(sky)
        tex, tp, tw, th = cls._sky_args(tex)
        return cls(host_lib().rth_scene_textured(ply.encode(), p, w, h, tp, tw, th, cam[0], cam[1]), sky)

    def __del__(self):
        try:
            host_lib().rth_scene_free(self.h)
        except Exception:
            pass

    # ---- host-side introspection (no GPU needed) ----
    def camera(self):
        c = Camera()
        host_lib().rth_scene_camera(self.h, C.byref(c))
        return c

    def tree(self, which=0):
        """(nodes[n,8] as u32 words, order[n_prims], stats[5]) of the first mesh tree (0) / object tree (1)."""
        sz = np.zeros(2, np.uint32)
        if host_lib().rth_scene_tree_size(self.h, which, sz.ctypes.data_as(_u32p)) != 0:
            return None
        nodes = np.zeros((int(sz[0]), 8), np.uint32)
        order = np.zeros(int(sz[1]), np.uint32)
        stats = np.zeros(5, np.uint32)
        host_lib().rth_scene_tree_dump(self.h, which, nodes.ctypes.data_as(_u32p), order.ctypes.data_as(_u32p), stats.ctypes.data_as(_u32p))
        return nodes, order, stats

    # ---- GPU ----
    def upload(self, device=0):
        ctx = host_lib().rth_scene_upload(self.h, device)
        if not ctx:
            raise RtamdError("upload failed: " + host_lib().rth_last_error().decode())
        self.ctx = C.c_void_p(ctx)
        return self

    def _check(self, rc, what):
        if rc != 0:
            raise RtamdError(f"{what} failed ({rc}): " + rtamd_lib().rtamd_last_error(self.ctx).decode())

    def _need_ctx(self):
        if self.ctx is None:
            raise RtamdError("scene not uploaded to a GPU")

    def set_option(self, opt, value):
        self._need_ctx()
        self._check(rtamd_lib().rtamd_set_option(self.ctx, opt, int(value)), "rtamd_set_option")

    def kernel_times(self, reset=False):
        self._need_ctx()
        cm, cl, am, al = C.c_double(), C.c_uint64(), C.c_double(), C.c_uint64()
        self._check(rtamd_lib().rtamd_kernel_times(self.ctx, C.byref(cm), C.byref(cl), C.byref(am), C.byref(al), int(reset)), "rtamd_kernel_times")
        return dict(closest_ms=cm.value, closest_launches=cl.value, any_ms=am.value, any_launches=al.value)

    def trace_closest(self, rays):
        """rays: float32 numpy array [7, n] (ox,oy,oz,dx,dy,dz,tmax), host memory.  Geometry::hit batched."""
        self._need_ctx()
        assert rays.dtype == np.float32 and rays.shape[0] == 7 and rays.flags.c_contiguous
        n = rays.shape[1]
        r = Rays(*[rays[i].ctypes.data for i in range(7)])
        obj = np.zeros(n, np.uint32); prim = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float32); u = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
        h = Hits(obj.ctypes.data, prim.ctypes.data, t.ctypes.data, u.ctypes.data, v.ctypes.data)
        self._check(rtamd_lib().rtamd_trace_closest(self.ctx, C.byref(r), n, C.byref(h)), "rtamd_trace_closest")
        return dict(obj=obj, prim=prim, t=t, u=u, v=v)

    def trace_any(self, rays):
        self._need_ctx()
        assert rays.dtype == np.float32 and rays.shape[0] == 7 and rays.flags.c_contiguous
        n = rays.shape[1]
        r = Rays(*[rays[i].ctypes.data for i in range(7)])
        out = np.zeros(n, np.uint8)
        self._check(rtamd_lib().rtamd_trace_any(self.ctx, C.byref(r), n, out.ctypes.data), "rtamd_trace_any")
        return out

    def trace_closest_device(self, rays, hits, n, stream=None):
        """rays: 7 device arrays, hits: 5 device arrays (torch tensors or raw addresses)."""
        self._need_ctx()
        r = Rays(*[_ptr(a) for a in rays])
        h = Hits(*[_ptr(a) for a in hits])
        self._check(rtamd_lib().rtamd_trace_closest_device(self.ctx, C.byref(r), n, C.byref(h), stream), "rtamd_trace_closest_device")

    def trace_closest_static(self, rays, hits, n, stream=None):
        """trace_closest_device through the entry point that allocates nothing (queue head in the library's static array)."""
        self._need_ctx()
        r = Rays(*[_ptr(a) for a in rays])
        h = Hits(*[_ptr(a) for a in hits])
        self._check(rtamd_lib().rtamd_trace_closest_static(self.ctx, C.byref(r), n, C.byref(h), stream), "rtamd_trace_closest_static")

    def trace_any_device(self, rays, out, n, stream=None):
        self._need_ctx()
        r = Rays(*[_ptr(a) for a in rays])
        self._check(rtamd_lib().rtamd_trace_any_device(self.ctx, C.byref(r), n, _ptr(out), stream), "rtamd_trace_any_device")

    @staticmethod
    def make_region(width, height, rect=None, tile=None, tile_first=0, tile_stride=1):
        x0, y0, w, h = rect if rect is not None else (0, 0, width, height)
        tw, th = tile if tile is not None else (w, h)
        return Region(x0, y0, w, h, tw, th, tile_first, tile_stride)

    @staticmethod
    def region_pixels(region):
        return rtamd_lib().rtamd_region_pixels(C.byref(region))

    def render(self, max_depth=5, shadow_rays=1, scatter_rays=1, antialiasing=1, region=None, camera=None, out_device=None):
        """parallel_render for the owned tiles of `region` (default: the whole image as one tile).

        Returns (rgb, stats): rgb is a float32 numpy array [h, w, 3] for a single-tile region, otherwise
        [n_tiles, tile_h, tile_w, 3].  With out_device (a torch tensor / device address) the image is
        written to GPU memory instead and rgb is None."""
        self._need_ctx()
        cam = camera or self.camera()
        region = region or self.make_region(cam.width, cam.height)
        ig = Integrator(max_depth, shadow_rays, scatter_rays, antialiasing)
        st = Stats()
        px = self.region_pixels(region)
        if out_device is not None:
            self._check(rtamd_lib().rtamd_render_device(self.ctx, C.byref(cam), C.byref(ig), C.byref(region), _ptr(out_device), C.byref(st)),
                        "rtamd_render_device")
            return None, st.as_dict()
        out = np.zeros(px * 3, np.float32)
        self._check(rtamd_lib().rtamd_render(self.ctx, C.byref(cam), C.byref(ig), C.byref(region), out.ctypes.data, C.byref(st)), "rtamd_render")
        n_tiles = px // (region.tile_w * region.tile_h)
        out = out.reshape(n_tiles, region.tile_h, region.tile_w, 3)
        if n_tiles == 1:
            out = out[0]
        return out, st.as_dict()


def build_tree(boxes, hit_cost=0.1, device=None):
    """AABBTree::new over boxes[n, 6] (min xyz, max xyz): on the host (device None) or on a GPU (rtamd_build_bvh).
    Returns (nodes[n_nodes, 8] as u32 words incl. the sentinel, order[n], seconds)."""
    boxes = np.ascontiguousarray(boxes, np.float32)
    n = boxes.shape[0]
    nodes = np.zeros((2 * n + 1, 8), np.uint32)
    order = np.zeros(n, np.uint32)
    n_nodes = C.c_uint32()
    sec = C.c_double()
    rc = host_lib().rth_build_tree(-1 if device is None else device, boxes.ctypes.data_as(_f32p), n, hit_cost, nodes.ctypes.data, order.ctypes.data_as(_u32p),
                                   C.byref(n_nodes), C.byref(sec))
    if rc != 0:
        raise RtamdError("build_tree failed: " + host_lib().rth_last_error().decode())
    return nodes[:n_nodes.value], order, sec.value


def render_multi(scene, device_ids, max_depth=5, shadow_rays=1, scatter_rays=1, antialiasing=1):
    """render::parallel_render of the C++ host over several GPU contexts (rtamd_render_multi): one context per entry of
    device_ids (ids may repeat), image tiles dealt round-robin, one host thread per context.  Returns (rgb[h, w, 3], stats)."""
    cam = scene.camera()
    out = np.zeros((cam.height, cam.width, 3), np.float32)
    st = Stats()
    ids = (C.c_int * len(device_ids))(*device_ids)
    params = (C.c_int * 4)(max_depth, shadow_rays, scatter_rays, antialiasing)
    if host_lib().rth_render_multi(scene.h, len(device_ids), ids, params, out.ctypes.data, C.byref(st)) != 0:
        raise RtamdError("render_multi failed: " + host_lib().rth_last_error().decode())
    return out, st.as_dict()


def assemble_tiles(tiles, region_full, tile_first, tile_stride, image=None):
    """Scatter the tile-major output of one rank (tiles [n, th, tw, 3]) into a row-major image [h, w, 3]."""
    w, h, tw, th = region_full.w, region_full.h, region_full.tile_w, region_full.tile_h
    tiles_x = (w + tw - 1) // tw
    tiles_y = (h + th - 1) // th
    if image is None:
        image = np.zeros((h, w, 3), np.float32)
    tiles = np.asarray(tiles).reshape(-1, th, tw, 3)
    for q, t in enumerate(range(tile_first, tiles_x * tiles_y, tile_stride)):
        ty, tx = divmod(t, tiles_x)
        y0, x0 = ty * th, tx * tw
        hh, ww = min(th, h - y0), min(tw, w - x0)
        image[y0:y0 + hh, x0:x0 + ww] = tiles[q, :hh, :ww]
    return image
