/*
 * rtamd.h -- C ABI of the MI355X (gfx950) render hot path.
 *
 * This is the boundary a Rust `-sys` crate of the reference renderer would bind: the host keeps its
 * Scene / Integrator / render entry points and its own BVH builder, flattens its scene into the
 * plain arrays below, and calls these functions instead of walking the scene on the CPU.
 *
 * Conventions
 *   - every function returns 0 on success and a non-zero RTAMD_E_* code on failure; the message is
 *     available from rtamd_last_error() (context-owned string, valid until the next call);
 *   - nothing aborts or throws across the boundary;
 *   - all arrays passed to rtamd_upload_scene() stay owned by the caller and are copied;
 *   - calls are blocking unless they take a stream; a context is bound to one GPU and is not
 *     re-entrant (use one host thread per context);
 *   - all floating point data is IEEE binary32, indices are uint32.
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTAMD_MISS 0xFFFFFFFFu

enum rtamd_error {
    RTAMD_OK = 0,
    RTAMD_E_INVALID = 1,   /* bad argument / inconsistent scene description */
    RTAMD_E_DEVICE = 2,    /* HIP runtime error */
    RTAMD_E_NOSCENE = 3,   /* trace/render before a scene was uploaded */
    RTAMD_E_UNSUPPORTED = 4 /* a valid request beyond a fixed limit of the device code (max_depth above 64) */
};

typedef struct rtamd_context rtamd_context;

/* ---- scene description ------------------------------------------------------------------------ */

/* One node of the reference's flat pre-order tree, `(AABB, skip, first)`:
 * replaces the element type of AABBTree::nodes, src/geom/bvh.rs:5-8.  The array includes the
 * trailing sentinel node `(empty, m, n)` pushed at src/geom/bvh.rs:164-165. */
typedef struct rtamd_bvh_node {
    float min[3];
    float max[3];
    uint32_t skip;
    uint32_t first;
} rtamd_bvh_node;

/* Mesh = AABBTree<MeshTriangle> over a shared [vertices || normals] array:
 * replaces struct Mesh / MeshTriangle, src/geom/mesh.rs:11-52.  `tri_indices` holds the three
 * vertex indices of every triangle in TREE (leaf) order, i.e. AABBTree::objs after construction. */
typedef struct rtamd_mesh {
    const rtamd_bvh_node* nodes;
    uint32_t n_nodes;
    const uint32_t* tri_indices;
    uint32_t n_tris;
    const float* vertices; /* 3 floats per vertex */
    const float* normals;  /* 3 floats per vertex */
    uint32_t n_verts;
} rtamd_mesh;

/* Cuboid{center, extent, inv_extent}: replaces struct Cuboid, src/geom/cuboid.rs:61-79. */
typedef struct rtamd_cuboid {
    float center[3];
    float extent[3];
    float inv_extent[3];
} rtamd_cuboid;

/* Standalone triangle: replaces struct Triangle, src/geom/triangle.rs:5-16. */
typedef struct rtamd_triangle {
    float a[3], b[3], c[3];
} rtamd_triangle;

/* One ray-space wrapper, applied outermost first: replaces Translate / Scale / Rotate,
 * src/geom/transform.rs:5-22.
 *   TRANSLATE: v[0..2] = offset            (transform.rs:55-72)
 *   SCALE:     v[0] = scale, v[1] = inv    (transform.rs:100-120)
 *   ROTATE:    v[0..8] = row-major 3x3     (transform.rs:175-198)
 *   BOUNDS:    v[0..2] = min, v[3..5] = max of the padded box of a BoundingBox<T> (aggregate.rs:59-96):
 *              the wrapped object is only tried by rays whose slab interval (t_in, t_out) of the box
 *              exists with t_in <= t_max and t_out >= 0 */
enum rtamd_op_kind { RTAMD_OP_TRANSLATE = 0, RTAMD_OP_SCALE = 1, RTAMD_OP_ROTATE = 2, RTAMD_OP_BOUNDS = 3 };
typedef struct rtamd_xform_op {
    uint32_t kind;
    float v[9];
} rtamd_xform_op;

/* Geometry primitive kinds an object can wrap. */
enum rtamd_geom_kind {
    RTAMD_GEOM_MESH = 0,        /* geom_index -> meshes[]          (mesh.rs) */
    RTAMD_GEOM_UNIT_SPHERE = 1, /* geom_index ignored              (sphere.rs:35-74); Sphere = ops {TRANSLATE, SCALE} */
    RTAMD_GEOM_CUBOID = 2,      /* geom_index -> cuboids[]         (cuboid.rs:81-105) */
    RTAMD_GEOM_UNIT_CUBOID = 3, /* geom_index ignored              (cuboid.rs:18-59) */
    RTAMD_GEOM_TRIANGLE = 4     /* geom_index -> triangles[]       (triangle.rs:40-69) */
};

/* Object = geometry + material: replaces ObjectImpl / Object, src/geom/aggregate.rs:98-148.
 * ops[op_first .. op_first+op_count) are the wrappers between the object and its primitive. */
typedef struct rtamd_object {
    uint32_t geom_kind;
    uint32_t geom_index;
    uint32_t material;
    uint32_t op_first;
    uint32_t op_count;
} rtamd_object;

/* AABBTree<Object> (src/geom/bvh.rs:173): leaves are objects[first_object + k], k in tree order. */
typedef struct rtamd_object_tree {
    const rtamd_bvh_node* nodes;
    uint32_t n_nodes;
    uint32_t first_object;
    uint32_t n_objects;
} rtamd_object_tree;

/* Scene root = GeometryList in order: replaces GeometryList, src/geom/aggregate.rs:23-45. */
enum rtamd_root_kind { RTAMD_ROOT_OBJECT = 0, RTAMD_ROOT_OBJECT_TREE = 1 };
typedef struct rtamd_root_item {
    uint32_t kind;
    uint32_t index; /* objects[] or trees[] */
} rtamd_root_item;

/* Textures: replaces the Texture impls of src/texture.rs:10-47,70-82.
 *   COLOR: the material's `color`; UV: (u, v, 0); CHECKERBOARD(tex_param = repeats);
 *   IMAGE(tex_param = index into rtamd_scene_desc::images, nearest texel). */
enum rtamd_texture_kind { RTAMD_TEX_COLOR = 0, RTAMD_TEX_UV = 1, RTAMD_TEX_CHECKERBOARD = 2, RTAMD_TEX_IMAGE = 3 };

/* Materials: replaces the Material impls of src/material.rs:17-141.
 *   NULL: green diffuse; METAL; GLOSSY(p0 = exponent, p1 = mirror fraction, texture);
 *   TRANSPARENT(p0 = index of refraction; Glass = 1.5, material.rs:72-78); LAMBERT(texture). */
enum rtamd_material_kind { RTAMD_MAT_NULL = 0, RTAMD_MAT_METAL = 1, RTAMD_MAT_GLOSSY = 2, RTAMD_MAT_TRANSPARENT = 3, RTAMD_MAT_LAMBERT = 4 };
typedef struct rtamd_material {
    uint32_t kind;
    float p0, p1;
    float color[3];
    uint32_t tex_kind;   /* rtamd_texture_kind of Lambartian<T> / Glossy<T> */
    int32_t tex_param;
} rtamd_material;

/* Decoded RGB8 image of a texture::Image, src/texture.rs:49-68 (row-major, 3 bytes per texel). */
typedef struct rtamd_image {
    const uint8_t* rgb;
    uint32_t w, h;
} rtamd_image;

/* Lights: replaces the Light impls of src/light.rs:12-158.  emission = color * intensity.
 *   AMBIENT; DIRECTION(vec = normalised direction, param = spread, <= 0: none);
 *   AO(param = distance); POINT(vec = position, param = radius of the emitting ball, 0 = a true point). */
enum rtamd_light_kind { RTAMD_LIGHT_AMBIENT = 0, RTAMD_LIGHT_POINT = 1, RTAMD_LIGHT_DIRECTION = 2, RTAMD_LIGHT_AO = 3 };
typedef struct rtamd_light {
    uint32_t kind;
    float vec[3];
    float param;
    float emission[3];
} rtamd_light;

/* Whole scene: replaces struct Scene {root, skybox, lights}, src/scene.rs:8-13 (the camera travels
 * with the render call).  skybox_rgb is the decoded RgbImage of texture::Image, src/texture.rs:49. */
typedef struct rtamd_scene_desc {
    const rtamd_mesh* meshes;           uint32_t n_meshes;
    const rtamd_cuboid* cuboids;        uint32_t n_cuboids;
    const rtamd_triangle* triangles;    uint32_t n_triangles;
    const rtamd_xform_op* ops;          uint32_t n_ops;
    const rtamd_object* objects;        uint32_t n_objects;
    const rtamd_object_tree* trees;     uint32_t n_trees;
    const rtamd_root_item* root;        uint32_t n_root;
    const rtamd_material* materials;    uint32_t n_materials;
    const rtamd_light* lights;          uint32_t n_lights;
    const uint8_t* skybox_rgb;          uint32_t sky_w, sky_h;
    const rtamd_image* images;          uint32_t n_images;   /* image textures of the materials (may be empty) */
} rtamd_scene_desc;

/* ---- ray batches ------------------------------------------------------------------------------- */

/* Structure-of-arrays ray batch: replaces `&Ray` + `t_max` of Geometry::hit / is_hit,
 * src/geom/mod.rs:30-33, one element per ray. */
typedef struct rtamd_rays {
    const float *ox, *oy, *oz; /* Ray::pos */
    const float *dx, *dy, *dz; /* Ray::dir */
    const float* tmax;
} rtamd_rays;

/* Closest-hit results: replaces Option<HitResult>, src/geom/mod.rs:20-26.
 * obj = index into objects[] (RTAMD_MISS on a miss); prim = index of the primitive inside the
 * object's innermost tree in tree order (0 for non-mesh objects); t is in the caller's space. */
typedef struct rtamd_hits {
    uint32_t* obj;
    uint32_t* prim;
    float *t, *u, *v;
} rtamd_hits;

/* ---- render ------------------------------------------------------------------------------------ */

/* Camera: replaces struct Camera after its builder ran, src/scene.rs:25-32. */
typedef struct rtamd_camera {
    float pos[3], dir[3], horizontal[3], vertical[3];
    uint32_t width, height;
} rtamd_camera;

/* Integrator parameters: replaces WhittedIntegrator, src/integrator.rs:12-19 (gamma is unused by
 * the reference and absent here).  scatter_rays = 1 (what src/main.rs uses) runs the wavefront
 * pipeline; any other value runs a depth-first kernel, one pixel per lane, which keeps the order in
 * which the recursion of src/integrator.rs:90-102 draws from the pixel's random stream. */
typedef struct rtamd_integrator {
    int32_t max_depth, shadow_rays, scatter_rays, antialiasing;
} rtamd_integrator;

/* The pixels one call renders: the rectangle (x0,y0,w,h) of the camera image cut into tiles of
 * tile_w x tile_h (row-major tile order, edge tiles clipped); this call owns tiles
 * tile_first, tile_first + tile_stride, ...  With tile_stride = 1 the whole rectangle is rendered.
 * Output layout: the owned tiles one after another, each a dense row-major tile_h x tile_w
 * block (clipped tiles keep the full block size; pixels outside the rectangle are zero). */
typedef struct rtamd_region {
    uint32_t x0, y0, w, h;
    uint32_t tile_w, tile_h;
    uint32_t tile_first, tile_stride;
} rtamd_region;

typedef struct rtamd_stats {
    uint64_t closest_rays; /* Scene::root.hit calls, src/integrator.rs:64 */
    uint64_t any_rays;     /* Scene::root.is_hit calls, src/integrator.rs:124 */
    uint64_t paths;        /* integrate_recur(depth 0) calls, src/integrator.rs:45 */
    uint64_t launches;     /* kernels launched */
    double device_ms;      /* HIP-event time of the whole render on the context's stream */
    /* filled when profiling is enabled: trace launches that carry closest-hit rays (they also carry the previous
     * bounce's shadow rays), the shadow-only launch that ends a sample, shade + resolve launches */
    double trace_closest_ms, trace_any_ms, shade_ms;
} rtamd_stats;

/* ---- entry points ------------------------------------------------------------------------------ */

/* Create / destroy a context on one GPU.  No counterpart in the reference (it has no device). */
int rtamd_create(int device_id, rtamd_context** out);
void rtamd_destroy(rtamd_context* ctx);
const char* rtamd_last_error(const rtamd_context* ctx);

/* Copies the scene to the GPU in the device layout.  Replaces building `Scene` in
 * src/main.rs:39-83 for the device side. */
int rtamd_upload_scene(rtamd_context* ctx, const rtamd_scene_desc* scene);

/* AABBTree::new / with_partition + partition_sah (src/geom/bvh.rs:10-177) on the GPU: the binned-SAH tree over n
 * primitive boxes (6 floats each: min xyz, max xyz; host memory), bit-identical to the reference's sequential builder
 * for finite boxes (node boxes, skip links, leaf ranges and the order of the primitives inside the leaves, split-cost
 * expression as written included).  nodes_out needs room for 2 n records (at most 2 n - 1 nodes plus the sentinel of
 * src/geom/bvh.rs:164-165); *n_nodes_out receives the count including the sentinel; order_out[k] = index of the input
 * box that is the k-th primitive in tree order (AABBTree::objs after construction).  For trees where the host build
 * dominates (millions of primitives); needs no uploaded scene. */
int rtamd_build_bvh(rtamd_context* ctx, const float* boxes, uint32_t n, float hit_cost, rtamd_bvh_node* nodes_out,
                    uint32_t* n_nodes_out, uint32_t* order_out);

/* Batched Geometry::hit on the scene root (src/geom/mod.rs:30, caller src/integrator.rs:64).
 * Host pointers; copies in and out. */
int rtamd_trace_closest(rtamd_context* ctx, const rtamd_rays* rays, size_t n, const rtamd_hits* out);
/* Batched Geometry::is_hit on the scene root (src/geom/mod.rs:31-33, caller src/integrator.rs:124).
 * out[i] = 1 if occluded. */
int rtamd_trace_any(rtamd_context* ctx, const rtamd_rays* rays, size_t n, uint8_t* out);

/* Same two calls on arrays that already live in this GPU's memory; asynchronous on `stream`
 * (a hipStream_t, NULL = the context's own stream, in which case the call blocks until done).
 * Every call takes its queue-head word in turn from a ring of 64 words owned by the context: at most 64 such calls of
 * one context may be in flight at a time (a 65th that overlaps the first shares its head word with it). */
int rtamd_trace_closest_device(rtamd_context* ctx, const rtamd_rays* rays, size_t n, const rtamd_hits* out, void* stream);
int rtamd_trace_any_device(rtamd_context* ctx, const rtamd_rays* rays, size_t n, uint8_t* out, void* stream);

/* rtamd_trace_closest_device for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  The call allocates and frees nothing and never
 * synchronises with the host when `stream` is given: the only device memory it uses beyond `rays` and `out` is one
 * queue-head word taken in turn from a ring of 64 words in a static array of the library's own image (256 bytes,
 * held by every GPU the library runs on until the process ends), zeroed on the stream right before the kernel.
 * Limits: n <= RTAMD_STATIC_MAX_RAYS (RTAMD_E_INVALID beyond it); the ring is shared by all contexts of one GPU, so at
 * most 64 such calls may be in flight on one GPU at a time -- a 65th that overlaps the first races with it on its
 * head word.  Results are those of rtamd_trace_closest_device, bit for bit. */
/* (the library keeps the ring's address per GPU in a table of 64 entries: on a context created for a device index of 64 or more
 * this entry point returns RTAMD_E_INVALID; every other entry point works there) */
#define RTAMD_STATIC_MAX_RAYS 0x7FFFFFFFull
int rtamd_trace_closest_static(rtamd_context* ctx, const rtamd_rays* rays, size_t n, const rtamd_hits* out, void* stream);

/* Number of floats-triples rtamd_render writes for a region (owned tiles x tile_w x tile_h). */
size_t rtamd_region_pixels(const rtamd_region* region);

/* render::parallel_render (src/render.rs:10-56) for the owned tiles of `region`:
 * out_rgb receives rtamd_region_pixels() x 3 floats (linear, un-quantised, as
 * WhittedIntegrator::calculate_pixel returns them, src/integrator.rs:33-56).
 * rtamd_render writes host memory; rtamd_render_device writes this GPU's memory and blocks until
 * the frame is complete. */
int rtamd_render(rtamd_context* ctx, const rtamd_camera* cam, const rtamd_integrator* integ,
                 const rtamd_region* region, float* out_rgb, rtamd_stats* stats);
int rtamd_render_device(rtamd_context* ctx, const rtamd_camera* cam, const rtamd_integrator* integ,
                        const rtamd_region* region, float* out_rgb_device, rtamd_stats* stats);

/* render::parallel_render (src/render.rs:10-56) of one frame over several GPUs, the form `render.rs` would call on a
 * multi-GPU node: the rectangle (x0, y0, w, h) of the camera image is cut into tile_w x tile_h tiles that are dealt
 * round-robin to the n contexts (normally one per GPU; several contexts on one GPU work too), each context renders
 * its tiles on a host thread of its own, and the tiles land row-major in out_rgb (w*h*3 floats, host memory).  Every
 * context must hold the same uploaded scene.  Pixels are identical to a single-context render: a pixel's random stream
 * depends only on its coordinates (src/integrator.rs:36).  stats: ray, path and launch counts summed over the
 * contexts, the *_ms fields the maximum.  On failure the code of the first failing context is returned and its message
 * is readable from that context (and from ctxs[0]). */
int rtamd_render_multi(rtamd_context* const* ctxs, int n, const rtamd_camera* cam, const rtamd_integrator* integ,
                       uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t tile_w, uint32_t tile_h,
                       float* out_rgb, rtamd_stats* stats);

/* render::parallel_render_image quantiser (src/render.rs:62-72): floor(c*256) clamped to [0,255],
 * on device memory, n = number of floats. */
int rtamd_quantize_device(rtamd_context* ctx, const float* rgb_device, size_t n, uint8_t* out_device);

/* Tuning / introspection. */
enum rtamd_option {
    RTAMD_OPT_PROFILE = 0,      /* 1: time each stage with HIP events (slower; fills *_ms) */
    RTAMD_OPT_MAX_PATHS = 1,    /* pixels processed per wavefront chunk */
    RTAMD_OPT_LDS_SCENE = 2,    /* 1 (default): stage small meshes in LDS; 0: always read through L2 */
    RTAMD_OPT_SCHEDULER = 3,    /* batch trace entry points: 1 (default) persistent phase-voting kernel, 0 plain nested loops */
    RTAMD_OPT_CHUNK = 4,        /* rays a wave claims per queue-head atomic (power of two, 64..4096, default 64) */
    RTAMD_OPT_BLOCKS_PER_CU = 5,/* persistent 512-thread blocks per CU (1..8, default 2) */
    RTAMD_OPT_STREAMS = 6,      /* render: pixel groups in flight at once, each on its own HIP stream (1..8, default 1) */
    RTAMD_OPT_MIN_BATCH = 8,    /* rays per wave below which a short ray queue is not spread over more waves (1..64, default 4) */
    RTAMD_OPT_STATIC_SHARE = 7, /* eighths (0..8) of a ray queue dealt to the waves up front; the rest is claimed dynamically.  Unless set,
                                   rtamd_upload_scene picks 6, or 3 for a scene with an object tree of >= 256 nodes */
    RTAMD_OPT_DEPTH_FIRST_ROUNDS = 9, /* scatter_rays != 1: 1 (default) rounds of one trace launch over every pixel's next ray, 0 one lane per pixel */
    RTAMD_OPT_TREE_DEEPENING = 10, /* object trees of >= 256 nodes walked in passes of growing reach instead of once in the
                                     reference's order, which is not front to back: 0 never; 1 (default) in rtamd_render*; 2 also in the
                                     batch entry points.  Same hits, with one exception: the reference reports a spurious hit at t = 0
                                     for a triangle its ray is parallel to within ~1e-7 rad (triangle.rs:26-28), wherever along the ray
                                     that triangle lies; a walk in passes reports the ray's real nearest hit when it finds one before
                                     its reach includes that triangle (measured: 5 of 16 million grazing-heavy rays through the 230 x 230 grid,
                                     none in 48 million rays through three other scenes) */
    RTAMD_OPT_LEAN_KERNELS = 11,  /* 1 (default): a scene that holds only meshes and cuboids (or only spheres and cuboids) under
                                     Translate / Scale / Rotate chains runs traversal kernels compiled without the code for the other
                                     object kinds and wrappers; 0: always the general kernels.  Same results, bit for bit */
    RTAMD_OPT_ORDERED_WALK = 12,  /* mesh trees read from global memory, and object trees of meshes / spheres / triangles in scenes too
                                     large for the LDS tables, walked nearer child first from 64-byte two-child records with a per-lane
                                     stack in LDS, instead of in the reference's fixed order (src/geom/bvh.rs:227-252): 0 never; 1
                                     (default) in rtamd_render*; 2 also in the batch entry points.  Same hits (equal t resolved by tree
                                     order as the reference does), occlusion queries exactly, closest hits with two exceptions: the
                                     spurious t = 0 hit described under RTAMD_OPT_TREE_DEEPENING, and, in object trees only, two objects
                                     hit within one rounding step of each other in a different object's scale (the reference compares
                                     each object's t in that object's own space, so the winner can depend on which is tested first) */
    RTAMD_OPT_ORDERED_STACK = 14, /* entries (0..15, default 15 = all the kernel holds: 10 for mesh trees in global memory, for object
                                     trees 8 beside them, 14 in scenes without meshes, 6 beside a mesh staged in LDS) of the per-lane stack
                                     an ordered walk may use; where it is full, the subtree at hand is walked in the reference's order.
                                     Same hits at any value (a test knob) */
    RTAMD_OPT_FRAME_KERNEL = 15,  /* rtamd_render*: the frame kernel -- every wave owns a set of pixels and runs them from their first camera ray
                                     to their last sample, tracing and shading in turn, in a handful of launches per 256 * waves pixel slots
                                     instead of a chain of launches per sample and bounce.  0 never; 1 wherever it can run (max_depth >= 1,
                                     at most 7 shadow rays per lit hit); 2 (default) where it is the faster form: scatter_rays != 1 on frames
                                     of up to 1.25 Mi pixel slots (1.2-1.5 times the rate of the rounds of launches; with scatter_rays = 1 the
                                     wavefront pipeline stays ahead by 10-15 %).  Same pixels, bit for bit, same ray counts in every mode */
    RTAMD_OPT_FRAME_STEP_MIN = 16, /* frame kernel: resumable pixels (1..64, default 64) from which a wave runs its integrator step
                                     without waiting for lanes to go idle */
    RTAMD_OPT_FRAME_REBALANCE = 17, /* frame kernel: how often (0..6, default 0) the units of 64 pixels are dealt anew to the waves for equal
                                     cost, by the rays and the time the launch before measured: the samples are then rendered in value + 1
                                     launches, [0, 1) [1, 4) [4, 16) ... (from 8 samples per pixel on; from 4 with scatter_rays != 1).  0: one
                                     launch, every wave with units from four places of the chunk.  Measured on the headline frame: 60.5 ms
                                     at 0, 63.3 at 2 (a wave's time is set more by the chains of dependent rays of its deepest pixels,
                                     which are served first, than by its number of rays).  Same pixels at any value */
    RTAMD_OPT_STRICT_ORDER = 13   /* 1: RTAMD_OPT_TREE_DEEPENING = 0 and RTAMD_OPT_ORDERED_WALK = 0 in one call: every entry point visits
                                     nodes, objects and triangles in the reference's order, with no excepted ray; 0: both back to their
                                     defaults */
};
int rtamd_set_option(rtamd_context* ctx, int option, int64_t value);
/* Average duration (ms) and launch count of the closest-hit / any-hit trace kernels since the last
 * reset, measured with HIP events on the stream the kernels ran on (needs RTAMD_OPT_PROFILE). */
int rtamd_kernel_times(rtamd_context* ctx, double* closest_ms_total, uint64_t* closest_launches,
                       double* any_ms_total, uint64_t* any_launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
