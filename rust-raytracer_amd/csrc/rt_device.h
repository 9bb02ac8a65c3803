// Device-side scene layout, exact-order float helpers and the traversal / intersection routines of
// the MI355X render hot path.  Compiled for gfx950 only, with -ffp-contract=off: every float
// operation below is one IEEE binary32 operation, written in the order the reference performs it,
// so hits (object, primitive, t, u, v) are bit-identical to the CPU renderer's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtamd {

// ------------------------------------------------------------------------------------------------
// Device scene layout (HBM).  All trees share one node array, all meshes one triangle array.
//
//   nodes   : 32 B per node = 2 x float4: {min.x min.y min.z max.x} {max.y max.z A B}
//             inner node: A = skip index (relative to the tree), B = 0
//             leaf node : A = LEAF_BIT | first, B = end  (primitive range in tree order; skip = i+1)
//             The reference stores (AABB, skip, first) and reads the next node's `first` as the end
//             of the range (src/geom/bvh.rs:7,232-233); a leaf's skip is always i+1 because it has
//             no children (bvh.rs:148), and an inner node's range is empty, so one word per case
//             is enough and the dependent load of nodes[i+1] disappears.
//   tris    : 48 B per triangle = 3 x float4: {a.x a.y a.z ab.x} {ab.y ab.z ac.x ac.y} {ac.z i j k}
//             ab = b - a and ac = c - a are the first two operations of moller_trumbore
//             (src/geom/triangle.rs:21-22) and do not depend on the ray, so they are done once on
//             upload; i,j,k are the vertex indices (bit patterns) for the normal lookup.
//   normals : one float4 per vertex.
// ------------------------------------------------------------------------------------------------
static constexpr uint32_t LEAF_BIT = 0x80000000u;
static constexpr uint32_t MISS = 0xFFFFFFFFu;

struct DevMesh { uint32_t node_base, n_nodes, tri_base, normal_base; };   // n_nodes excludes the sentinel
// LDS bytes of a mesh as the scheduler kernels stage it: 32-byte nodes, then per triangle two 16-byte quads and one float, each in a
// plane of its own (the float plane rounded up to a whole quad)
__host__ __device__ inline uint32_t lds_mesh_bytes(uint32_t n_nodes, uint32_t n_tris) { return 32u * n_nodes + 32u * n_tris + ((4u * n_tris + 15u) & ~15u); }
// deepen: bits 0-1 walk in passes of growing reach (rt_sched.h; 1 a tree of meshes, 2 a tree of cheap objects), bit 2 (TREE_ORDERED)
// the tree may be walked nearer child first (every object a mesh, a sphere or a triangle, the tree strictly binary)
struct DevTree { uint32_t node_base, n_nodes, first_object, deepen; };
static constexpr uint32_t TREE_DEEPEN_MASK = 3u, TREE_ORDERED = 4u;
struct DevCuboid { float center[3], extent[3], inv_extent[3]; float pad[3]; };
struct DevTriangle { float a[3], b[3], c[3]; float pad[3]; };
struct DevOp { uint32_t kind; float v[9]; uint32_t pad[2]; };
// 96 bytes = 6 x float4.  Mesh objects carry a copy of their mesh record.  When the wrapper chain of
// the object is a subsequence of Translate -> Scale -> Rotate (outermost first, each at most once:
// what the reference's scenes build), xf_kind has the XF_T / XF_S / XF_R bits and the parameters sit
// in the record itself, so entering the object needs no walk over the op list; any other chain sets
// XF_GENERIC and is walked op by op.  scale = 1 when the chain has no Scale.
struct DevObject {
    uint32_t geom_kind, geom_index, material, xf_kind;
    uint32_t op_first, op_count, node_base, n_nodes;
    uint32_t tri_base; float scale, inv_scale; uint32_t normal_base;
    float off[3]; float rot[9];
    // A padded box around the object in the space of the scene's rays, and the reach of the tree's visit slack in
    // that space's t (0.01 in the object's t, bvh.rs:231, times the chain's scale, plus a margin).  A ray whose slab
    // interval of this box is empty or lies outside [-slack, t_max + slack] cannot pass the object's own first test
    // after the wrappers, so the scheduler kernels skip the object without entering it.  Only objects that are costly
    // to enter and easy to miss (meshes under a rotation) are flagged XF_BOX; the others are entered directly.
    float bmin[3], bmax[3]; float slack; uint32_t pad1;
};
static_assert(sizeof(DevObject) == 128, "object record = 8 float4");
// Instance record of an object, 32 bytes: what distinguishes the object from the others of its class.  A class is a
// DevObject record shared by every object with the same primitive, the same wrapper kinds and the same rotation
// (for instance all copies of one transformed mesh, or all spheres): off = 0, scale / inv_scale unused, material unused,
// and bmin / bmax the object's padded box relative to its offset.  The scheduler kernels enter an object from its
// instance record (one 32-byte read from global memory) and its class (LDS) instead of the 128-byte object record.
struct DevInst { float off[3]; float scale; float inv_scale; uint32_t cls; uint32_t material; uint32_t pad; };
static_assert(sizeof(DevInst) == 32, "instance record = 2 float4");
static constexpr uint32_t CLS_NONE = 0xFFFFFFFFu;   // the object has no class: use its full record
static constexpr uint32_t MAX_CLASSES = 32;
enum : uint32_t { XF_T = 1, XF_S = 2, XF_R = 4, XF_GENERIC = 8, XF_BOX = 16 /* test the object's box (bmin, bmax) before entering */,
                  XF_OUT_INLINE = 32 /* a hit is carried out of the object from the inline parameters as well (reconstruct_hit) */,
                  XF_T_OUT = 64 /* ... adding `off` last: the chain holds a Translate, possibly only one by +0 */ };
struct DevRootItem { uint32_t kind, index; };
struct DevMaterial { uint32_t kind; float p0, p1; float color[3]; uint32_t tex_kind; int32_t tex_param; };
struct DevImage { uint32_t first, w, h, pad; };   // texels at sky[first ...], row-major
enum { TEX_COLOR = 0, TEX_UV = 1, TEX_CHECKERBOARD = 2, TEX_IMAGE = 3 };
struct DevLight { uint32_t kind; float vec[3]; float param; float emission[3]; };

enum { OP_TRANSLATE = 0, OP_SCALE = 1, OP_ROTATE = 2, OP_BOUNDS = 3 };
enum { GEOM_MESH = 0, GEOM_UNIT_SPHERE = 1, GEOM_CUBOID = 2, GEOM_UNIT_CUBOID = 3, GEOM_TRIANGLE = 4 };
// What a scene's objects need besides mesh walks.  The scheduler kernels exist in a general form and in a lean one that is
// compiled for the commonest mix only (see sched_features): code for kinds a scene does not hold costs a skipped region per
// object visit even when no lane enters it.
enum : uint32_t { FEAT_SPHERE = 1u, FEAT_CUBOID = 2u, FEAT_OTHER = 4u /* unit cuboid, standalone triangle */, FEAT_GENERIC_XF = 8u, FEAT_ALL = 15u };
enum { ROOT_OBJECT = 0, ROOT_TREE = 1 };
enum { MAT_NULL = 0, MAT_METAL = 1, MAT_GLOSSY = 2, MAT_TRANSPARENT = 3, MAT_LAMBERT = 4 };
enum { LIGHT_AMBIENT = 0, LIGHT_POINT = 1, LIGHT_DIRECTION = 2, LIGHT_AO = 3 };

struct DevScene {
    const float4* nodes;
    const float4* tris;
    const float4* normals;
    const DevMesh* meshes;
    const DevTree* trees;
    const DevCuboid* cuboids;
    const DevTriangle* triangles;
    const DevOp* ops;
    const DevObject* objects;
    const DevRootItem* root;
    const DevMaterial* materials;
    const DevLight* lights;
    const uint32_t* sky;  // RGBA8, one word per texel; the texels of the image textures follow the skybox
    const DevImage* images;
    uint32_t n_root, n_lights, sky_w, sky_h;
    uint32_t n_objects, n_meshes;
    uint32_t root_first;      // root list = plain objects root_first, root_first + 1, ...: that first index; else MISS
    uint32_t lds_mesh;        // mesh index staged in LDS by the trace kernels, or MISS
    uint32_t lds_bytes;       // bytes of that mesh as global memory holds it (32-byte nodes + 48-byte triangles): what the plain kernels stage
    uint32_t lds_sched_bytes; // bytes of LDS the scheduler kernels need for it (36 bytes per triangle, see LdsMesh)
    uint32_t n_ops, n_cuboids;
    uint32_t any_box;         // some object carries XF_BOX: rays get their reciprocal direction when they are fetched
    uint32_t features;        // FEAT_* bits of the object kinds and wrapper chains present
    uint32_t tab_bytes;       // bytes of LDS for the scene tables (objects, ops, cuboids, root), 0 = keep them in global memory
    // Large scenes (tab_bytes = 0): instance records of all objects in global memory, the classes and the small tables
    // (cuboids, trees, root list) in LDS.  mid_bytes = 0: not available (too many classes or cuboids).
    const DevInst* insts;
    const DevObject* classes;
    uint32_t n_classes, n_trees;
    uint32_t mid_bytes;
    uint32_t deepen;          // object-tree walks of the scheduler kernels are repeated with a growing t_max limit (rt_sched.h)
    uint32_t plain_boxes;     // every box of every mesh tree is finite with min <= max: node_passes_plain may stand in for node_passes
    uint32_t plain_tree_boxes;// the same for the object trees
    // Two-child records of every tree, for walks that visit the nearer child first: record i of a tree (4 x float4 at
    // wide[4 * (node_base + i)], 64 bytes, meaningful for inner nodes only) holds the 32-byte records of node i's two
    // children side by side, left (node i + 1) first; the right child's index is the left child's skip link, or i + 2
    // behind a left leaf.
    const float4* wide;
    uint32_t ordered;         // ORD_MESH: mesh trees in global memory are walked nearer child first; ORD_TREE: object trees flagged TREE_ORDERED too
    uint32_t ordered_cap;     // entries of the ordered walks' per-lane stack a walk may use (at most what the kernel instance holds)
};
static constexpr uint32_t ORD_MESH = 1u, ORD_TREE = 2u;

// ------------------------------------------------------------------------------------------------
// Exact-order float3 helpers (src/math/vec3d.rs:115-126,181-287).
// ------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 fill(float s) { return mk(s, s, s); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, f3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalize(f3 a) {  // vec3d.rs:100-112
    float len = sqrtf(dot(a, a));
    if (len > 0.0f) return a / len;
    return fill(__uint_as_float(0x7FC00000u));
}
__device__ __forceinline__ bool is_zero(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
// max!/min! compare-selects (src/common.rs:31-53); NOT fmaxf/fminf (different NaN behaviour).
__device__ __forceinline__ float max2(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float min2(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ bool signbit32(float x) { return (__float_as_uint(x) >> 31) != 0u; }

__device__ __forceinline__ void ortho_axes(f3 a, f3& b, f3& c) {  // vec3d.rs:128-141
    if (fabsf(a.x) > fabsf(a.y)) {
        float inv_len = 1.0f / sqrtf(a.x * a.x + a.z * a.z);
        b = mk(-a.z * inv_len, 0.0f, a.x * inv_len);
    } else {
        float inv_len = 1.0f / sqrtf(a.y * a.y + a.z * a.z);
        b = mk(0.0f, a.z * inv_len, -a.y * inv_len);
    }
    c = cross(a, b);
}
// Mat3D::apply / transpose_apply (src/math/mat3d.rs:139-169): accumulate from 0.0 in j order.
__device__ __forceinline__ f3 mat_apply(const float* m, f3 a) {
    f3 b;
    b.x = ((0.0f + m[0] * a.x) + m[1] * a.y) + m[2] * a.z;
    b.y = ((0.0f + m[3] * a.x) + m[4] * a.y) + m[5] * a.z;
    b.z = ((0.0f + m[6] * a.x) + m[7] * a.y) + m[8] * a.z;
    return b;
}
__device__ __forceinline__ f3 mat_transpose_apply(const float* m, f3 a) {
    f3 b;
    b.x = ((0.0f + m[0] * a.x) + m[3] * a.y) + m[6] * a.z;
    b.y = ((0.0f + m[1] * a.x) + m[4] * a.y) + m[7] * a.z;
    b.z = ((0.0f + m[2] * a.x) + m[5] * a.y) + m[8] * a.z;
    return b;
}

// ------------------------------------------------------------------------------------------------
// Memory accessors for the node / triangle arrays of one tree: global (L2/HBM) or LDS.
// ------------------------------------------------------------------------------------------------
struct GlobalMem {
    const float4* nodes;
    const float4* tris;
    __device__ __forceinline__ void node(uint32_t i, float4& n0, float4& n1) const { n0 = nodes[2 * i]; n1 = nodes[2 * i + 1]; }
    __device__ __forceinline__ void tri(uint32_t k, float4& q0, float4& q1, float4& q2) const {
        q0 = tris[3 * k]; q1 = tris[3 * k + 1]; q2 = tris[3 * k + 2];
    }
};

// ------------------------------------------------------------------------------------------------
// AABB::fast_intersect_ray (src/math/aabb.rs:61-87) + the visit test of AABBTree::traverse
// (src/geom/bvh.rs:230-231).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool node_passes(float4 n0, float4 n1, f3 pos, f3 inv, bool ngx, bool ngy, bool ngz, float t_max) {
    float ax = (n0.x - pos.x) * inv.x, ay = (n0.y - pos.y) * inv.y, az = (n0.z - pos.z) * inv.z;
    float bx = (n0.w - pos.x) * inv.x, by = (n1.x - pos.y) * inv.y, bz = (n1.y - pos.z) * inv.z;
    float a0 = ngx ? bx : ax, b0 = ngx ? ax : bx;
    float a1 = ngy ? by : ay, b1 = ngy ? ay : by;
    float a2 = ngz ? bz : az, b2 = ngz ? az : bz;
    float t0 = max2(a0, max2(a1, a2));
    float t1 = min2(b0, min2(b1, b2));
    return (t0 < t1) && (t0 <= t_max + 0.01f) && (t1 >= -0.01f);
}

// The same test for a ray and a box that cannot produce a NaN: the box finite with min <= max, the ray's origin finite, its
// reciprocal direction finite and not zero.  The slab products then are numbers (possibly infinite) with a <= b wherever the
// reciprocal is positive and a >= b wherever it is negative, so the swap by sign is the smaller and the larger of the two,
// and without NaNs the reference's `max!` / `min!` (common.rs:32-53) are the plain maximum and minimum; a zero may come out
// with the other sign, which no comparison sees.  Same verdict as node_passes, in 8 instructions instead of 14 and half the
// dependent steps.
__device__ __forceinline__ bool node_passes_plain(float4 n0, float4 n1, f3 pos, f3 inv, float t_max) {
    float ax = (n0.x - pos.x) * inv.x, ay = (n0.y - pos.y) * inv.y, az = (n0.z - pos.z) * inv.z;
    float bx = (n0.w - pos.x) * inv.x, by = (n1.x - pos.y) * inv.y, bz = (n1.y - pos.z) * inv.z;
    // (the instructions by name: the compiler's fminf / fmaxf first pass every operand through a quieting v_max x, x)
    float nx, ny, nz, fx, fy, fz, t0, t1;
    asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(ax), "v"(bx));
    asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ay), "v"(by));
    asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(az), "v"(bz));
    asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(ax), "v"(bx));
    asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ay), "v"(by));
    asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(az), "v"(bz));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(nx), "v"(ny), "v"(nz));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(fx), "v"(fy), "v"(fz));
    return (t0 < t1) && (t0 <= t_max + 0.01f) && (t1 >= -0.01f);
}
// Both tests again, also handing out the entry parameter t0 of the box: what an ordered walk sorts the two children of a
// node by, and what it keeps with a postponed child so that the child is dropped without a read once the ray's t_max has
// shrunk below it (`t0 <= t_max + 0.01` is the only part of the visit test that depends on t_max).
template <bool EXACT>
__device__ __forceinline__ bool node_test(float4 n0, float4 n1, f3 pos, f3 inv, float t_max, float& t0_out) {
    float ax = (n0.x - pos.x) * inv.x, ay = (n0.y - pos.y) * inv.y, az = (n0.z - pos.z) * inv.z;
    float bx = (n0.w - pos.x) * inv.x, by = (n1.x - pos.y) * inv.y, bz = (n1.y - pos.z) * inv.z;
    float t0, t1;
    if (EXACT) {
        const bool ngx = (__float_as_uint(inv.x) >> 31) != 0u, ngy = (__float_as_uint(inv.y) >> 31) != 0u, ngz = (__float_as_uint(inv.z) >> 31) != 0u;
        float a0 = ngx ? bx : ax, b0 = ngx ? ax : bx;
        float a1 = ngy ? by : ay, b1 = ngy ? ay : by;
        float a2 = ngz ? bz : az, b2 = ngz ? az : bz;
        float m = a1 > a2 ? a1 : a2;
        t0 = a0 > m ? a0 : m;
        float w = b1 < b2 ? b1 : b2;
        t1 = b0 < w ? b0 : w;
    } else {
        float nx, ny, nz, fx, fy, fz;
        asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(ax), "v"(bx));
        asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ay), "v"(by));
        asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(az), "v"(bz));
        asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(ax), "v"(bx));
        asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ay), "v"(by));
        asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(az), "v"(bz));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(nx), "v"(ny), "v"(nz));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(fx), "v"(fy), "v"(fz));
    }
    t0_out = t0;
    return (t0 < t1) && (t0 <= t_max + 0.01f) && (t1 >= -0.01f);
}

// Whether node_passes_plain is licensed for a ray: origin finite, reciprocal direction a normal number (a denormal one
// could be flushed to zero and meet an overflowed difference).  Class masks of v_cmp_class_f32: bit 0 sNaN, 1 qNaN, 2 -inf,
// 3 -normal, 4 -denormal, 5 -0, 6 +0, 7 +denormal, 8 +normal, 9 +inf.
__device__ __forceinline__ bool plain_ray(f3 pos, f3 inv) {
    constexpr int NUMBER = 0x1F8, NONZERO = 0x108;
    return __builtin_amdgcn_classf(pos.x, NUMBER) && __builtin_amdgcn_classf(pos.y, NUMBER) && __builtin_amdgcn_classf(pos.z, NUMBER) &&
           __builtin_amdgcn_classf(inv.x, NONZERO) && __builtin_amdgcn_classf(inv.y, NONZERO) && __builtin_amdgcn_classf(inv.z, NONZERO);
}

// moller_trumbore (src/geom/triangle.rs:18-38) on the pre-subtracted edges, followed by the accept
// test of MeshTriangle::hit (src/geom/mesh.rs:130).  A near-parallel ray (|det| < 1e-12) yields
// t = u = v = 0, which the accept test lets through, exactly as the reference does.
__device__ __forceinline__ bool tri_test(float4 q0, float4 q1, float4 q2, f3 pos, f3 dir, float t_max, float& t, float& u, float& v) {
    f3 a = mk(q0.x, q0.y, q0.z), ab = mk(q0.w, q1.x, q1.y), ac = mk(q1.z, q1.w, q2.x);
    f3 pv = cross(dir, ac);
    float det = dot(ab, pv);
    if (fabsf(det) < 1e-12f) {
        t = 0.0f; u = 0.0f; v = 0.0f;
    } else {
        float inv_det = 1.0f / det;
        f3 tv = pos - a;
        f3 qv = cross(tv, ab);
        t = dot(qv, ac) * inv_det;
        u = dot(tv, pv) * inv_det;
        v = dot(qv, dir) * inv_det;
    }
    return t >= 0.0f && t <= t_max && u >= 0.0f && v >= 0.0f && u + v <= 1.0f;
}

// tri_test without control flow: the same operations in the same order, with the near-parallel case selected at the
// end instead of branched around (the quotient of that case is computed and dropped).  For code that tests several
// triangles at once and wants their instruction streams interleaved.
__device__ __forceinline__ bool tri_test_flat(float4 q0, float4 q1, float4 q2, f3 pos, f3 dir, float t_max, float& t, float& u, float& v) {
    const f3 a = mk(q0.x, q0.y, q0.z), ab = mk(q0.w, q1.x, q1.y), ac = mk(q1.z, q1.w, q2.x);
    const f3 pv = cross(dir, ac);
    const float det = dot(ab, pv);
    const bool flat = fabsf(det) < 1e-12f;
    const float inv_det = 1.0f / det;
    const f3 tv = pos - a;
    const f3 qv = cross(tv, ab);
    const float tt = dot(qv, ac) * inv_det, uu = dot(tv, pv) * inv_det, vv = dot(qv, dir) * inv_det;
    t = flat ? 0.0f : tt; u = flat ? 0.0f : uu; v = flat ? 0.0f : vv;
    return (t >= 0.0f) & (t <= t_max) & (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f);
}

// AABBTree<MeshTriangle>::hit / is_hit (src/geom/bvh.rs:207-281): stackless skip-link walk in the
// reference's fixed left-first order.  Closest: t_max shrinks on every accepted triangle, a later
// triangle with the same t replaces an earlier one (`t <= t_max`).  Any: constant t_max, first
// accepted triangle ends the walk.
template <bool ANY, class Mem>
__device__ __forceinline__ bool traverse_mesh(const Mem& mem, uint32_t n_nodes, f3 pos, f3 dir, float& t_max,
                                              uint32_t& prim, float& hu, float& hv) {
    f3 inv = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);  // bvh.rs:220 recomputes 1/dir
    bool ngx = signbit32(inv.x), ngy = signbit32(inv.y), ngz = signbit32(inv.z);
    bool found = false;
    uint32_t i = 0;
    while (i < n_nodes) {
        float4 n0, n1;
        mem.node(i, n0, n1);
        uint32_t A = __float_as_uint(n1.z), B = __float_as_uint(n1.w);
        bool leaf = (A & LEAF_BIT) != 0u;
        if (node_passes(n0, n1, pos, inv, ngx, ngy, ngz, t_max)) {
            if (leaf) {
                for (uint32_t k = A & ~LEAF_BIT; k < B; k++) {
                    float4 q0, q1, q2;
                    mem.tri(k, q0, q1, q2);
                    float t, u, v;
                    if (tri_test(q0, q1, q2, pos, dir, t_max, t, u, v)) {
                        if (ANY) return true;
                        t_max = t; prim = k; hu = u; hv = v; found = true;
                    }
                }
            }
            i += 1;
        } else {
            i = leaf ? i + 1 : A;
        }
    }
    return found;
}

// sphere_intersect + UnitSphere::hit / is_hit (src/geom/sphere.rs:22-69).
__device__ __forceinline__ bool sphere_roots(f3 pos, f3 dir, float& t0, float& t1) {
    float a = -dot(dir, pos);
    float b = a * a - dot(pos, pos) + 1.0f;
    if (b >= 0.0f) {
        float d = sqrtf(b);
        t0 = a - d; t1 = a + d;
        return true;
    }
    return false;
}
__device__ __forceinline__ bool unit_sphere_hit(f3 pos, f3 dir, float t_max, float& t) {
    float t0, t1;
    if (!sphere_roots(pos, dir, t0, t1)) return false;
    if (t0 >= 0.0f && t0 <= t_max) { t = t0; return true; }
    if (t1 >= 0.0f && t1 <= t_max) { t = t1; return true; }
    return false;
}
__device__ __forceinline__ bool unit_sphere_any(f3 pos, f3 dir, float t_max) {
    float t0, t1;
    if (sphere_roots(pos, dir, t0, t1)) return t0 <= t_max && t1 >= 0.0f;
    return false;
}

// ray_box_intersection (src/geom/cuboid.rs:8-16): symmetric min!/max! form.
__device__ __forceinline__ void box_roots(f3 pos, f3 inv_dir, f3 mn, f3 mx, float& t0, float& t1) {
    f3 a = (mn - pos) * inv_dir;
    f3 b = (mx - pos) * inv_dir;
    t0 = max2(min2(a.x, b.x), max2(min2(a.y, b.y), min2(a.z, b.z)));
    t1 = min2(max2(a.x, b.x), min2(max2(a.y, b.y), max2(a.z, b.z)));
}
// UnitCuboid::hit (cuboid.rs:19-48), t only.
__device__ __forceinline__ bool unit_cuboid_hit(f3 pos, f3 dir, float t_max, float& t) {
    f3 inv = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);  // Ray::new, src/math/ray.rs:15
    float t0, t1;
    box_roots(pos, inv, fill(-1.0f), fill(1.0f), t0, t1);
    if (t0 > t1 || t0 > t_max || t1 < 0.0f) return false;
    t = (t0 > 0.0f) ? t0 : t1;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Scene tables (objects, wrapper ops, root list, cuboids) are read through an accessor so that the
// traversal code is the same whether they sit in global memory or in an LDS copy.  Records are
// fetched whole with 16-byte loads (one round trip per record instead of one per field).
// ------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const v4f lds_float4;
__device__ __forceinline__ float4 lds_load4(lds_float4* p) {
    const v4f v = *p;
    return make_float4(v.x, v.y, v.z, v.w);
}
struct OpRec { uint32_t kind; float v[9]; };
struct CuboidRec { f3 center, extent, inv_extent; };

// Head of an object record (the first three float4) and its inline wrapper parameters (the last three).
struct ObjHead {
    uint32_t geom_kind, geom_index, material, xf_kind, op_first, op_count, node_base, n_nodes, tri_base, normal_base;
    float scale, inv_scale;
};
struct ObjXf { f3 off; float rot[9]; };
struct InstRec { f3 off; float scale, inv_scale; uint32_t cls; };
__device__ __forceinline__ InstRec unpack_inst(float4 a, float4 b) {
    InstRec r;
    r.off = mk(a.x, a.y, a.z); r.scale = a.w; r.inv_scale = b.x; r.cls = __float_as_uint(b.y);
    return r;
}
struct ObjBox { f3 mn, mx; float slack; };
__device__ __forceinline__ ObjBox unpack_box(float4 g, float4 h) {
    ObjBox b;
    b.mn = mk(g.x, g.y, g.z); b.mx = mk(g.w, h.x, h.y); b.slack = h.z;
    return b;
}
__device__ __forceinline__ ObjHead unpack_object(float4 a, float4 b, float4 c) {
    ObjHead o;
    o.geom_kind = __float_as_uint(a.x); o.geom_index = __float_as_uint(a.y); o.material = __float_as_uint(a.z); o.xf_kind = __float_as_uint(a.w);
    o.op_first = __float_as_uint(b.x); o.op_count = __float_as_uint(b.y); o.node_base = __float_as_uint(b.z); o.n_nodes = __float_as_uint(b.w);
    o.tri_base = __float_as_uint(c.x); o.scale = c.y; o.inv_scale = c.z; o.normal_base = __float_as_uint(c.w);
    return o;
}
__device__ __forceinline__ ObjXf unpack_xf(float4 d, float4 e, float4 f) {
    ObjXf x;
    x.off = mk(d.x, d.y, d.z);
    x.rot[0] = d.w; x.rot[1] = e.x; x.rot[2] = e.y; x.rot[3] = e.z; x.rot[4] = e.w; x.rot[5] = f.x; x.rot[6] = f.y; x.rot[7] = f.z; x.rot[8] = f.w;
    return x;
}
__device__ __forceinline__ OpRec unpack_op(float4 a, float4 b, float4 c) {
    OpRec o;
    o.kind = __float_as_uint(a.x);
    o.v[0] = a.y; o.v[1] = a.z; o.v[2] = a.w; o.v[3] = b.x; o.v[4] = b.y; o.v[5] = b.z; o.v[6] = b.w; o.v[7] = c.x; o.v[8] = c.y;
    return o;
}
__device__ __forceinline__ CuboidRec unpack_cuboid(float4 a, float4 b, float4 c) {
    CuboidRec r;
    r.center = mk(a.x, a.y, a.z); r.extent = mk(a.w, b.x, b.y); r.inv_extent = mk(b.z, b.w, c.x);
    return r;
}
struct GlobalTab {
    static constexpr bool small_scene = false;
    static constexpr bool has_inst = false;
    const DevScene* S;
    __device__ __forceinline__ DevTree tree(uint32_t i) const { return S->trees[i]; }
    __device__ __forceinline__ InstRec inst(uint32_t) const { InstRec r; r.off = mk(0, 0, 0); r.scale = 1.f; r.inv_scale = 1.f; r.cls = CLS_NONE; return r; }
    __device__ __forceinline__ ObjHead cls_head(uint32_t i) const { return object(i); }
    __device__ __forceinline__ ObjXf cls_xf(uint32_t i) const { return object_xf(i); }
    __device__ __forceinline__ ObjBox cls_box(uint32_t i) const { return object_box(i); }
    __device__ __forceinline__ ObjHead object(uint32_t i) const { const float4* p = (const float4*)(S->objects + i); return unpack_object(p[0], p[1], p[2]); }
    __device__ __forceinline__ ObjXf object_xf(uint32_t i) const { const float4* p = (const float4*)(S->objects + i); return unpack_xf(p[3], p[4], p[5]); }
    __device__ __forceinline__ ObjBox object_box(uint32_t i) const { const float4* p = (const float4*)(S->objects + i); return unpack_box(p[6], p[7]); }
    __device__ __forceinline__ OpRec op(uint32_t i) const { const float4* p = (const float4*)(S->ops + i); return unpack_op(p[0], p[1], p[2]); }
    __device__ __forceinline__ CuboidRec cuboid(uint32_t i) const { const float4* p = (const float4*)(S->cuboids + i); return unpack_cuboid(p[0], p[1], p[2]); }
    __device__ __forceinline__ DevRootItem root(uint32_t i) const { return S->root[i]; }
};
struct LdsTab {
    static constexpr bool small_scene = true;   // the tables fit the LDS budget
    static constexpr bool has_inst = false;
    const DevScene* S;
    __device__ __forceinline__ DevTree tree(uint32_t i) const { return S->trees[i]; }
    __device__ __forceinline__ InstRec inst(uint32_t) const { InstRec r; r.off = mk(0, 0, 0); r.scale = 1.f; r.inv_scale = 1.f; r.cls = CLS_NONE; return r; }
    __device__ __forceinline__ ObjHead cls_head(uint32_t i) const { return object(i); }
    __device__ __forceinline__ ObjXf cls_xf(uint32_t i) const { return object_xf(i); }
    __device__ __forceinline__ ObjBox cls_box(uint32_t i) const { return object_box(i); }
    lds_float4* objects;  // 8 per object
    lds_float4* ops;      // 3 per op
    lds_float4* cuboids;  // 3 per cuboid
    __attribute__((address_space(3))) const uint32_t* roots;  // kind, index pairs
    __device__ __forceinline__ ObjHead object(uint32_t i) const { return unpack_object(lds_load4(objects + 8 * i), lds_load4(objects + 8 * i + 1), lds_load4(objects + 8 * i + 2)); }
    __device__ __forceinline__ ObjXf object_xf(uint32_t i) const { return unpack_xf(lds_load4(objects + 8 * i + 3), lds_load4(objects + 8 * i + 4), lds_load4(objects + 8 * i + 5)); }
    __device__ __forceinline__ ObjBox object_box(uint32_t i) const { return unpack_box(lds_load4(objects + 8 * i + 6), lds_load4(objects + 8 * i + 7)); }
    __device__ __forceinline__ OpRec op(uint32_t i) const { return unpack_op(lds_load4(ops + 3 * i), lds_load4(ops + 3 * i + 1), lds_load4(ops + 3 * i + 2)); }
    __device__ __forceinline__ CuboidRec cuboid(uint32_t i) const { return unpack_cuboid(lds_load4(cuboids + 3 * i), lds_load4(cuboids + 3 * i + 1), lds_load4(cuboids + 3 * i + 2)); }
    __device__ __forceinline__ DevRootItem root(uint32_t i) const { return DevRootItem{roots[2 * i], roots[2 * i + 1]}; }
};

// Large scenes: instance records (32 bytes per object) in global memory; classes, cuboids, object-tree headers and the
// root list in LDS; full object records and wrapper ops (objects without a class, generic chains) in global memory.
struct MidTab {
    static constexpr bool small_scene = false;
    static constexpr bool has_inst = true;
    const DevScene* S;
    lds_float4* classes;  // 8 per class
    lds_float4* cuboids;  // 3 per cuboid
    __attribute__((address_space(3))) const uint32_t* trees;   // node_base, n_nodes, first_object, deepen per tree
    __attribute__((address_space(3))) const uint32_t* roots;   // kind, index pairs
    __device__ __forceinline__ ObjHead object(uint32_t i) const { const float4* p = (const float4*)(S->objects + i); return unpack_object(p[0], p[1], p[2]); }
    __device__ __forceinline__ ObjXf object_xf(uint32_t i) const { const float4* p = (const float4*)(S->objects + i); return unpack_xf(p[3], p[4], p[5]); }
    __device__ __forceinline__ ObjBox object_box(uint32_t i) const { const float4* p = (const float4*)(S->objects + i); return unpack_box(p[6], p[7]); }
    __device__ __forceinline__ OpRec op(uint32_t i) const { const float4* p = (const float4*)(S->ops + i); return unpack_op(p[0], p[1], p[2]); }
    __device__ __forceinline__ CuboidRec cuboid(uint32_t i) const { return unpack_cuboid(lds_load4(cuboids + 3 * i), lds_load4(cuboids + 3 * i + 1), lds_load4(cuboids + 3 * i + 2)); }
    __device__ __forceinline__ DevRootItem root(uint32_t i) const { return DevRootItem{roots[2 * i], roots[2 * i + 1]}; }
    __device__ __forceinline__ DevTree tree(uint32_t i) const { return DevTree{trees[4 * i], trees[4 * i + 1], trees[4 * i + 2], trees[4 * i + 3]}; }
    __device__ __forceinline__ InstRec inst(uint32_t i) const { const float4* p = (const float4*)(S->insts + i); return unpack_inst(p[0], p[1]); }
    __device__ __forceinline__ ObjHead cls_head(uint32_t c) const { return unpack_object(lds_load4(classes + 8 * c), lds_load4(classes + 8 * c + 1), lds_load4(classes + 8 * c + 2)); }
    __device__ __forceinline__ ObjXf cls_xf(uint32_t c) const { return unpack_xf(lds_load4(classes + 8 * c + 3), lds_load4(classes + 8 * c + 4), lds_load4(classes + 8 * c + 5)); }
    __device__ __forceinline__ ObjBox cls_box(uint32_t c) const { return unpack_box(lds_load4(classes + 8 * c + 6), lds_load4(classes + 8 * c + 7)); }
};

// ------------------------------------------------------------------------------------------------
// Ray-space wrappers going in (src/geom/transform.rs:57-58,104-106,178-180).
// ------------------------------------------------------------------------------------------------
// Returns false when a BoundingBox wrapper of the chain turns the ray away (aggregate.rs:75-94).
template <class Tab>
__device__ __forceinline__ bool ops_in(const Tab& tab, uint32_t first, uint32_t count, f3& pos, f3& dir, float& t_max) {
    for (uint32_t k = 0; k < count; k++) {
        const OpRec op = tab.op(first + k);
        if (op.kind == OP_TRANSLATE) {
            pos = pos - mk(op.v[0], op.v[1], op.v[2]);
        } else if (op.kind == OP_SCALE) {
            pos = pos * op.v[1];
            t_max = t_max * op.v[1];
        } else if (op.kind == OP_ROTATE) {
            pos = mat_transpose_apply(op.v, pos);
            dir = mat_transpose_apply(op.v, dir);
        } else {  // OP_BOUNDS: AABB::intersect_ray with the ray of this space (aabb.rs:47-59)
            float t0, t1;
            box_roots(pos, mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z), mk(op.v[0], op.v[1], op.v[2]), mk(op.v[3], op.v[4], op.v[5]), t0, t1);
            if (!(t0 < t1 && t0 <= t_max && t1 >= 0.0f)) return false;
        }
    }
    return true;
}
// Hit distance coming back out: Scale multiplies t (transform.rs:107).
template <class Tab>
__device__ __forceinline__ float ops_t_out(const Tab& tab, uint32_t first, uint32_t count, float t) {
    for (uint32_t k = count; k-- > 0;) {
        const OpRec op = tab.op(first + k);
        if (op.kind == OP_SCALE) t = t * op.v[0];
    }
    return t;
}
// Hit position / normal coming back out (transform.rs:61,108,183-184).
template <class Tab>
__device__ __forceinline__ void ops_hit_out(const Tab& tab, uint32_t first, uint32_t count, f3& pos, f3& nrm) {
    for (uint32_t k = count; k-- > 0;) {
        const OpRec op = tab.op(first + k);
        if (op.kind == OP_TRANSLATE) {
            pos = pos + mk(op.v[0], op.v[1], op.v[2]);
        } else if (op.kind == OP_SCALE) {
            pos = pos * op.v[0];
        } else if (op.kind == OP_ROTATE) {
            pos = mat_apply(op.v, pos);
            nrm = mat_apply(op.v, nrm);
        }
    }
}

// The same wrappers applied from an object record's inline parameters (xf_kind without XF_GENERIC):
// identical operations in identical order, without the walk over the op list.
__device__ __forceinline__ void xf_in_pos(uint32_t xf_kind, const ObjXf& x, float inv_scale, f3& pos, float& t_max) {
    if (xf_kind & XF_T) pos = pos - x.off;
    if (xf_kind & XF_S) { pos = pos * inv_scale; t_max = t_max * inv_scale; }
    if (xf_kind & XF_R) pos = mat_transpose_apply(x.rot, pos);
}
__device__ __forceinline__ f3 xf_in_dir(uint32_t xf_kind, const ObjXf& x, f3 dir) {
    return (xf_kind & XF_R) ? mat_transpose_apply(x.rot, dir) : dir;
}

// ------------------------------------------------------------------------------------------------
// Closest-hit state of one ray.
// ------------------------------------------------------------------------------------------------
struct Best {
    uint32_t obj, prim;
    float t_local, u, v;
};

// Closest-hit test of a non-mesh primitive in object space: UnitSphere::hit (sphere.rs:36-45),
// Cuboid::hit -> UnitCuboid::hit (cuboid.rs:82-93,19-28), Triangle::hit (triangle.rs:41-44).
// u, v are only produced for the triangle; sphere / cuboid uv come from reconstruct_hit().
template <uint32_t FEAT = FEAT_ALL, class Tab>
__device__ __forceinline__ bool primitive_closest(const DevScene& S, const Tab& tab, const ObjHead& ob, f3 pos, f3 dir, float t_max, float& t, float& u, float& v) {
    if ((FEAT & FEAT_SPHERE) && ob.geom_kind == GEOM_UNIT_SPHERE) {
        return unit_sphere_hit(pos, dir, t_max, t);
    } else if ((FEAT & FEAT_CUBOID) && (ob.geom_kind == GEOM_CUBOID || (FEAT & (FEAT_SPHERE | FEAT_OTHER)) == 0u)) {   // (the only kind left)
        const CuboidRec c = tab.cuboid(ob.geom_index);
        f3 p = (pos - c.center) * c.inv_extent;  // cuboid.rs:83-85
        f3 d = dir * c.inv_extent;
        return unit_cuboid_hit(p, d, t_max, t);
    } else if (!(FEAT & FEAT_OTHER)) {
        return false;   // (not reached: the scene holds no other kind)
    } else if (ob.geom_kind == GEOM_UNIT_CUBOID) {
        return unit_cuboid_hit(pos, dir, t_max, t);
    } else {  // GEOM_TRIANGLE
        const DevTriangle tr = S.triangles[ob.geom_index];
        f3 a = mk(tr.a[0], tr.a[1], tr.a[2]);
        f3 ab = mk(tr.b[0], tr.b[1], tr.b[2]) - a, ac = mk(tr.c[0], tr.c[1], tr.c[2]) - a;
        float4 q0 = make_float4(a.x, a.y, a.z, ab.x), q1 = make_float4(ab.y, ab.z, ac.x, ac.y), q2 = make_float4(ac.z, 0.f, 0.f, 0.f);
        return tri_test(q0, q1, q2, pos, dir, t_max, t, u, v);
    }
}
// Occlusion test of a non-mesh primitive: UnitSphere::is_hit (sphere.rs:63-69), Cuboid::is_hit
// (cuboid.rs:95-100, world-space slab with Ray::inv_dir), UnitCuboid::is_hit (cuboid.rs:50-54),
// Triangle::is_hit (triangle.rs:59-62).
template <uint32_t FEAT = FEAT_ALL, class Tab>
__device__ __forceinline__ bool primitive_any(const DevScene& S, const Tab& tab, const ObjHead& ob, f3 pos, f3 dir, float t_max) {
    if ((FEAT & FEAT_SPHERE) && ob.geom_kind == GEOM_UNIT_SPHERE) {
        return unit_sphere_any(pos, dir, t_max);
    } else if ((FEAT & FEAT_CUBOID) && (ob.geom_kind == GEOM_CUBOID || (FEAT & (FEAT_SPHERE | FEAT_OTHER)) == 0u)) {
        const CuboidRec c = tab.cuboid(ob.geom_index);
        f3 ctr = c.center, ext = c.extent;
        f3 inv = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
        float t0, t1;
        box_roots(pos, inv, ctr - ext, ctr + ext, t0, t1);
        return t0 <= t1 && t0 <= t_max && t1 >= 0.0f;
    } else if (!(FEAT & FEAT_OTHER)) {
        return false;   // (not reached: the scene holds no other kind)
    } else if (ob.geom_kind == GEOM_UNIT_CUBOID) {
        f3 inv = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
        float t0, t1;
        box_roots(pos, inv, fill(-1.0f), fill(1.0f), t0, t1);
        return t0 <= t1 && t0 <= t_max && t1 >= 0.0f;
    } else {
        const DevTriangle tr = S.triangles[ob.geom_index];
        f3 a = mk(tr.a[0], tr.a[1], tr.a[2]);
        f3 ab = mk(tr.b[0], tr.b[1], tr.b[2]) - a, ac = mk(tr.c[0], tr.c[1], tr.c[2]) - a;
        float4 q0 = make_float4(a.x, a.y, a.z, ab.x), q1 = make_float4(ab.y, ab.z, ac.x, ac.y), q2 = make_float4(ac.z, 0.f, 0.f, 0.f);
        float t, u, v;
        return tri_test(q0, q1, q2, pos, dir, t_max, t, u, v);
    }
}

// Object::hit (src/geom/aggregate.rs:104-111) for the closest-hit query: returns true and shrinks
// t_max (caller's space) when the object is hit within t_max.
template <class Mem>
__device__ __forceinline__ bool object_closest(const DevScene& S, const Mem* lds_mem, uint32_t oi, f3 pos, f3 dir, float& t_max, Best& best) {
    const GlobalTab tab{&S};
    const ObjHead ob = tab.object(oi);
    float tl_max = t_max;
    if (!ops_in(tab, ob.op_first, ob.op_count, pos, dir, tl_max)) return false;
    float t = 0.0f, u = 0.0f, v = 0.0f;
    uint32_t prim = 0;
    bool hit = false;
    if (ob.geom_kind == GEOM_MESH) {
        const DevMesh m = S.meshes[ob.geom_index];
        if (lds_mem != nullptr && ob.geom_index == S.lds_mesh) {
            hit = traverse_mesh<false>(*lds_mem, m.n_nodes, pos, dir, tl_max, prim, u, v);
        } else {
            GlobalMem gm{S.nodes + 2 * (size_t)m.node_base, S.tris + 3 * (size_t)m.tri_base};
            hit = traverse_mesh<false>(gm, m.n_nodes, pos, dir, tl_max, prim, u, v);
        }
        t = tl_max;
    } else {
        hit = primitive_closest(S, tab, ob, pos, dir, tl_max, t, u, v);
    }
    if (hit) {
        best.obj = oi; best.prim = prim; best.t_local = t; best.u = u; best.v = v;
        t_max = ops_t_out(tab, ob.op_first, ob.op_count, t);
    }
    return hit;
}

// Object::is_hit (aggregate.rs:113-118) for the occlusion query.
template <class Mem>
__device__ __forceinline__ bool object_any(const DevScene& S, const Mem* lds_mem, uint32_t oi, f3 pos, f3 dir, float t_max) {
    const GlobalTab tab{&S};
    const ObjHead ob = tab.object(oi);
    if (!ops_in(tab, ob.op_first, ob.op_count, pos, dir, t_max)) return false;
    if (ob.geom_kind == GEOM_MESH) {
        const DevMesh m = S.meshes[ob.geom_index];
        uint32_t prim; float u, v;
        if (lds_mem != nullptr && ob.geom_index == S.lds_mesh) {
            return traverse_mesh<true>(*lds_mem, m.n_nodes, pos, dir, t_max, prim, u, v);
        }
        GlobalMem gm{S.nodes + 2 * (size_t)m.node_base, S.tris + 3 * (size_t)m.tri_base};
        return traverse_mesh<true>(gm, m.n_nodes, pos, dir, t_max, prim, u, v);
    }
    return primitive_any(S, tab, ob, pos, dir, t_max);
}

// GeometryList::hit (src/geom/aggregate.rs:24-35) over the root items; an object tree item is
// AABBTree<Object>::hit (src/geom/bvh.rs:259-269) in world space.
template <class Mem>
__device__ __forceinline__ bool scene_closest(const DevScene& S, const Mem* lds_mem, f3 pos, f3 dir, float t_max, Best& best, float& t_hit) {
    bool found = false;
    for (uint32_t r = 0; r < S.n_root; r++) {
        const DevRootItem it = S.root[r];
        if (it.kind == ROOT_OBJECT) {
            found |= object_closest(S, lds_mem, it.index, pos, dir, t_max, best);
        } else {
            const DevTree tr = S.trees[it.index];
            const float4* nodes = S.nodes + 2 * (size_t)tr.node_base;
            f3 inv = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
            bool ngx = signbit32(inv.x), ngy = signbit32(inv.y), ngz = signbit32(inv.z);
            uint32_t i = 0;
            while (i < tr.n_nodes) {
                float4 n0 = nodes[2 * i], n1 = nodes[2 * i + 1];
                uint32_t A = __float_as_uint(n1.z), B = __float_as_uint(n1.w);
                bool leaf = (A & LEAF_BIT) != 0u;
                if (node_passes(n0, n1, pos, inv, ngx, ngy, ngz, t_max)) {
                    if (leaf) {
                        for (uint32_t k = A & ~LEAF_BIT; k < B; k++)
                            found |= object_closest(S, lds_mem, tr.first_object + k, pos, dir, t_max, best);
                    }
                    i += 1;
                } else {
                    i = leaf ? i + 1 : A;
                }
            }
        }
    }
    t_hit = t_max;
    return found;
}

// GeometryList::is_hit (aggregate.rs:37-45) / AABBTree<Object>::is_hit (bvh.rs:271-281).
template <class Mem>
__device__ __forceinline__ bool scene_any(const DevScene& S, const Mem* lds_mem, f3 pos, f3 dir, float t_max) {
    for (uint32_t r = 0; r < S.n_root; r++) {
        const DevRootItem it = S.root[r];
        if (it.kind == ROOT_OBJECT) {
            if (object_any(S, lds_mem, it.index, pos, dir, t_max)) return true;
        } else {
            const DevTree tr = S.trees[it.index];
            const float4* nodes = S.nodes + 2 * (size_t)tr.node_base;
            f3 inv = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
            bool ngx = signbit32(inv.x), ngy = signbit32(inv.y), ngz = signbit32(inv.z);
            uint32_t i = 0;
            while (i < tr.n_nodes) {
                float4 n0 = nodes[2 * i], n1 = nodes[2 * i + 1];
                uint32_t A = __float_as_uint(n1.z), B = __float_as_uint(n1.w);
                bool leaf = (A & LEAF_BIT) != 0u;
                if (node_passes(n0, n1, pos, inv, ngx, ngy, ngz, t_max)) {
                    if (leaf) {
                        for (uint32_t k = A & ~LEAF_BIT; k < B; k++)
                            if (object_any(S, lds_mem, tr.first_object + k, pos, dir, t_max)) return true;
                    }
                    i += 1;
                } else {
                    i = leaf ? i + 1 : A;
                }
            }
        }
    }
    return false;
}

// HitResult{pos, norm, uv} of the winning primitive in the caller's space
// (mesh.rs:131-143, sphere.rs:47-59, cuboid.rs:30-47,88, triangle.rs:45-51, then the wrappers
// on the way out).  Needs the original ray; recomputes the object-space ray with the same
// operations as the traversal did.
__device__ __forceinline__ void reconstruct_hit(const DevScene& S, const Best& b, f3 pos, f3 dir, f3& hpos, f3& hnorm, float& hu, float& hv) {
    const GlobalTab tab{&S};
    const ObjHead ob = tab.object(b.obj);
    float unbounded = __builtin_huge_valf();   // the ray got through every BoundingBox wrapper when it was traced
    // A Translate -> Scale -> Rotate chain is applied from the object record's inline parameters, as the traversal does
    // (the same operations in the same order): the record is one read, the op list a chain of dependent ones.
    const bool inline_xf = !(ob.xf_kind & XF_GENERIC);
    // (on the way out a Translate by +0, which takes no part in entering, still turns -0 into +0: the upload says whether the
    // chain's way out is the inline parameters' too)
    const bool inline_out = (ob.xf_kind & XF_OUT_INLINE) != 0u;
    ObjXf xf;
    if (inline_xf) {
        xf = tab.object_xf(b.obj);
        xf_in_pos(ob.xf_kind, xf, ob.inv_scale, pos, unbounded);
        dir = xf_in_dir(ob.xf_kind, xf, dir);
    } else {
        (void)ops_in(tab, ob.op_first, ob.op_count, pos, dir, unbounded);
    }
    f3 p, n;
    hu = b.u; hv = b.v;
    if (ob.geom_kind == GEOM_MESH) {
        float4 q2 = S.tris[3 * ((size_t)ob.tri_base + b.prim) + 2];
        const float4* nr = S.normals + ob.normal_base;
        float4 na = nr[__float_as_uint(q2.y)], nb = nr[__float_as_uint(q2.z)], nc = nr[__float_as_uint(q2.w)];
        float w = 1.0f - b.u - b.v;
        n = (w * mk(na.x, na.y, na.z) + b.u * mk(nb.x, nb.y, nb.z)) + b.v * mk(nc.x, nc.y, nc.z);
        p = pos + fill(b.t_local) * dir;
    } else if (ob.geom_kind == GEOM_UNIT_SPHERE) {
        p = pos + fill(b.t_local) * dir;
        n = p;
        const float PI_F = 3.14159274101257324219f;
        hu = atan2f(n.x, n.y) / PI_F * 0.5f + 0.5f;
        hv = acosf(n.z) / PI_F;
    } else if (ob.geom_kind == GEOM_CUBOID || ob.geom_kind == GEOM_UNIT_CUBOID) {
        f3 lp = pos, ld = dir;
        CuboidRec c;
        c.center = mk(0, 0, 0); c.extent = mk(1, 1, 1); c.inv_extent = mk(1, 1, 1);
        if (ob.geom_kind == GEOM_CUBOID) {
            c = tab.cuboid(ob.geom_index);
            lp = (pos - c.center) * c.inv_extent;
            ld = dir * c.inv_extent;
        }
        f3 q = lp + fill(b.t_local) * ld;
        f3 qa = mk(fabsf(q.x), fabsf(q.y), fabsf(q.z));
        if (qa.x > qa.y && qa.x > qa.z) { n = mk(q.x, 0.0f, 0.0f); hu = q.y; hv = q.z; }
        else if (qa.y > qa.z) { n = mk(0.0f, q.y, 0.0f); hu = q.x; hv = q.z; }
        else { n = mk(0.0f, 0.0f, q.z); hu = q.x; hv = q.y; }
        p = q;
        if (ob.geom_kind == GEOM_CUBOID) p = q * c.extent + c.center;
    } else {
        const DevTriangle tr = S.triangles[ob.geom_index];
        f3 a = mk(tr.a[0], tr.a[1], tr.a[2]);
        n = cross(mk(tr.b[0], tr.b[1], tr.b[2]) - a, mk(tr.c[0], tr.c[1], tr.c[2]) - a);
        p = pos + fill(b.t_local) * dir;
    }
    if (inline_out) {   // ops_hit_out for that chain, innermost wrapper first
        if (ob.xf_kind & XF_R) { p = mat_apply(xf.rot, p); n = mat_apply(xf.rot, n); }
        if (ob.xf_kind & XF_S) p = p * ob.scale;
        if (ob.xf_kind & XF_T_OUT) p = p + xf.off;
    } else {
        ops_hit_out(tab, ob.op_first, ob.op_count, p, n);
    }
    hpos = p; hnorm = n;
}

}  // namespace rtamd
