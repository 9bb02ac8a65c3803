// Host side of the MI355X render path: the scene-building API of the reference renderer
// (geometry wrappers, materials, lights, camera, integrator parameters, render entry points),
// restated in C++ because no Rust toolchain exists in this build environment.  Nothing here walks
// a scene on the CPU: Geometry::hit / is_hit and parallel_render are batched calls into the C ABI
// of include/rtamd.h, and they fail loudly when the GPU library reports an error.
//
// Names follow the reference: src/geom/*.rs, src/scene.rs, src/integrator.rs, src/render.rs,
// src/material.rs, src/light.rs, src/loader/ply.rs.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "rtamd.h"

namespace rth {

// ---- math (src/math/vec3d.rs, aabb.rs, mat3d.rs) -------------------------------------------------
struct Vec3D {
    float x = 0.f, y = 0.f, z = 0.f;
    Vec3D() {}
    Vec3D(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3D fill(float v) { return Vec3D(v, v, v); }
    static Vec3D x_axis() { return Vec3D(1.f, 0.f, 0.f); }
    static Vec3D y_axis() { return Vec3D(0.f, 1.f, 0.f); }
    static Vec3D z_axis() { return Vec3D(0.f, 0.f, 1.f); }
    static Vec3D one() { return fill(1.f); }
    static Vec3D zero() { return fill(0.f); }
    float get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float norm() const;
    Vec3D normalize() const;
    static float dot(Vec3D a, Vec3D b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
    static Vec3D cross(Vec3D a, Vec3D b) {
        return Vec3D(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    }
};
inline Vec3D operator+(Vec3D a, Vec3D b) { return Vec3D(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3D operator-(Vec3D a, Vec3D b) { return Vec3D(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3D operator*(Vec3D a, Vec3D b) { return Vec3D(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3D operator/(Vec3D a, Vec3D b) { return Vec3D(a.x / b.x, a.y / b.y, a.z / b.z); }
inline Vec3D operator-(Vec3D a) { return Vec3D(-a.x, -a.y, -a.z); }
inline Vec3D operator*(Vec3D a, float s) { return Vec3D(a.x * s, a.y * s, a.z * s); }
inline Vec3D operator*(float s, Vec3D a) { return Vec3D(s * a.x, s * a.y, s * a.z); }
inline Vec3D operator/(Vec3D a, float s) { return Vec3D(a.x / s, a.y / s, a.z / s); }
inline Vec3D vec3d(float x, float y, float z) { return Vec3D(x, y, z); }

struct AABB {
    Vec3D min, max;
    AABB();  // empty box (+inf, -inf), aabb.rs:12-17
    AABB(Vec3D mn, Vec3D mx) : min(mn), max(mx) {}
    static AABB from_point(Vec3D p) { return AABB(p, p); }
    AABB union_with(const AABB& o) const;      // aabb.rs:31-36 (compare-select min!/max!)
    AABB union_point(Vec3D p) const { return union_with(from_point(p)); }
    static AABB from_points(Vec3D p, Vec3D q) { return from_point(p).union_point(q); }
    float surface_area() const;                // aabb.rs:42-45
};

struct Mat3D {
    float m[9];  // row-major
    static Mat3D identity();
    static Mat3D new_rotation(Vec3D axis, float angle);   // mat3d.rs:47-64, including its `1.0 * c` term
    static Mat3D new_reflection(Vec3D axis);              // mat3d.rs:66-80, including its `0.0 * 2.0 * b * c` term
    static Mat3D multiply(const Mat3D& a, const Mat3D& b);
    Vec3D apply(Vec3D a) const;
    Vec3D transpose_apply(Vec3D a) const;
};

inline float to_radians(float deg) { return deg * (3.14159274101257324219f / 180.0f); }

// ---- materials / lights / textures (src/material.rs, light.rs, texture.rs) -----------------------
typedef Vec3D Color;
static const Color COLOR_WHITE(1.f, 1.f, 1.f);
static const Color COLOR_BLACK(0.f, 0.f, 0.f);

struct Image;
struct Texture {  // trait Texture and its impls, texture.rs:10-47, :70-82
    uint32_t kind = RTAMD_TEX_COLOR;
    Color color = Color(1.f, 1.f, 1.f);
    int32_t param = 0;
    std::shared_ptr<const Image> image;
    Texture() {}
    Texture(Color c) : color(c) {}                               // impl Texture for Color
    Texture(std::shared_ptr<const Image> img) : kind(RTAMD_TEX_IMAGE), image(img) {}   // impl Texture for Image
};
inline Texture UVTexture() { Texture t; t.kind = RTAMD_TEX_UV; return t; }
inline Texture Checkerboard(int repeats) { Texture t; t.kind = RTAMD_TEX_CHECKERBOARD; t.param = repeats; return t; }

struct Material { rtamd_material m; std::shared_ptr<const Image> image; };
Material NullMaterial();
Material Metal();
Material Glossy(float exponent, float mirror, Texture t);
Material Transparent(float ior);
inline Material Glass() { return Transparent(1.5f); }   // material.rs:72-78
Material Lambartian(Texture t);                          // [sic], material.rs:135

struct Light { rtamd_light l; };
Light AmbientLight(Color color, float intensity);
Light DirectionLight(Vec3D dir, float spread, Color color, float intensity);
Light AmbientOcclusion(float dist, Color color, float intensity);
Light PointLight(Vec3D pos, float radius, Color color, float intensity);

struct Image {  // texture::Image, texture.rs:49-68 (decoded RGB8; PPM P6 reader for file input)
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
    static std::shared_ptr<Image> from_rgb(const uint8_t* rgb, uint32_t w, uint32_t h);
    static std::shared_ptr<Image> open_ppm(const std::string& path);
};

// ---- flattening ----------------------------------------------------------------------------------
struct Mesh;
struct FlatScene {  // owns every array rtamd_scene_desc points to
    std::vector<std::shared_ptr<const Mesh>> mesh_refs;
    std::unordered_map<const Mesh*, uint32_t> mesh_lookup;
    std::vector<rtamd_mesh> meshes;
    std::vector<rtamd_cuboid> cuboids;
    std::vector<rtamd_triangle> triangles;
    std::vector<rtamd_xform_op> ops;
    std::vector<rtamd_object> objects;
    std::vector<rtamd_object_tree> trees;
    std::vector<rtamd_root_item> root;
    std::vector<rtamd_material> materials;
    std::vector<rtamd_light> lights;
    std::vector<std::shared_ptr<const Image>> image_refs;
    std::vector<rtamd_image> images;
    uint32_t mesh_index(const std::shared_ptr<const Mesh>& m);
    uint32_t image_index(const std::shared_ptr<const Image>& img);
};

// ---- geometry (src/geom/*.rs) ---------------------------------------------------------------------
struct Geometry {
    virtual ~Geometry() {}
    virtual AABB bounding_box() const = 0;
    // Appends the wrappers of this node to `ops` and, at a primitive, emits the object record.
    virtual void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const = 0;
};
typedef std::shared_ptr<const Geometry> GeomPtr;

struct BvhTree {  // result of AABBTree::new: nodes incl. sentinel + the permutation of the inputs
    std::vector<rtamd_bvh_node> nodes;
    std::vector<uint32_t> order;  // order[k] = input index of the k-th object in tree order
};
// Binned-SAH pre-order build with skip links: AABBTree::with_partition + partition_sah,
// src/geom/bvh.rs:10-171 (including its split-cost expression as written).
BvhTree build_aabb_tree(const std::vector<AABB>& boxes, float hit_cost);
// AABBTree::print_stats (bvh.rs:179-204): "BVH statistics: N objs, M nodes, L leafs, min/max/avg leaf: a/b/c".  The reference prints
// it after every AABBTree::new (bvh.rs:173-177); here the builders print it when bvh_print_stats_enabled() is set (rtamd_main sets it).
void print_stats(const BvhTree& t);
bool& bvh_print_stats_enabled();
struct Device;
// The same tree, built on the GPU (rtamd_build_bvh): identical nodes and order; worth it from some 10^5 boxes on.
BvhTree build_aabb_tree(Device& dev, const std::vector<AABB>& boxes, float hit_cost);
void bvh_statistics(const BvhTree& t, uint32_t out[5]);  // bvh.rs:179-204

struct Mesh : Geometry, std::enable_shared_from_this<Mesh> {  // mesh.rs:16-81
    std::vector<float> vertices, normals;  // 3 floats per vertex
    std::vector<uint32_t> tri_indices;     // tree order, 3 per triangle
    std::vector<uint32_t> face_of_prim;    // original face index of each tree-order triangle
    BvhTree tree;
    // builder: build the tree on that GPU instead of on the host (same tree)
    static std::shared_ptr<Mesh> from_vertices(const std::vector<Vec3D>& vertices, const std::vector<std::array<uint32_t, 3>>& faces, Device* builder = nullptr);
    // mesh.rs:83-116: vertices shared by bit pattern, numbered in order of first appearance
    static std::shared_ptr<Mesh> from_triangles(const std::vector<std::array<Vec3D, 3>>& tris);
    AABB bounding_box() const override;
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};
struct UnitSphere : Geometry {
    AABB bounding_box() const override { return AABB::from_points(-Vec3D::one(), Vec3D::one()); }
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};
struct UnitCuboid : Geometry {
    AABB bounding_box() const override { return AABB(-Vec3D::one(), Vec3D::one()); }
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};
struct Cuboid : Geometry {  // cuboid.rs:61-79
    Vec3D center, extent, inv_extent;
    Cuboid(Vec3D a, Vec3D b);
    static GeomPtr make(Vec3D a, Vec3D b) { return std::make_shared<Cuboid>(a, b); }
    AABB bounding_box() const override { return AABB(center - extent, center + extent); }
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};
struct Triangle : Geometry {  // triangle.rs:5-16
    Vec3D a, b, c;
    Triangle(Vec3D a_, Vec3D b_, Vec3D c_) : a(a_), b(b_), c(c_) {}
    AABB bounding_box() const override { return AABB::from_point(a).union_point(b).union_point(c); }
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};

struct Translate : Geometry {  // transform.rs:29-79
    GeomPtr obj; Vec3D offset;
    Translate(GeomPtr o, Vec3D off) : obj(o), offset(off) {}
    static std::shared_ptr<Translate> make(GeomPtr o) { return std::make_shared<Translate>(o, Vec3D::zero()); }
    std::shared_ptr<Translate> translate(Vec3D off) const { return std::make_shared<Translate>(obj, offset + off); }
    std::shared_ptr<Translate> translate_x(float d) const { return translate(Vec3D(d, 0.f, 0.f)); }
    std::shared_ptr<Translate> translate_y(float d) const { return translate(Vec3D(0.f, d, 0.f)); }
    std::shared_ptr<Translate> translate_z(float d) const { return translate(Vec3D(0.f, 0.f, d)); }
    AABB bounding_box() const override;
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};
struct Scale : Geometry {  // transform.rs:81-128
    GeomPtr obj; float scale, inv_scale;
    Scale(GeomPtr o, float s);
    AABB bounding_box() const override;
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};
struct Rotate : Geometry {  // transform.rs:130-221
    GeomPtr obj; Mat3D mat;
    Rotate(GeomPtr o, Mat3D m) : obj(o), mat(m) {}
    AABB bounding_box() const override;
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};
struct BoundingBox : Geometry {  // aggregate.rs:58-96: slab prefilter around an object, box padded by 0.001 * (dx + dy * dz)
    GeomPtr obj; AABB box;
    explicit BoundingBox(GeomPtr o);
    static std::shared_ptr<BoundingBox> make(GeomPtr o) { return std::make_shared<BoundingBox>(o); }
    AABB bounding_box() const override { return box; }
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override;
};
struct Transform : Geometry {  // Translate<Scale<Rotate<T>>>, transform.rs:223-309
    GeomPtr obj; Vec3D offset; float scale_ = 1.0f; Mat3D mat = Mat3D::identity();
    explicit Transform(GeomPtr o) : obj(o) {}
    static std::shared_ptr<Transform> make(GeomPtr o) { return std::make_shared<Transform>(o); }
    std::shared_ptr<Transform> translate(Vec3D off) const;
    std::shared_ptr<Transform> scale(float s) const;
    std::shared_ptr<Transform> rotate(Vec3D axis, float angle) const;
    std::shared_ptr<Transform> rotate_x(float a) const { return rotate(Vec3D::x_axis(), a); }
    std::shared_ptr<Transform> rotate_y(float a) const { return rotate(Vec3D::y_axis(), a); }
    std::shared_ptr<Transform> rotate_z(float a) const { return rotate(Vec3D::z_axis(), a); }
    std::shared_ptr<Transform> reflect(Vec3D axis) const;   // transform.rs:274-279
    std::shared_ptr<Transform> reflect_x() const { return reflect(Vec3D::x_axis()); }
    std::shared_ptr<Transform> reflect_y() const { return reflect(Vec3D::y_axis()); }
    std::shared_ptr<Transform> reflect_z() const { return reflect(Vec3D::z_axis()); }
    GeomPtr chain() const;  // the explicit Translate(Scale(Rotate(obj)))
    AABB bounding_box() const override { return chain()->bounding_box(); }
    void flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const override { chain()->flatten(fs, ops, material); }
};
GeomPtr Sphere(Vec3D center, float radius);  // Translate<Scale<UnitSphere>>, sphere.rs:14-19

struct Object {  // aggregate.rs:121-138
    GeomPtr geometry; Material material;
    static Object with_material(GeomPtr g, Material m) { return Object{g, m}; }
    static Object make(GeomPtr g) { return Object{g, NullMaterial()}; }
    AABB bounding_box() const { return geometry->bounding_box(); }
};
// One entry of the root GeometryList: a plain Object or an AABBTree<Object>.
struct ObjectTree {  // AABBTree<Object>::new(objs, hit_cost), bvh.rs:173-177
    std::vector<Object> objs;  // tree order
    BvhTree tree;
    static std::shared_ptr<ObjectTree> make(std::vector<Object> objs, float hit_cost);
    AABB bounding_box() const;
};
struct GeometryList {  // aggregate.rs:7-56
    struct Item { bool is_tree; Object object; std::shared_ptr<const ObjectTree> tree; };
    std::vector<Item> items;
    static GeometryList from_vec(std::vector<Object> v);
    // Extension (the reference walks a GeometryList linearly, aggregate.rs:24-45): every run of at least `min_run` plain
    // objects of the list is replaced by one AABBTree<Object> (bvh.rs:173, hit_cost 1.0) over the run, in place.  The hits
    // are those of the list except where two objects of a run are hit at exactly the same t (the list reports the later
    // object, the tree the later one in tree order).  Returns the number of trees built.
    size_t accelerate(size_t min_run = 16);
    void push(Object o) { items.push_back(Item{false, o, nullptr}); }
    void push(std::shared_ptr<const ObjectTree> t) { items.push_back(Item{true, Object(), t}); }
    AABB bounding_box() const;
};

// ---- camera / scene / integrator (src/scene.rs, src/integrator.rs) --------------------------------
struct Camera {  // scene.rs:25-90
    size_t width, height;
    Vec3D pos, dir, horizontal, vertical;
    Camera(size_t w, size_t h);
    Camera& position(Vec3D p) { pos = p; return *this; }
    Camera& look_towards(Vec3D d, Vec3D up);
    Camera& look_at(Vec3D lookat, Vec3D up) { return look_towards(lookat - pos, up); }
    Camera& perspective(float fov);
    rtamd_camera flat() const;
};

struct WhittedIntegrator {  // integrator.rs:12-31 (gamma is never applied by the reference)
    int max_depth = 10, shadow_rays = 5, scatter_rays = 1, antialiasing = 1;
    float gamma = 2.2f;
    rtamd_integrator flat() const { return rtamd_integrator{max_depth, shadow_rays, scatter_rays, antialiasing}; }
};

struct GpuError : std::runtime_error { using std::runtime_error::runtime_error; };

// One GPU.  Owns the rtamd context.
struct Device {
    rtamd_context* ctx = nullptr;
    explicit Device(int device_id);
    ~Device();
    Device(const Device&) = delete;
    void check(int rc, const char* what) const;
};

struct Scene {  // scene.rs:8-13
    GeometryList root;
    // Extension, on by default: when the scene is flattened, every run of at least `accelerate_min` plain objects of the
    // root list is walked through an AABBTree<Object> built over the run (GeometryList::accelerate) instead of linearly;
    // the reference's own scenes (4 objects) are below the threshold.  linear_root = true keeps the reference's walk.
    bool linear_root = false;
    size_t accelerate_min = 64;
    std::shared_ptr<const Image> skybox;
    std::vector<Light> lights;
    Camera camera = Camera(1, 1);

    // Flatten + copy to a GPU (must be called again after the scene changes).  For several GPUs: upload() to the first,
    // upload_flat() (the arrays of the last flatten()) to the others.
    void flatten();
    void upload(Device& dev);
    void upload_flat(Device& dev);
    const FlatScene& flat() const { return *flat_; }

    // Batched Geometry::hit / is_hit on the root (geom/mod.rs:30-33).
    void hit(Device& dev, const rtamd_rays& rays, size_t n, const rtamd_hits& out) const;
    void is_hit(Device& dev, const rtamd_rays& rays, size_t n, uint8_t* out) const;

private:
    std::shared_ptr<FlatScene> flat_;
};

namespace render {
// render.rs:10-56: linear RGB, row-major width*height*3; prints the reference's timing line.
std::vector<float> parallel_render(Device& dev, const Scene& scene, const WhittedIntegrator& integ, rtamd_stats* stats = nullptr, bool quiet = false);
// The same frame over several GPU contexts (the scene uploaded to each): 64x64 image tiles dealt round-robin, one host
// thread per context, tiles written straight into the row-major frame (rtamd_render_multi).  Pixels are identical to the
// single-context frame: a pixel's random stream depends on its coordinates only (integrator.rs:36).
std::vector<float> parallel_render(const std::vector<Device*>& devs, const Scene& scene, const WhittedIntegrator& integ, rtamd_stats* stats = nullptr, bool quiet = false);
// render.rs:58-75: quantised RGB8 image.
std::vector<uint8_t> parallel_render_image(Device& dev, const Scene& scene, const WhittedIntegrator& integ);
void save_ppm(const std::string& path, const std::vector<uint8_t>& rgb, size_t w, size_t h);
// RGB8 PNG (main.rs:100 saves test_{i}.png through the image crate): stored deflate blocks, no compression.
void save_png(const std::string& path, const std::vector<uint8_t>& rgb, size_t w, size_t h);
}  // namespace render

namespace loader {
// loader/ply.rs:254-282 (ASCII PLY; the element name is token 1 of the `element` line).  Extension: `format
// binary_little_endian 1.0` / `binary_big_endian 1.0` files (which ply.rs:89-91 rejects) are read too, so that the
// 69 451-face bunny main.rs:24 names can be loaded in the form it is usually distributed in.
void load_ply(const std::string& path, std::vector<Vec3D>& vertices, std::vector<std::array<uint32_t, 3>>& faces);
std::shared_ptr<Mesh> load_ply_as_mesh(const std::string& path);
}  // namespace loader

// ---- scene scripts --------------------------------------------------------------------------------
// src/main.rs:22-83 with caller-supplied asset locations.
Scene scene_main(const std::string& ply_path, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h);
// Synthetic scenes of the larger benchmark configurations (generator: splitmix64, see rt_scenes.cpp).
Scene scene_spheres(int n, uint64_t seed, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h, float scale = 1.0f);
// flat 1: every instance is a mesh of its own with world-space vertices (no ray transform); flat 2: the instances of each
// of the three materials merged into one world-space mesh (its tree built on `builder` when given); see rt_scenes.cpp
Scene scene_grid(const std::string& ply_path, int nx, int ny, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h, int flat = 0, Device* builder = nullptr);
// n spheres and cuboids in a plain GeometryList (the top-level extension's test scene)
Scene scene_scatter(int n, uint64_t seed, bool accelerate, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h);
Scene scene_terrain(const std::string& ply_path, int n, uint64_t seed, bool with_bunny, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h);
Scene scene_textured(const std::string& ply_path, std::shared_ptr<const Image> skybox, std::shared_ptr<const Image> tex, size_t cam_w, size_t cam_h);

}  // namespace rth
