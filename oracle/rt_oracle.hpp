// TEST INFRASTRUCTURE -- CPU oracle for the render hot path.
//
// This is a scalar C++17 restatement of the algorithms of the reference CPU ray
// tracer (stijnh/rust-raytracer).  It exists only so that tests, the smoke check
// and the benchmark's cpu_baseline leg can check / time the HIP path against it.
// Nothing in the shipped library may include, link or call this code.
//
// Every function cites the reference file:line it follows.  Build with
//   g++ -std=c++17 -O2 -ffp-contract=off (no fast-math)
// so that every float operation is a single IEEE-754 binary32 operation in the
// order the reference source writes it (rustc never contracts to FMA).
//
// PARITY PINNING
//   * hit / t / u / v path (+ - * / sqrt and comparisons only): pinned by the
//     analytic known-answer tests in tests/test_oracle_kat.py, each derived from
//     the cited reference lines.  The reference ships no tests, golden vectors
//     or fixtures of its own, and cannot be built here (no Rust toolchain).
//   * random stream (rand 0.6 SmallRng, source absent from the reference tree):
//     PARITY UNPINNED -- restated from the published algorithm (PCG
//     Mcg128Xsl64, rand_core seed_from_u64), no reference vector exists.
//   * libm transcendentals (atan2/acos/pow/sin/cos/tan) and JPEG decoding of the
//     skybox: PARITY UNPINNED (implementation-defined last-bit behaviour).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- common.rs:31-53
// max!/min! are compare-selects, NOT fmax/fmin: a NaN in the left operand is
// dropped, a NaN in the right operand is returned.
static inline float max2(float a, float b) { return a > b ? a : b; }
static inline float min2(float a, float b) { return a < b ? a : b; }
static inline float max3(float a, float b, float c) { return max2(a, max2(b, c)); }
static inline float min3(float a, float b, float c) { return min2(a, min2(b, c)); }

// ---------------------------------------------------------------- math/vec3d.rs
struct V3 {
    float d[3];
    V3() : d{0.f, 0.f, 0.f} {}
    V3(float x, float y, float z) : d{x, y, z} {}
    static V3 fill(float x) { return V3(x, x, x); }      // vec3d.rs:28
    float operator[](int i) const { return d[i]; }
    float& operator[](int i) { return d[i]; }
};
// vec3d.rs:181-219 component-wise operators
static inline V3 operator+(V3 a, V3 b) { return V3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
static inline V3 operator-(V3 a, V3 b) { return V3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
static inline V3 operator*(V3 a, V3 b) { return V3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
static inline V3 operator/(V3 a, V3 b) { return V3(a[0] / b[0], a[1] / b[1], a[2] / b[2]); }
static inline V3 operator-(V3 a) { return V3(-a[0], -a[1], -a[2]); }
// vec3d.rs:245-281 scalar forms go through fill()
static inline V3 operator*(V3 a, float s) { return a * V3::fill(s); }
static inline V3 operator*(float s, V3 a) { return V3::fill(s) * a; }
static inline V3 operator/(V3 a, float s) { return a / V3::fill(s); }
static inline V3 operator/(float s, V3 a) { return V3::fill(s) / a; }
// vec3d.rs:115-126
static inline float dot(V3 a, V3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline V3 cross(V3 a, V3 b) {
    return V3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
static inline float norm_squared(V3 a) { return dot(a, a); }           // vec3d.rs:95
static inline float norm(V3 a) { return std::sqrt(norm_squared(a)); }   // vec3d.rs:90
static inline V3 normalize(V3 a) {                                      // vec3d.rs:100-112
    float len = norm(a);
    if (len > 0.0f) return a / V3::fill(len);
    float n = std::numeric_limits<float>::quiet_NaN();
    return V3::fill(n);
}
static inline bool is_zero(V3 a) { return a[0] == 0.0f && a[1] == 0.0f && a[2] == 0.0f; }  // :80
void ortho_axes(V3 a, V3& b, V3& c);                                    // vec3d.rs:128-141

// ---------------------------------------------------------------- math/ray.rs:10-21
struct Ray {
    V3 pos, dir, inv_dir;
    Ray() {}
    Ray(V3 p, V3 d) : pos(p), dir(d), inv_dir(1.0f / d) {}
    V3 at(float t) const { return pos + V3::fill(t) * dir; }
};

// ---------------------------------------------------------------- math/aabb.rs
struct AABB {
    V3 min, max;
    AABB()                                                              // aabb.rs:12-17
        : min(V3::fill(std::numeric_limits<float>::infinity())),
          max(V3::fill(-std::numeric_limits<float>::infinity())) {}
    AABB(V3 mn, V3 mx) : min(mn), max(mx) {}
    static AABB from_point(V3 p) { return AABB(p, p); }
    AABB union_(AABB o) const {                                         // aabb.rs:31-36
        return AABB(V3(min2(min[0], o.min[0]), min2(min[1], o.min[1]), min2(min[2], o.min[2])),
                    V3(max2(max[0], o.max[0]), max2(max[1], o.max[1]), max2(max[2], o.max[2])));
    }
    AABB union_point(V3 p) const { return union_(from_point(p)); }
    static AABB from_points(V3 p, V3 q) { return from_point(p).union_point(q); }
    float surface_area() const {                                        // aabb.rs:42-45
        V3 dl = max - min;
        return 2.0f * (dl[0] * dl[1] + dl[1] * dl[2] + dl[2] * dl[0]);
    }
    // aabb.rs:61-87
    bool fast_intersect_ray(V3 ray_pos, V3 inv_dir, const bool neg[3], float& t0, float& t1) const;
    // aabb.rs:47-59
    bool intersect_ray(V3 ray_pos, V3 inv_dir, float& t0, float& t1) const;
};

// ---------------------------------------------------------------- math/mat3d.rs
struct Mat3 {
    float m[3][3];
    static Mat3 identity();
    static Mat3 new_rotation(V3 axis, float angle);      // mat3d.rs:47-64 (keeps the `1.0 * c` slip)
    static Mat3 new_reflection(V3 axis);                 // mat3d.rs:66-80 (keeps the `0.0 * 2.0 * b * c` slip)
    static Mat3 multiply(const Mat3& a, const Mat3& b);  // mat3d.rs:113-133
    V3 apply(V3 a) const;                                 // mat3d.rs:139-153
    V3 transpose_apply(V3 a) const;                       // mat3d.rs:155-169
};

// ---------------------------------------------------------------- rand 0.6 SmallRng (unpinned)
struct SmallRng {
    unsigned __int128 state;
    static SmallRng seed_from_u64(uint64_t s);
    uint64_t next_u64();
    uint32_t next_u32() { return (uint32_t)next_u64(); }
    float gen_f32() {  // Standard f32: 24 random bits * 2^-24
        uint32_t v = next_u32() >> 8;
        return (1.0f / 16777216.0f) * (float)v;
    }
};

// ---------------------------------------------------------------- texture.rs / material.rs / light.rs
typedef V3 Color;

struct Image {  // texture.rs:49-82
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;
    Color color_at(float u, float v) const;
};

enum TextureKind { TEX_COLOR = 0, TEX_UV = 1, TEX_CHECKERBOARD = 2, TEX_IMAGE = 3 };
struct Texture {  // texture.rs:10-47, :70-82
    int kind = TEX_COLOR;
    Color color = Color(1.f, 1.f, 1.f);   // impl Texture for Color, texture.rs:14-18
    int repeats = 0;                       // Checkerboard(i32), texture.rs:28-34
    std::shared_ptr<const Image> image;    // texture::Image
    Color color_at(float u, float v) const;
};

enum MaterialKind { MAT_NULL = 0, MAT_METAL = 1, MAT_GLOSSY = 2, MAT_TRANSPARENT = 3, MAT_LAMBERT = 4 };
struct Material {  // material.rs:7-141
    int kind = MAT_NULL;
    float p0 = 0.f;  // Glossy exponent | Transparent ior
    float p1 = 0.f;  // Glossy mirror fraction
    Texture tex;     // Lambartian<T>.0 / Glossy<T>.2
    Color sample_at(float u, float v) const;
    bool scatter(V3 n, V3 i, SmallRng& rng, V3& out, Color& w) const;
};
V3 reflection(V3 normal, V3 i);                      // material.rs:83-85
bool refraction(V3 normal, V3 i, float ior, V3& o);  // material.rs:87-101
float fresnel(V3 normal, V3 i, float ior);           // material.rs:103-121

enum LightKind { LIGHT_AMBIENT = 0, LIGHT_POINT = 1, LIGHT_DIRECTION = 2, LIGHT_AO = 3 };
struct Light {  // light.rs:5-158
    int kind = LIGHT_AMBIENT;
    V3 vec;             // Direction: normalised dir | Point: position
    float param = 0.f;  // Direction: spread (<=0: none) | AO: dist | Point: radius
    Color emission;
    bool is_delta_distribution() const;
    void sample_incidence(V3 pos, V3 normal, SmallRng& rng, V3& dir, float& t_max, Color& ill) const;
};

// ---------------------------------------------------------------- geom/mod.rs:20-34
struct HitResult {
    V3 pos, norm;
    float t = 0.f;
    float uv[2] = {0.f, 0.f};
    const Material* material = nullptr;
    // additions for parity bookkeeping (not in the reference):
    int32_t obj = -1;         // index in the root GeometryList / object tree leaf order
    uint32_t prim = 0;        // leaf-order index inside the innermost AABBTree
    float t_local = 0.f;      // t in the innermost object's space
};

struct Counters {
    // visit counts of the reference-order traversal; the *_any fields count the part of each total
    // that was spent inside occlusion (is_hit) queries
    uint64_t nodes_visited = 0, tris_tested = 0, spheres_tested = 0, cuboids_tested = 0;
    uint64_t closest_rays = 0, any_rays = 0, degenerate_tris = 0;
    uint64_t nodes_visited_any = 0, tris_tested_any = 0, spheres_tested_any = 0, cuboids_tested_any = 0;
    bool in_any = false;  // set while an is_hit query runs on this thread
    void add(const Counters& o) {
        nodes_visited += o.nodes_visited; tris_tested += o.tris_tested;
        spheres_tested += o.spheres_tested; cuboids_tested += o.cuboids_tested;
        closest_rays += o.closest_rays; any_rays += o.any_rays; degenerate_tris += o.degenerate_tris;
        nodes_visited_any += o.nodes_visited_any; tris_tested_any += o.tris_tested_any;
        spheres_tested_any += o.spheres_tested_any; cuboids_tested_any += o.cuboids_tested_any;
    }
    void node() { nodes_visited++; if (in_any) nodes_visited_any++; }
    void tri() { tris_tested++; if (in_any) tris_tested_any++; }
    void sphere() { spheres_tested++; if (in_any) spheres_tested_any++; }
    void cuboid() { cuboids_tested++; if (in_any) cuboids_tested_any++; }
};
extern thread_local Counters g_cnt;
// Diagnostic (not in the reference): when set, integrate_recur appends one record per closest-hit ray of the calling thread:
// {depth, object (-1: miss), primitive, material kind (-1: miss), t, origin xyz, direction xyz} = 11 floats.
extern thread_local std::vector<float>* g_path_log;

struct Geometry {
    virtual ~Geometry() {}
    virtual AABB bounding_box() const = 0;
    virtual bool hit(const Ray& ray, float t_max, HitResult& h) const = 0;
    virtual bool is_hit(const Ray& ray, float t_max) const {  // geom/mod.rs:31-33
        HitResult h;
        return hit(ray, t_max, h);
    }
};
typedef std::shared_ptr<const Geometry> GeomPtr;

// geom/triangle.rs:18-38
void moller_trumbore(V3 a, V3 b, V3 c, const Ray& ray, float out[3]);

struct BvhNode { AABB bb; uint32_t skip, first; };  // bvh.rs:7

// bvh.rs:5-286, generic over the leaf type
template <class T>
struct AABBTree : Geometry {
    std::vector<T> objs;
    std::vector<BvhNode> nodes;
    AABBTree(std::vector<T> o, float hit_cost);
    AABB bounding_box() const override { return nodes[0].bb; }
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
    void stats(uint32_t out[5]) const;  // objs, nodes, leafs, min leaf, max leaf (bvh.rs:179-204)
};

struct MeshData { std::vector<V3> data; };  // [vertices || normals], mesh.rs:36-39
struct MeshTriangle {                       // mesh.rs:11-14,119-157
    uint32_t v[3];
    std::shared_ptr<const MeshData> data;
    uint32_t face;  // original face index (bookkeeping)
    AABB bounding_box() const;
    bool hit(const Ray& ray, float t_max, HitResult& h) const;
    bool is_hit(const Ray& ray, float t_max) const { HitResult h; return hit(ray, t_max, h); }
};
struct Mesh : Geometry {  // mesh.rs:16-81
    std::unique_ptr<AABBTree<MeshTriangle>> tree;
    std::shared_ptr<const MeshData> data;
    static std::shared_ptr<Mesh> from_vertices(const std::vector<V3>& vertices,
                                               const std::vector<std::array<uint32_t, 3>>& faces);
    static std::shared_ptr<Mesh> from_triangles(const std::vector<std::array<V3, 3>>& tris);  // mesh.rs:83-116
    AABB bounding_box() const override { return tree->bounding_box(); }
    bool hit(const Ray& r, float t, HitResult& h) const override { return tree->hit(r, t, h); }
    bool is_hit(const Ray& r, float t) const override { return tree->is_hit(r, t); }
};

struct Triangle : Geometry {  // triangle.rs:40-69
    V3 a, b, c;
    Triangle(V3 a_, V3 b_, V3 c_) : a(a_), b(b_), c(c_) {}
    AABB bounding_box() const override { return AABB::from_point(a).union_point(b).union_point(c); }
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};

struct UnitSphere : Geometry {  // sphere.rs:22-74
    AABB bounding_box() const override { return AABB::from_points(-V3::fill(1.f), V3::fill(1.f)); }
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};
struct UnitCuboid : Geometry {  // cuboid.rs:18-59
    AABB bounding_box() const override { return AABB(-V3::fill(1.f), V3::fill(1.f)); }
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};
struct Cuboid : Geometry {  // cuboid.rs:61-105
    V3 center, extent, inv_extent;
    Cuboid(V3 a, V3 b);
    AABB bounding_box() const override { return AABB(center - extent, center + extent); }
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};

// transform.rs:5-309
struct Translate : Geometry {
    GeomPtr obj; V3 offset;
    Translate(GeomPtr o, V3 off) : obj(o), offset(off) {}
    AABB bounding_box() const override;
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};
struct Scale : Geometry {
    GeomPtr obj; float scale, inv_scale;
    Scale(GeomPtr o, float s) : obj(o), scale(s), inv_scale(1.0f / s) {}
    AABB bounding_box() const override;
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};
struct Rotate : Geometry {
    GeomPtr obj; Mat3 mat;
    Rotate(GeomPtr o, Mat3 m) : obj(o), mat(m) {}
    AABB bounding_box() const override;
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};
// Transform<T> = Translate<Scale<Rotate<T>>> with the builder methods of transform.rs:223-292
struct TransformBuilder {
    GeomPtr obj; V3 offset; float scale_ = 1.0f; Mat3 mat = Mat3::identity();
    explicit TransformBuilder(GeomPtr o) : obj(o) {}
    TransformBuilder& translate(V3 off) { offset = offset + off; return *this; }
    TransformBuilder& scale(float s) { scale_ = scale_ * s; offset = offset * s; return *this; }
    TransformBuilder& rotate(V3 axis, float angle);
    TransformBuilder& reflect(V3 axis);                  // transform.rs:274-279
    GeomPtr build() const;
};
GeomPtr make_sphere(V3 center, float radius);  // sphere.rs:14-19

// aggregate.rs:98-148 (ObjectImpl stamps the material; index is bookkeeping)
struct Object : Geometry {
    GeomPtr geometry; Material material; int32_t index = -1;
    Object() {}
    Object(GeomPtr g, Material m) : geometry(g), material(m) {}
    AABB bounding_box() const override { return geometry->bounding_box(); }
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override { return geometry->is_hit(ray, t_max); }
};
// aggregate.rs:58-96
struct BoundingBox : Geometry {
    GeomPtr obj; AABB box;
    explicit BoundingBox(GeomPtr o);
    AABB bounding_box() const override { return box; }
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};
// aggregate.rs:23-56
struct GeometryList : Geometry {
    std::vector<GeomPtr> items;
    AABB bounding_box() const override;
    bool hit(const Ray& ray, float t_max, HitResult& h) const override;
    bool is_hit(const Ray& ray, float t_max) const override;
};

// ---------------------------------------------------------------- scene.rs
struct Camera {  // scene.rs:25-90
    size_t width, height;
    V3 pos, dir, horizontal, vertical;
    Camera(size_t w, size_t h);
    Camera& position(V3 p) { pos = p; return *this; }
    Camera& look_towards(V3 d, V3 up);
    Camera& look_at(V3 lookat, V3 up) { return look_towards(lookat - pos, up); }
    Camera& perspective(float fov);
    Ray generate_ray(float x, float y) const;
};
struct Scene {  // scene.rs:8-21
    GeomPtr root;
    std::shared_ptr<const Image> skybox;
    std::vector<Light> lights;
    Camera camera = Camera(1, 1);
    Color calculate_background(const Ray& ray) const;
};

// ---------------------------------------------------------------- integrator.rs
struct WhittedIntegrator {  // integrator.rs:12-31
    int max_depth = 10, shadow_rays = 5, scatter_rays = 1, antialiasing = 1;
    float gamma = 2.2f;
    Color calculate_pixel(const Scene& scene, size_t cx, size_t cy) const;
    Color integrate_recur(const Scene& scene, const Ray& ray, int depth, SmallRng& rng) const;
    Color illumination(const Scene& scene, const Light& light, V3 pos, V3 normal, SmallRng& rng) const;
};

// loader/ply.rs:254-277 with the element-name slip fixed (token 1 is the name)
bool load_ply(const std::string& path, std::vector<V3>& vertices,
              std::vector<std::array<uint32_t, 3>>& faces, std::string& err);

}  // namespace orc
