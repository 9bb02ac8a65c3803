// TEST INFRASTRUCTURE -- CPU oracle for the render hot path (see rt_oracle.hpp).
// Scalar restatement of the reference algorithms; never linked into the product.
#include "rt_oracle.hpp"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace orc {

thread_local Counters g_cnt;
thread_local std::vector<float>* g_path_log = nullptr;

static const float PI_F = 3.14159265358979323846f;  // std::f32::consts::PI

// ---------------------------------------------------------------- vec3d.rs:128-141
void ortho_axes(V3 a, V3& b, V3& c) {
    if (std::fabs(a[0]) > std::fabs(a[1])) {
        float inv_len = 1.0f / std::sqrt(a[0] * a[0] + a[2] * a[2]);
        b = V3(-a[2] * inv_len, 0.0f, a[0] * inv_len);
    } else {
        float inv_len = 1.0f / std::sqrt(a[1] * a[1] + a[2] * a[2]);
        b = V3(0.0f, a[2] * inv_len, -a[1] * inv_len);
    }
    c = cross(a, b);
}

// ---------------------------------------------------------------- aabb.rs:61-87
bool AABB::fast_intersect_ray(V3 ray_pos, V3 inv_dir, const bool neg[3], float& t0, float& t1) const {
    V3 a = (min - ray_pos) * inv_dir;
    V3 b = (max - ray_pos) * inv_dir;
    for (int i = 0; i < 3; i++) {
        if (neg[i]) std::swap(a[i], b[i]);
    }
    t0 = max3(a[0], a[1], a[2]);
    t1 = min3(b[0], b[1], b[2]);
    return t0 < t1;
}

bool AABB::intersect_ray(V3 ray_pos, V3 inv_dir, float& t0, float& t1) const {  // aabb.rs:47-59
    V3 a = (min - ray_pos) * inv_dir;
    V3 b = (max - ray_pos) * inv_dir;
    t0 = max3(min2(a[0], b[0]), min2(a[1], b[1]), min2(a[2], b[2]));
    t1 = min3(max2(a[0], b[0]), max2(a[1], b[1]), max2(a[2], b[2]));
    return t0 < t1;
}

// ---------------------------------------------------------------- mat3d.rs
Mat3 Mat3::identity() {
    Mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[i][j] = (i == j) ? 1.0f : 0.0f;
    return r;
}
Mat3 Mat3::new_rotation(V3 axis, float angle) {  // mat3d.rs:47-64
    axis = normalize(axis);
    float x = axis[0], y = axis[1], z = axis[2];
    float c = std::cos(angle);
    float s = std::sin(angle);
    Mat3 r;
    r.m[0][0] = c + x * x * (1.0f - c);
    r.m[0][1] = x * y * (1.0f - c) - z * s;
    r.m[0][2] = x * z * (1.0f - c) + y * s;
    r.m[1][0] = y * x * (1.0f - c) + z * s;
    r.m[1][1] = c + y * y * (1.0f - c);
    r.m[1][2] = y * z * (1.0f * c) - x * s;  // sic: `1.0 * c`, mat3d.rs:59
    r.m[2][0] = z * x * (1.0f - c) - y * s;
    r.m[2][1] = z * y * (1.0f - c) + x * s;
    r.m[2][2] = c + z * z * (1.0f - c);
    return r;
}
Mat3 Mat3::new_reflection(V3 axis) {  // mat3d.rs:66-80
    axis = normalize(axis);
    float a = axis[0], b = axis[1], c = axis[2];
    Mat3 r;
    r.m[0][0] = 1.0f - 2.0f * a * a;
    r.m[0][1] = 0.0f - 2.0f * a * b;
    r.m[0][2] = 0.0f - 2.0f * a * c;
    r.m[1][0] = 0.0f - 2.0f * a * b;
    r.m[1][1] = 1.0f - 2.0f * b * b;
    r.m[1][2] = 0.0f - 2.0f * b * c;
    r.m[2][0] = 0.0f - 2.0f * a * c;
    r.m[2][1] = 0.0f * 2.0f * b * c;  // sic, mat3d.rs:78
    r.m[2][2] = 1.0f - 2.0f * c * c;
    return r;
}
Mat3 Mat3::multiply(const Mat3& a, const Mat3& b) {  // mat3d.rs:113-133
    Mat3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float v = 0.0f;
            for (int k = 0; k < 3; k++) v += a.m[i][k] * b.m[k][j];
            r.m[i][j] = v;
        }
    return r;
}
V3 Mat3::apply(V3 a) const {  // mat3d.rs:139-153
    V3 b = V3::fill(0.0f);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) b[i] += m[i][j] * a[j];
    return b;
}
V3 Mat3::transpose_apply(V3 a) const {  // mat3d.rs:155-169
    V3 b = V3::fill(0.0f);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) b[i] += m[j][i] * a[j];
    return b;
}

// ---------------------------------------------------------------- rand 0.6 SmallRng
// rand_core SeedableRng::seed_from_u64 (PCG32 expansion of the u64 into the
// 16 seed bytes) + rand_pcg Mcg128Xsl64 (state |= 1; XSL-RR 128/64 output).
SmallRng SmallRng::seed_from_u64(uint64_t state) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    uint32_t words[4];
    for (int k = 0; k < 4; k++) {
        state = state * MUL + INC;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        words[k] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    uint64_t lo = (uint64_t)words[0] | ((uint64_t)words[1] << 32);
    uint64_t hi = (uint64_t)words[2] | ((uint64_t)words[3] << 32);
    SmallRng r;
    r.state = ((unsigned __int128)hi << 64) | (unsigned __int128)lo;
    r.state |= 1;
    return r;
}
uint64_t SmallRng::next_u64() {
    const unsigned __int128 MULT =
        ((unsigned __int128)0x2360ED051FC65DA4ULL << 64) | (unsigned __int128)0x4385DF649FCCF645ULL;
    state = state * MULT;
    uint32_t rot = (uint32_t)(state >> 122);
    uint64_t xsl = (uint64_t)(state >> 64) ^ (uint64_t)state;
    return (xsl >> rot) | (xsl << ((64 - rot) & 63));
}

// ---------------------------------------------------------------- texture.rs:70-82
static inline uint32_t sat_u32(float x) {  // Rust `as u32`: saturating, NaN -> 0
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
static inline float rust_min(float a, float b) {  // f32::min: returns the non-NaN operand
    if (a != a) return b;
    if (b != b) return a;
    return a < b ? a : b;
}
static inline float rust_max(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}
Color Image::color_at(float u, float v) const {
    uint32_t x = sat_u32(rust_max(rust_min(u * (float)w, (float)w - 1.0f), 0.0f));
    uint32_t y = sat_u32(rust_max(rust_min(v * (float)h, (float)h - 1.0f), 0.0f));
    const uint8_t* p = &rgb[((size_t)y * (size_t)w + x) * 3];
    return Color((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
}

// ---------------------------------------------------------------- material.rs
V3 reflection(V3 normal, V3 i) { return i - 2.0f * normal * dot(normal, i); }
bool refraction(V3 normal, V3 i, float ior, V3& o) {
    float side = dot(normal, i);
    float cosi, eta;
    V3 n;
    if (side < 0.0f) { cosi = -side; eta = 1.0f / ior; n = normal; }
    else { cosi = side; eta = ior; n = -normal; }
    float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (k > 0.0f) {
        o = eta * i + (eta * cosi - std::sqrt(k)) * n;
        return true;
    }
    return false;
}
float fresnel(V3 normal, V3 i, float ior) {
    float side = dot(normal, i);
    float cosi, etat, etai;
    if (side < 0.0f) { cosi = -side; etat = ior; etai = 1.0f; }
    else { cosi = side; etat = 1.0f; etai = ior; }
    float sint = etai / etat * std::sqrt(1.0f - cosi * cosi);
    if (sint < 1.0f) {
        float cost = std::sqrt(1.0f - sint * sint);
        float rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
        float rp = ((etat * cosi) - (etai * cost)) / ((etai * cosi) + (etat * cost));
        return (rs * rs + rp * rp) / 2.0f;
    }
    return 1.0f;
}
static inline int32_t sat_i32(float x) {  // Rust `as i32`: saturating, NaN -> 0
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
Color Texture::color_at(float u, float v) const {
    switch (kind) {
        case TEX_UV: return Color(u, v, 0.0f);                    // texture.rs:22-26
        case TEX_CHECKERBOARD: {                                  // texture.rs:36-47
            int32_t i = sat_i32(u * (float)repeats);
            int32_t j = sat_i32(v * (float)repeats);
            return ((i % 2 == 0) ^ (j % 2 == 0)) ? Color(0.f, 0.f, 0.f) : Color(1.f, 1.f, 1.f);
        }
        case TEX_IMAGE: return image->color_at(u, v);             // texture.rs:70-82
        default: return color;                                    // texture.rs:14-18
    }
}
Color Material::sample_at(float u, float v) const {
    switch (kind) {
        case MAT_NULL: return Color(0.f, 1.f, 0.f);               // material.rs:19-23
        case MAT_LAMBERT: return tex.color_at(u, v);             // :137-141
        case MAT_GLOSSY: return tex.color_at(u, v) * (1.0f - p1);  // :41-43
        default: return Color(0.f, 0.f, 0.f);                    // Metal/Transparent/Glass
    }
}
bool Material::scatter(V3 n, V3 i, SmallRng& rng, V3& out_dir, Color& w) const {
    switch (kind) {
        case MAT_METAL: {  // material.rs:32-35
            out_dir = i - 2.0f * n * dot(n, i);
            w = Color(1.f, 1.f, 1.f);
            return true;
        }
        case MAT_GLOSSY: {  // material.rs:45-69 (the acceptance test uses `out`, not `o`)
            V3 out = reflection(n, i);
            V3 a, b;
            ortho_axes(out, a, b);
            float side = dot(n, i) < 0.0f ? 1.0f : -1.0f;
            for (int k = 0; k < 10; k++) {
                float theta = rng.gen_f32() * 2.0f * PI_F;
                float u = rng.gen_f32();
                float p = std::pow(u, 2.0f / (1.0f + p0));
                float r = std::sqrt(1.0f - p);
                float x = r * std::cos(theta);
                float y = r * std::sin(theta);
                float z = std::sqrt(p);
                V3 o = x * a + y * b + z * out;
                if (dot(n, out) * side > 0.0f) {
                    out_dir = o;
                    w = Color(1.f, 1.f, 1.f) * p1;
                    return true;
                }
            }
            return false;
        }
        case MAT_TRANSPARENT: {  // material.rs:123-133
            V3 o;
            if (rng.gen_f32() < fresnel(n, i, p0)) {
                o = reflection(n, i);
            } else {
                if (!refraction(n, i, p0, o)) return false;
            }
            out_dir = o;
            w = Color(1.f, 1.f, 1.f);
            return true;
        }
        default: return false;  // material.rs:12-14
    }
}

// ---------------------------------------------------------------- light.rs
bool Light::is_delta_distribution() const {
    switch (kind) {
        case LIGHT_AMBIENT: return true;                  // light.rs:29-31
        case LIGHT_POINT: return param == 0.0f;           // :76-78
        case LIGHT_DIRECTION: return param > 0.0f;        // :125-127 (inverted on purpose: spread.is_some())
        default: return false;                            // :7-9
    }
}
void Light::sample_incidence(V3 pos, V3 normal, SmallRng& rng, V3& dir, float& t_max, Color& ill) const {
    switch (kind) {
        case LIGHT_AMBIENT: {  // light.rs:24-27
            dir = normal; t_max = 0.0f; ill = emission;
            return;
        }
        case LIGHT_DIRECTION: {  // light.rs:97-123
            V3 o;
            if (param > 0.0f) {
                float theta = rng.gen_f32() * 2.0f * PI_F;
                float u = rng.gen_f32();
                float p = std::pow(u, param);
                float r = std::sqrt(1.0f - p);
                float x = r * std::cos(theta);
                float y = r * std::sin(theta);
                float z = std::sqrt(p);
                V3 a, b;
                ortho_axes(vec, a, b);
                o = x * a + y * b - z * vec;
            } else {
                o = -vec;
            }
            float c = rust_max(dot(o, normal), 0.0f);
            dir = o; t_max = 1e12f; ill = emission * c;
            return;
        }
        case LIGHT_AO: {  // light.rs:144-158
            float theta = rng.gen_f32() * 2.0f * PI_F;
            float u = rng.gen_f32();
            float r = std::sqrt(u);
            float x = r * std::cos(theta);
            float y = r * std::sin(theta);
            float z = std::sqrt(1.0f - u);
            V3 a, b;
            ortho_axes(normal, a, b);
            dir = a * x + b * y + normal * z;
            t_max = param; ill = emission;
            return;
        }
        case LIGHT_POINT: {  // light.rs:50-74; gen_range(-1,1) = UniformFloat::sample_single (unpinned)
            V3 offset;
            for (;;) {
                float f[3];
                for (int k = 0; k < 3; k++) {
                    for (;;) {
                        uint32_t bits = (rng.next_u32() >> 9) | 0x3F800000u;  // [1,2)
                        float v12;
                        std::memcpy(&v12, &bits, 4);
                        float scale = 1.0f - (-1.0f);
                        float off = -1.0f - scale;
                        float res = v12 * scale + off;
                        if (res < 1.0f) { f[k] = res; break; }
                    }
                }
                V3 fv(f[0], f[1], f[2]);
                if (norm_squared(fv) < 1.0f) { offset = vec + fv * param - pos; break; }
            }
            float dist_sq = norm_squared(offset);
            float dist = std::sqrt(dist_sq);
            dir = offset / dist;
            float c = rust_max(dot(dir, normal), 0.0f);
            t_max = dist; ill = emission / dist_sq * c;
            return;
        }
    }
}

// ---------------------------------------------------------------- triangle.rs:18-38
void moller_trumbore(V3 a, V3 b, V3 c, const Ray& ray, float out[3]) {
    V3 ab = b - a;
    V3 ac = c - a;
    V3 pv = cross(ray.dir, ac);
    float det = dot(ab, pv);
    if (std::fabs(det) < 1e-12f) {
        g_cnt.degenerate_tris++;
        out[0] = 0.0f; out[1] = 0.0f; out[2] = 0.0f;
        return;
    }
    float inv_det = 1.0f / det;
    V3 tv = ray.pos - a;
    V3 qv = cross(tv, ab);
    out[0] = dot(qv, ac) * inv_det;
    out[1] = dot(tv, pv) * inv_det;
    out[2] = dot(qv, ray.dir) * inv_det;
}
bool Triangle::hit(const Ray& ray, float t_max, HitResult& h) const {  // triangle.rs:41-57
    float r[3];
    g_cnt.tri();
    moller_trumbore(a, b, c, ray, r);
    float t = r[0], u = r[1], v = r[2];
    if (t >= 0.0f && t <= t_max && u >= 0.0f && v >= 0.0f && u + v <= 1.0f) {
        h = HitResult();
        h.t = t; h.t_local = t;
        h.norm = cross(b - a, c - a);
        h.pos = ray.at(t);
        h.uv[0] = u; h.uv[1] = v;
        return true;
    }
    return false;
}
bool Triangle::is_hit(const Ray& ray, float t_max) const {  // triangle.rs:59-62
    float r[3];
    g_cnt.tri();
    moller_trumbore(a, b, c, ray, r);
    return r[0] >= 0.0f && r[0] <= t_max && r[1] >= 0.0f && r[2] >= 0.0f && r[1] + r[2] <= 1.0f;
}

// ---------------------------------------------------------------- mesh.rs
AABB MeshTriangle::bounding_box() const {  // mesh.rs:149-157
    const std::vector<V3>& d = data->data;
    return AABB::from_point(d[v[0]]).union_point(d[v[1]]).union_point(d[v[2]]);
}
bool MeshTriangle::hit(const Ray& ray, float t_max, HitResult& h) const {  // mesh.rs:119-147
    const std::vector<V3>& d = data->data;
    V3 a = d[v[0]], b = d[v[1]], c = d[v[2]];
    float r[3];
    g_cnt.tri();
    moller_trumbore(a, b, c, ray, r);
    float t = r[0], u = r[1], w = r[2];
    if (t >= 0.0f && t <= t_max && u >= 0.0f && w >= 0.0f && u + w <= 1.0f) {
        size_t offset = d.size() / 2;
        V3 na = d[offset + v[0]], nb = d[offset + v[1]], nc = d[offset + v[2]];
        h = HitResult();
        h.norm = (1.0f - u - w) * na + u * nb + w * nc;
        h.t = t; h.t_local = t;
        h.pos = ray.at(t);
        h.uv[0] = u; h.uv[1] = w;
        return true;
    }
    return false;
}
std::shared_ptr<Mesh> Mesh::from_triangles(const std::vector<std::array<V3, 3>>& tris) {  // mesh.rs:83-116
    std::vector<V3> vertices;
    std::vector<std::array<uint32_t, 3>> faces;
    // the reference keys a HashMap by the 12 bytes of the vertex; ids are handed out in order of first appearance,
    // so the container's iteration order plays no part
    std::vector<std::array<uint32_t, 3>> keys;
    for (const auto& tri : tris) {
        std::array<uint32_t, 3> face{};
        for (int i = 0; i < 3; i++) {
            std::array<uint32_t, 3> key;
            const float v[3] = {tri[i][0], tri[i][1], tri[i][2]};
            std::memcpy(key.data(), v, 12);
            uint32_t index = (uint32_t)keys.size();
            for (uint32_t k = 0; k < keys.size(); k++)
                if (keys[k] == key) { index = k; break; }
            if (index == keys.size()) { keys.push_back(key); vertices.push_back(tri[i]); }
            face[i] = index;
        }
        faces.push_back(face);
    }
    return from_vertices(vertices, faces);
}
std::shared_ptr<Mesh> Mesh::from_vertices(const std::vector<V3>& vertices,
                                          const std::vector<std::array<uint32_t, 3>>& faces) {
    // mesh.rs:54-81: area-weighted vertex normals, then mesh.rs:21-52
    size_t n = vertices.size();
    std::vector<V3> normals(n, V3::fill(0.0f));
    for (const auto& f : faces) {
        V3 a = vertices[f[0]], b = vertices[f[1]], c = vertices[f[2]];
        V3 e1 = b - a, e2 = c - a;
        V3 nrm = cross(e1, e2);
        normals[f[0]] = normals[f[0]] + nrm;
        normals[f[1]] = normals[f[1]] + nrm;
        normals[f[2]] = normals[f[2]] + nrm;
    }
    for (auto& nv : normals) nv = normalize(nv);
    auto md = std::make_shared<MeshData>();
    md->data = vertices;
    md->data.insert(md->data.end(), normals.begin(), normals.end());
    std::vector<MeshTriangle> tris;
    tris.reserve(faces.size());
    for (size_t k = 0; k < faces.size(); k++) {
        MeshTriangle t;
        t.v[0] = faces[k][0]; t.v[1] = faces[k][1]; t.v[2] = faces[k][2];
        t.data = md;
        t.face = (uint32_t)k;
        tris.push_back(t);
    }
    auto mesh = std::make_shared<Mesh>();
    mesh->data = md;
    mesh->tree.reset(new AABBTree<MeshTriangle>(std::move(tris), 0.1f));  // mesh.rs:50
    return mesh;
}

// ---------------------------------------------------------------- bvh.rs
static inline size_t bucket_index(float ratio) {  // bvh.rs:40: floor() as usize, then .min(15)
    float f = std::floor(ratio * 16.0f);
    size_t idx;
    if (!(f == f)) idx = 0;            // NaN -> 0
    else if (f <= 0.0f) idx = 0;       // negative saturates to 0
    else if (f >= 1.8446744e19f) idx = (size_t)-1;
    else idx = (size_t)f;
    return idx < 15 ? idx : 15;
}

template <class T>
static size_t partition_sah(AABB* bbox, T* objs, size_t n) {  // bvh.rs:10-97
    auto surface = [](const AABB& b) {
        V3 dl = b.max - b.min;
        return dl[0] * dl[1] + dl[1] * dl[2] + dl[2] * dl[0];
    };
    std::vector<V3> centers(n);
    for (size_t k = 0; k < n; k++) centers[k] = (bbox[k].min + bbox[k].max) * 0.5f;

    float best_cost = std::numeric_limits<float>::infinity();
    int best_axis = 0;
    float best_mid = 0.0f;
    const float INF = std::numeric_limits<float>::infinity();

    for (int axis = 0; axis < 3; axis++) {
        float mn = INF, mx = -INF;
        for (size_t k = 0; k < n; k++) {
            mn = min2(mn, centers[k][axis]);
            mx = max2(mx, centers[k][axis]);
        }
        const int NB = 16;
        size_t bucket_sizes[NB];
        AABB bucket_bbox[NB];
        for (int b = 0; b < NB; b++) bucket_sizes[b] = 0;
        for (size_t k = 0; k < n; k++) {
            float ratio = (centers[k][axis] - mn) / (mx - mn);
            size_t index = bucket_index(ratio);
            bucket_sizes[index] += 1;
            bucket_bbox[index] = bucket_bbox[index].union_(bbox[k]);
        }
        for (int i = 1; i < NB; i++) {
            size_t lsize = 0, rsize = 0;
            for (int b = 0; b < i; b++) lsize += bucket_sizes[b];
            for (int b = i; b < NB; b++) rsize += bucket_sizes[b];
            if (lsize == 0 || rsize == 0) continue;
            AABB lbox, rbox;
            for (int b = 0; b < i; b++) lbox = lbox.union_(bucket_bbox[b]);
            for (int b = i; b < NB; b++) rbox = rbox.union_(bucket_bbox[b]);
            float cost = (float)lsize * surface(lbox) + (float)rsize * surface(rbox);
            if (cost < best_cost) {
                float mid = ((float)i / (float)NB) * (mx - mn) + mn;
                best_cost = cost; best_axis = axis; best_mid = mid;
            }
        }
    }

    size_t i = 0, j = n;  // bvh.rs:74-94: centroid >= mid goes LEFT
    for (;;) {
        while (i < n && centers[i][best_axis] >= best_mid) i++;
        while (j > 0 && centers[j - 1][best_axis] < best_mid) j--;
        if (i + 1 < j) {
            std::swap(bbox[i], bbox[j - 1]);
            std::swap(centers[i], centers[j - 1]);
            std::swap(objs[i], objs[j - 1]);
        } else {
            break;
        }
    }
    return i;
}

template <class T>
static void construct_node(std::vector<AABB>& bbox, std::vector<T>& objs, size_t begin, size_t end,
                           std::vector<BvhNode>& nodes, float hit_cost) {  // bvh.rs:100-149
    AABB bb;
    for (size_t k = begin; k < end; k++) bb = bb.union_(bbox[k]);
    size_t index = nodes.size();
    nodes.push_back(BvhNode{AABB(), 0xFFFFFFFFu, 0xFFFFFFFFu});

    size_t len = partition_sah(bbox.data() + begin, objs.data() + begin, end - begin);
    size_t mid = begin + len;
    size_t lsize = mid - begin, rsize = end - mid;
    if (lsize > 0 && rsize > 0) {
        AABB lb;
        for (size_t k = begin; k < mid; k++) lb = lb.union_(bbox[k]);
        float lsurface = lb.surface_area();
        AABB rb;  // bvh.rs:129-132: the "right" surface is folded over the LEFT range
        for (size_t k = begin; k < mid; k++) rb = rb.union_(bbox[k]);
        float rsurface = rb.surface_area();
        float surface = bb.surface_area();
        float split_cost = 1.0f + hit_cost * (lsurface / surface) * (float)lsize +
                           hit_cost * (rsurface / surface) * (float)rsize;
        float nosplit_cost = hit_cost * (float)(lsize + rsize);
        if (split_cost < nosplit_cost) {
            construct_node(bbox, objs, begin, mid, nodes, hit_cost);
            construct_node(bbox, objs, mid, end, nodes, hit_cost);
        }
    }
    nodes[index] = BvhNode{bb, (uint32_t)nodes.size(), (uint32_t)begin};
}

template <class T>
AABBTree<T>::AABBTree(std::vector<T> o, float hit_cost) : objs(std::move(o)) {  // bvh.rs:151-171
    size_t n = objs.size();
    std::vector<AABB> bbox(n);
    for (size_t k = 0; k < n; k++) bbox[k] = objs[k].bounding_box();
    construct_node(bbox, objs, 0, n, nodes, hit_cost);
    uint32_t m = (uint32_t)nodes.size();
    nodes.push_back(BvhNode{AABB(), m, (uint32_t)n});
}

template <class T>
void AABBTree<T>::stats(uint32_t out[5]) const {  // bvh.rs:179-204
    size_t n = nodes.size();
    uint32_t max_leaf = 0, min_leaf = (uint32_t)objs.size(), num_leaf = 0;
    for (size_t i = 0; i + 1 < n; i++) {
        uint32_t size = nodes[i + 1].first - nodes[i].first;
        if (size > 0) {
            min_leaf = min_leaf < size ? min_leaf : size;
            max_leaf = max_leaf > size ? max_leaf : size;
            num_leaf++;
        }
    }
    out[0] = (uint32_t)objs.size(); out[1] = (uint32_t)n; out[2] = num_leaf;
    out[3] = min_leaf; out[4] = max_leaf;
}

static inline void stamp_leaf_index(const MeshTriangle&, HitResult& h, size_t idx) { h.prim = (uint32_t)idx; }
static inline void stamp_leaf_index(const Object&, HitResult&, size_t) {}

// bvh.rs:207-255 specialised into the two wrappers of bvh.rs:259-281
template <class T>
bool AABBTree<T>::hit(const Ray& ray, float t_max, HitResult& result) const {
    size_t i = 0;
    size_t n = nodes.size() - 1;
    bool found = false;
    V3 inv = 1.0f / ray.dir;
    bool neg[3] = {std::signbit(inv[0]), std::signbit(inv[1]), std::signbit(inv[2])};
    while (i < n) {
        const BvhNode& node = nodes[i];
        g_cnt.node();
        float t0, t1;
        if (node.bb.fast_intersect_ray(ray.pos, inv, neg, t0, t1)) {
            if (t0 <= t_max + 0.01f && t1 >= -0.01f) {
                size_t begin = node.first, end = nodes[i + 1].first;
                for (size_t k = begin; k < end; k++) {
                    HitResult h;
                    if (objs[k].hit(ray, t_max, h)) {
                        stamp_leaf_index(objs[k], h, k);
                        t_max = h.t;
                        result = h;
                        found = true;
                    }
                }
                i += 1;
            } else {
                i = node.skip;
            }
        } else {
            i = node.skip;
        }
    }
    return found;
}
template <class T>
bool AABBTree<T>::is_hit(const Ray& ray, float t_max) const {
    size_t i = 0;
    size_t n = nodes.size() - 1;
    V3 inv = 1.0f / ray.dir;
    bool neg[3] = {std::signbit(inv[0]), std::signbit(inv[1]), std::signbit(inv[2])};
    while (i < n) {
        const BvhNode& node = nodes[i];
        g_cnt.node();
        float t0, t1;
        if (node.bb.fast_intersect_ray(ray.pos, inv, neg, t0, t1)) {
            if (t0 <= t_max + 0.01f && t1 >= -0.01f) {
                size_t begin = node.first, end = nodes[i + 1].first;
                for (size_t k = begin; k < end; k++) {
                    if (objs[k].is_hit(ray, t_max)) return true;
                }
                i += 1;
            } else {
                i = node.skip;
            }
        } else {
            i = node.skip;
        }
    }
    return false;
}
template struct AABBTree<MeshTriangle>;
template struct AABBTree<Object>;

// ---------------------------------------------------------------- sphere.rs
static inline bool sphere_intersect(const Ray& ray, float& t0, float& t1) {  // sphere.rs:22-33
    float a = -dot(ray.dir, ray.pos);
    float b = a * a - dot(ray.pos, ray.pos) + 1.0f;
    if (b >= 0.0f) {
        float d = std::sqrt(b);
        t0 = a - d; t1 = a + d;
        return true;
    }
    return false;
}
bool UnitSphere::hit(const Ray& ray, float t_max, HitResult& h) const {  // sphere.rs:36-60
    g_cnt.sphere();
    float t0, t1;
    if (!sphere_intersect(ray, t0, t1)) return false;
    float t;
    if (t0 >= 0.0f && t0 <= t_max) t = t0;
    else if (t1 >= 0.0f && t1 <= t_max) t = t1;
    else return false;
    V3 pos = ray.at(t);
    V3 nrm = pos;
    float u = std::atan2(nrm[0], nrm[1]) / PI_F * 0.5f + 0.5f;
    float v = std::acos(nrm[2]) / PI_F;
    h = HitResult();
    h.t = t; h.t_local = t; h.norm = nrm; h.pos = pos; h.uv[0] = u; h.uv[1] = v;
    return true;
}
bool UnitSphere::is_hit(const Ray& ray, float t_max) const {  // sphere.rs:62-69
    g_cnt.sphere();
    float t0, t1;
    if (sphere_intersect(ray, t0, t1)) return t0 <= t_max && t1 >= 0.0f;
    return false;
}
GeomPtr make_sphere(V3 center, float radius) {  // sphere.rs:14-19
    GeomPtr unit = std::make_shared<UnitSphere>();
    GeomPtr sc = std::make_shared<Scale>(unit, radius);
    return std::make_shared<Translate>(sc, center);
}

// ---------------------------------------------------------------- cuboid.rs
static inline void ray_box_intersection(const Ray& ray, V3 mn, V3 mx, float& t0, float& t1) {  // :8-16
    V3 a = (mn - ray.pos) * ray.inv_dir;
    V3 b = (mx - ray.pos) * ray.inv_dir;
    t0 = max3(min2(a[0], b[0]), min2(a[1], b[1]), min2(a[2], b[2]));
    t1 = min3(max2(a[0], b[0]), max2(a[1], b[1]), max2(a[2], b[2]));
}
bool UnitCuboid::hit(const Ray& ray, float t_max, HitResult& h) const {  // cuboid.rs:19-48
    g_cnt.cuboid();
    float t0, t1;
    ray_box_intersection(ray, -V3::fill(1.0f), V3::fill(1.0f), t0, t1);
    float t;
    if (t0 > t1 || t0 > t_max || t1 < 0.0f) return false;
    else if (t0 > 0.0f) t = t0;
    else t = t1;
    V3 p = ray.at(t);
    V3 pa(std::fabs(p[0]), std::fabs(p[1]), std::fabs(p[2]));
    V3 n;
    float u, v;
    if (pa[0] > pa[1] && pa[0] > pa[2]) { n = V3(p[0], 0.0f, 0.0f); u = p[1]; v = p[2]; }
    else if (pa[1] > pa[2]) { n = V3(0.0f, p[1], 0.0f); u = p[0]; v = p[2]; }
    else { n = V3(0.0f, 0.0f, p[2]); u = p[0]; v = p[1]; }
    h = HitResult();
    h.pos = p; h.norm = n; h.uv[0] = u; h.uv[1] = v; h.t = t; h.t_local = t;
    return true;
}
bool UnitCuboid::is_hit(const Ray& ray, float t_max) const {  // cuboid.rs:50-54
    g_cnt.cuboid();
    float t0, t1;
    ray_box_intersection(ray, -V3::fill(1.0f), V3::fill(1.0f), t0, t1);
    return t0 <= t1 && t0 <= t_max && t1 >= 0.0f;
}
Cuboid::Cuboid(V3 a, V3 b) {  // cuboid.rs:69-78
    center = (a + b) * 0.5f;
    V3 e = a - center;
    extent = V3(std::fabs(e[0]), std::fabs(e[1]), std::fabs(e[2]));
    inv_extent = 1.0f / extent;
}
bool Cuboid::hit(const Ray& ray, float t_max, HitResult& h) const {  // cuboid.rs:82-93
    V3 p = (ray.pos - center) * inv_extent;
    V3 d = ray.dir * inv_extent;
    Ray nr(p, d);
    UnitCuboid uc;
    if (uc.hit(nr, t_max, h)) {
        h.pos = h.pos * extent + center;
        return true;
    }
    return false;
}
bool Cuboid::is_hit(const Ray& ray, float t_max) const {  // cuboid.rs:95-100
    g_cnt.cuboid();
    float t0, t1;
    ray_box_intersection(ray, center - extent, center + extent, t0, t1);
    return t0 <= t1 && t0 <= t_max && t1 >= 0.0f;
}

// ---------------------------------------------------------------- transform.rs
AABB Translate::bounding_box() const {  // :75-78
    AABB b = obj->bounding_box();
    return AABB(b.min + offset, b.max + offset);
}
bool Translate::hit(const Ray& ray, float t_max, HitResult& h) const {  // :57-66
    Ray nr(ray.pos - offset, ray.dir);
    if (obj->hit(nr, t_max, h)) {
        h.pos = h.pos + offset;
        return true;
    }
    return false;
}
bool Translate::is_hit(const Ray& ray, float t_max) const {  // :69-72
    Ray nr(ray.pos - offset, ray.dir);
    return obj->is_hit(nr, t_max);
}
AABB Scale::bounding_box() const {  // :123-127
    AABB b = obj->bounding_box();
    return AABB(b.min * scale, b.max * scale);
}
bool Scale::hit(const Ray& ray, float t_max, HitResult& h) const {  // :102-113
    Ray nr(ray.pos * inv_scale, ray.dir);
    if (obj->hit(nr, t_max * inv_scale, h)) {
        h.t *= scale;
        h.pos = h.pos * scale;
        return true;
    }
    return false;
}
bool Scale::is_hit(const Ray& ray, float t_max) const {  // :116-120
    Ray nr(ray.pos * inv_scale, ray.dir);
    return obj->is_hit(nr, t_max * inv_scale);
}
AABB Rotate::bounding_box() const {  // :201-220
    AABB old = obj->bounding_box();
    V3 corners[2] = {old.min, old.max};
    AABB nb;
    for (int i = 0; i < 8; i++) {
        V3 p(corners[i % 2][0], corners[(i / 2) % 2][1], corners[(i / 4) % 2][2]);
        nb = nb.union_point(mat.apply(p));
    }
    return nb;
}
bool Rotate::hit(const Ray& ray, float t_max, HitResult& h) const {  // :177-189
    V3 p = mat.transpose_apply(ray.pos);
    V3 d = mat.transpose_apply(ray.dir);
    Ray nr(p, d);
    if (obj->hit(nr, t_max, h)) {
        h.pos = mat.apply(h.pos);
        h.norm = mat.apply(h.norm);
        return true;
    }
    return false;
}
bool Rotate::is_hit(const Ray& ray, float t_max) const {  // :192-198
    V3 p = mat.transpose_apply(ray.pos);
    V3 d = mat.transpose_apply(ray.dir);
    Ray nr(p, d);
    return obj->is_hit(nr, t_max);
}
TransformBuilder& TransformBuilder::rotate(V3 axis, float angle) {  // transform.rs:255-260
    V3 off = mat.transpose_apply(offset);
    mat = Mat3::multiply(Mat3::new_rotation(axis, angle), mat);  // :138-142
    offset = mat.apply(off);
    return *this;
}
TransformBuilder& TransformBuilder::reflect(V3 axis) {  // transform.rs:274-279, Rotate::reflect :156-160
    V3 off = mat.transpose_apply(offset);
    mat = Mat3::multiply(Mat3::new_reflection(axis), mat);
    offset = mat.apply(off);
    return *this;
}
GeomPtr TransformBuilder::build() const {  // Translate<Scale<Rotate<T>>>, transform.rs:224-230
    GeomPtr r = std::make_shared<Rotate>(obj, mat);
    GeomPtr s = std::make_shared<Scale>(r, scale_);
    return std::make_shared<Translate>(s, offset);
}

// ---------------------------------------------------------------- aggregate.rs
bool Object::hit(const Ray& ray, float t_max, HitResult& h) const {  // :104-111
    if (geometry->hit(ray, t_max, h)) {
        h.material = &material;
        if (h.obj < 0) h.obj = index;
        return true;
    }
    return false;
}
AABB GeometryList::bounding_box() const {  // :47-55
    AABB b;
    for (const auto& g : items) b = b.union_(g->bounding_box());
    return b;
}
BoundingBox::BoundingBox(GeomPtr o) : obj(o) {  // aggregate.rs:62-69
    AABB bbox = o->bounding_box();
    V3 delta = bbox.max - bbox.min;
    V3 pad = V3::fill(delta[0] + delta[1] * delta[2]) * 0.001f;
    box = AABB::from_points(bbox.min - pad, bbox.max + pad);
}
bool BoundingBox::hit(const Ray& ray, float t_max, HitResult& h) const {  // aggregate.rs:73-81
    float t_in, t_out;
    if (box.intersect_ray(ray.pos, ray.inv_dir, t_in, t_out)) {
        if (t_in <= t_max && t_out >= 0.0f) return obj->hit(ray, t_max, h);
    }
    return false;
}
bool BoundingBox::is_hit(const Ray& ray, float t_max) const {  // aggregate.rs:83-91
    float t_in, t_out;
    if (box.intersect_ray(ray.pos, ray.inv_dir, t_in, t_out)) {
        if (t_in <= t_max && t_out >= 0.0f) return obj->is_hit(ray, t_max);
    }
    return false;
}
bool GeometryList::hit(const Ray& ray, float t_max, HitResult& result) const {  // :24-35
    bool found = false;
    for (const auto& g : items) {
        HitResult h;
        if (g->hit(ray, t_max, h)) {
            t_max = h.t;
            result = h;
            found = true;
        }
    }
    return found;
}
bool GeometryList::is_hit(const Ray& ray, float t_max) const {  // :37-45
    for (const auto& g : items)
        if (g->is_hit(ray, t_max)) return true;
    return false;
}

// ---------------------------------------------------------------- scene.rs
Camera::Camera(size_t w, size_t h)  // scene.rs:35-48
    : width(w), height(h), pos(V3::fill(0.0f)), dir(0.f, 0.f, 1.f), horizontal(1.f, 0.f, 0.f),
      vertical(0.f, 1.f, 0.f) {
    look_towards(V3(0.f, 0.f, 1.f), V3(0.f, 1.f, 0.f));
    perspective(60.0f);
}
Camera& Camera::look_towards(V3 d, V3 up) {  // scene.rs:55-64
    d = normalize(d);
    V3 horz = normalize(cross(up, d));
    V3 vert = normalize(cross(d, horz));
    dir = d;
    horizontal = horz * norm(horizontal);
    vertical = vert * norm(vertical);
    return *this;
}
Camera& Camera::perspective(float fov) {  // scene.rs:71-78
    float fac = std::tan((fov / 2.0f) * (PI_F / 180.0f));  // f32::to_radians = x * (PI / 180)
    float aspect = (float)height / (float)width;
    horizontal = horizontal * (fac / norm(horizontal));
    vertical = vertical * (fac / norm(vertical) * aspect);
    return *this;
}
Ray Camera::generate_ray(float x, float y) const {  // scene.rs:84-90
    float u = 2.0f * (x / (float)width) - 1.0f;
    float v = 2.0f * (y / (float)height) - 1.0f;
    V3 d = dir + u * horizontal + v * vertical;
    return Ray(pos, normalize(d));
}
Color Scene::calculate_background(const Ray& ray) const {  // scene.rs:16-21
    float x = ray.dir[0], y = ray.dir[1], z = ray.dir[2];
    float u = std::atan2(x, y) / PI_F * 0.5f + 0.5f;
    float v = std::acos(z) / PI_F;
    return skybox->color_at(u, v);
}

// ---------------------------------------------------------------- integrator.rs
Color WhittedIntegrator::calculate_pixel(const Scene& scene, size_t cx, size_t cy) const {  // :33-56
    int n = antialiasing;
    uint64_t seed = (uint64_t)cx ^ __builtin_bswap64((uint64_t)cy);  // cx.to_le() ^ cy.to_be()
    SmallRng rng = SmallRng::seed_from_u64(seed);
    std::vector<Color> pixels;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            float x = (float)cx + ((float)i + 0.5f) / (float)n - 0.5f;
            float y = (float)cy + ((float)j + 0.5f) / (float)n - 0.5f;
            Ray ray = scene.camera.generate_ray(x, y);
            pixels.push_back(integrate_recur(scene, ray, 0, rng));
        }
    }
    float len = (float)pixels.size();
    Color sum = V3::fill(0.0f);
    for (const Color& p : pixels) sum = sum + p;
    return sum / len;
}
Color WhittedIntegrator::integrate_recur(const Scene& scene, const Ray& ray, int depth,
                                         SmallRng& rng) const {  // :58-105
    if (depth >= max_depth) return scene.calculate_background(ray);
    HitResult hit;
    g_cnt.closest_rays++;
    const bool found = scene.root->hit(ray, 1e12f, hit);
    if (g_path_log) {
        const float rec[11] = {(float)depth, found ? (float)hit.obj : -1.0f, found ? (float)hit.prim : 0.0f, found ? (float)hit.material->kind : -1.0f,
                               found ? hit.t : 0.0f, ray.pos[0], ray.pos[1], ray.pos[2], ray.dir[0], ray.dir[1], ray.dir[2]};
        g_path_log->insert(g_path_log->end(), rec, rec + 11);
    }
    if (!found) return scene.calculate_background(ray);

    Color color = V3::fill(0.0f);
    float u = hit.uv[0], v = hit.uv[1];
    V3 n = normalize(hit.norm);
    bool outside = dot(n, -ray.dir) > 0.0f;
    V3 p_out = hit.pos + n * 0.001f;
    V3 p_in = hit.pos - n * 0.001f;
    V3 p = outside ? p_out : p_in;

    Color diffuse = hit.material->sample_at(u, v);
    if (!is_zero(diffuse)) {
        V3 illum = V3::fill(0.0f);
        for (const Light& light : scene.lights) illum = illum + illumination(scene, light, p, n, rng);
        color = color + diffuse * illum;
    }
    for (int k = 0; k < scatter_rays; k++) {
        V3 out;
        Color w;
        if (hit.material->scatter(n, ray.dir, rng, out, w)) {
            V3 p2 = dot(out, n) > 0.0f ? p_out : p_in;
            color = color + w * integrate_recur(scene, Ray(p2, out), depth + 1, rng) / (float)scatter_rays;
        }
    }
    return color;
}
Color WhittedIntegrator::illumination(const Scene& scene, const Light& light, V3 pos, V3 normal,
                                      SmallRng& rng) const {  // :107-132
    Color total = V3::fill(0.0f);
    int n = light.is_delta_distribution() ? 1 : shadow_rays;
    for (int k = 0; k < n; k++) {
        V3 dir;
        float t_max;
        Color ill;
        light.sample_incidence(pos, normal, rng, dir, t_max, ill);
        Ray ray(pos, dir);
        bool visible = true;
        if (!(t_max == 0.0f)) {
            g_cnt.any_rays++;
            g_cnt.in_any = true;
            visible = !scene.root->is_hit(ray, t_max);
            g_cnt.in_any = false;
        }
        if (visible) total = total + ill;
    }
    return total / (float)n;
}

// ---------------------------------------------------------------- loader/ply.rs
static std::vector<std::string> split_ws(const std::string& s) {
    std::vector<std::string> out;
    std::istringstream is(s);
    std::string tok;
    while (is >> tok) out.push_back(tok);
    return out;
}
static bool norm_type(const std::vector<std::string>& parts, std::string& out) {  // ply.rs:22-42
    auto one = [](const std::string& t, std::string& o) {
        if (t == "char" || t == "int8") o = "int8";
        else if (t == "uchar" || t == "uint8") o = "uint8";
        else if (t == "short" || t == "int16") o = "int16";
        else if (t == "ushort" || t == "uint16") o = "uint16";
        else if (t == "int" || t == "int32") o = "int32";
        else if (t == "uint" || t == "uint32") o = "uint32";
        else if (t == "float" || t == "float32") o = "float32";
        else if (t == "double" || t == "float64") o = "float64";
        else return false;
        return true;
    };
    if (parts.size() == 1) return one(parts[0], out);
    if (parts.size() == 3 && parts[0] == "list") {
        std::string a, b;
        if (!one(parts[1], a) || !one(parts[2], b)) return false;
        out = "list " + a + " " + b;
        return true;
    }
    return false;
}
bool load_ply(const std::string& path, std::vector<V3>& vertices,
              std::vector<std::array<uint32_t, 3>>& faces, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = "Error while reading file"; return false; }
    std::string line;
    size_t lineno = 0;
    std::vector<std::string> cur;
    auto next = [&]() -> const std::vector<std::string>& {
        cur.clear();
        if (std::getline(f, line)) { cur = split_ws(line); lineno++; }
        return cur;
    };
    auto fail = [&](const std::string& m) { err = "Parse error at line " + std::to_string(lineno) + ": " + m; return false; };
    if (next() != std::vector<std::string>{"ply"}) return fail("expecting 'ply'");
    if (next() != std::vector<std::string>{"format", "ascii", "1.0"}) return fail("expected 'format ascii 1.0'");
    next();
    struct Segment { std::string name; size_t num; std::vector<std::pair<std::string, std::string>> props; };
    std::vector<Segment> segs;
    for (;;) {
        if (cur.empty()) break;
        if (cur[0] == "comment") { next(); continue; }
        if (cur[0] != "element") break;
        Segment s;
        s.name = cur.size() > 1 ? cur[1] : "";  // reference takes token 0 here (ply.rs:108); fixed
        if (cur.size() < 3) return fail("failed to parse number");
        try { s.num = (size_t)std::stoull(cur[2]); } catch (...) { return fail("failed to parse number"); }
        for (;;) {
            next();
            if (cur.empty()) break;
            if (cur[0] == "comment") continue;
            if (cur[0] != "property") break;
            std::vector<std::string> rest(cur.begin() + 1, cur.end());
            std::string key = rest.empty() ? "" : rest.back();
            if (!rest.empty()) rest.pop_back();
            std::string typ;
            if (!norm_type(rest, typ)) return fail("invalid type");
            s.props.push_back({key, typ});
        }
        segs.push_back(s);
    }
    if (cur != std::vector<std::string>{"end_header"}) return fail("expected 'end_header'");
    for (const Segment& s : segs) {
        if (s.name == "vertex") {  // ply.rs:146-196
            int xi = -1, yi = -1, zi = -1;
            for (size_t k = 0; k < s.props.size(); k++) {
                if (s.props[k].second != "float32") continue;
                if (s.props[k].first == "x") xi = (int)k;
                else if (s.props[k].first == "y") yi = (int)k;
                else if (s.props[k].first == "z") zi = (int)k;
            }
            if (xi < 0) return fail("vertex has no x property");
            if (yi < 0) return fail("vertex has no y property");
            if (zi < 0) return fail("vertex has no z property");
            for (size_t k = 0; k < s.num; k++) {
                next();
                if (cur.size() != s.props.size()) return fail("expected " + std::to_string(s.props.size()) + " elements");
                char* e;
                float x = std::strtof(cur[xi].c_str(), &e); if (*e) return fail("failed to parse x coordinate");
                float y = std::strtof(cur[yi].c_str(), &e); if (*e) return fail("failed to parse y coordinate");
                float z = std::strtof(cur[zi].c_str(), &e); if (*e) return fail("failed to parse z coordinate");
                vertices.push_back(V3(x, y, z));
            }
        } else if (s.name == "face") {  // ply.rs:198-252
            int index = -1;
            for (size_t k = 0; k < s.props.size(); k++)
                if ((s.props[k].first == "vertex_indices" || s.props[k].first == "vertex_index") &&
                    s.props[k].second == "list uint8 int32") { index = (int)k; break; }
            if (index < 0) return fail("face has not vertex_indices property");
            std::vector<uint32_t> idx;
            for (size_t k = 0; k < s.num; k++) {
                next();
                if ((size_t)index >= cur.size()) return fail("failed to parse integer");
                size_t n;
                try { n = (size_t)std::stoull(cur[index]); } catch (...) { return fail("failed to parse integer"); }
                if (n < 3 || index + n >= cur.size()) return fail("invalid number of vertices");
                idx.clear();
                for (size_t q = index + 1; q < index + 1 + n; q++) {
                    unsigned long long x;
                    try { x = std::stoull(cur[q]); } catch (...) { return fail("failed to parse integer"); }
                    if (x >= vertices.size()) return fail("invalid vertex identifier");
                    idx.push_back((uint32_t)x);
                }
                for (size_t q = 2; q < idx.size(); q++) faces.push_back({idx[0], idx[q - 1], idx[q]});  // :246-248
            }
        }
    }
    return true;
}

}  // namespace orc
