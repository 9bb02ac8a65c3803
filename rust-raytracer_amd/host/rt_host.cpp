// Host-side scene construction, BVH build, flattening and the render entry points (see rt_host.hpp).
// Build with -ffp-contract=off: the tree builder and the camera/transform set-up must round exactly
// like the reference's f32 arithmetic, because tree shape and instance matrices feed the GPU kernels.
#include "rt_host.hpp"

#include <map>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace rth {

static const float INF = std::numeric_limits<float>::infinity();
static inline float sel_max(float a, float b) { return a > b ? a : b; }  // max!, common.rs:32-41
static inline float sel_min(float a, float b) { return a < b ? a : b; }  // min!, common.rs:43-53

// ---- math ----------------------------------------------------------------------------------------
float Vec3D::norm() const { return std::sqrt(dot(*this, *this)); }
Vec3D Vec3D::normalize() const {  // vec3d.rs:100-112
    float len = norm();
    if (len > 0.0f) return *this / len;
    return fill(std::numeric_limits<float>::quiet_NaN());
}
AABB::AABB() : min(Vec3D::fill(INF)), max(Vec3D::fill(-INF)) {}
AABB AABB::union_with(const AABB& o) const {
    return AABB(Vec3D(sel_min(min.x, o.min.x), sel_min(min.y, o.min.y), sel_min(min.z, o.min.z)),
                Vec3D(sel_max(max.x, o.max.x), sel_max(max.y, o.max.y), sel_max(max.z, o.max.z)));
}
float AABB::surface_area() const {
    Vec3D d = max - min;
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
Mat3D Mat3D::identity() {
    Mat3D r;
    for (int i = 0; i < 9; i++) r.m[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    return r;
}
Mat3D Mat3D::new_rotation(Vec3D axis, float angle) {
    axis = axis.normalize();
    float x = axis.x, y = axis.y, z = axis.z;
    float c = std::cos(angle), s = std::sin(angle);
    Mat3D r;
    r.m[0] = c + x * x * (1.0f - c);
    r.m[1] = x * y * (1.0f - c) - z * s;
    r.m[2] = x * z * (1.0f - c) + y * s;
    r.m[3] = y * x * (1.0f - c) + z * s;
    r.m[4] = c + y * y * (1.0f - c);
    r.m[5] = y * z * (1.0f * c) - x * s;  // as the reference writes it (mat3d.rs:59)
    r.m[6] = z * x * (1.0f - c) - y * s;
    r.m[7] = z * y * (1.0f - c) + x * s;
    r.m[8] = c + z * z * (1.0f - c);
    return r;
}
Mat3D Mat3D::new_reflection(Vec3D axis) {  // mat3d.rs:66-80
    axis = axis.normalize();
    float a = axis.x, b = axis.y, c = axis.z;
    Mat3D r;
    r.m[0] = 1.0f - 2.0f * a * a;
    r.m[1] = 0.0f - 2.0f * a * b;
    r.m[2] = 0.0f - 2.0f * a * c;
    r.m[3] = 0.0f - 2.0f * a * b;
    r.m[4] = 1.0f - 2.0f * b * b;
    r.m[5] = 0.0f - 2.0f * b * c;
    r.m[6] = 0.0f - 2.0f * a * c;
    r.m[7] = 0.0f * 2.0f * b * c;  // as the reference writes it (mat3d.rs:78)
    r.m[8] = 1.0f - 2.0f * c * c;
    return r;
}
Mat3D Mat3D::multiply(const Mat3D& a, const Mat3D& b) {
    Mat3D r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float v = 0.0f;
            for (int k = 0; k < 3; k++) v += a.m[3 * i + k] * b.m[3 * k + j];
            r.m[3 * i + j] = v;
        }
    return r;
}
Vec3D Mat3D::apply(Vec3D a) const {
    float in[3] = {a.x, a.y, a.z}, out[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i] += m[3 * i + j] * in[j];
    return Vec3D(out[0], out[1], out[2]);
}
Vec3D Mat3D::transpose_apply(Vec3D a) const {
    float in[3] = {a.x, a.y, a.z}, out[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i] += m[3 * j + i] * in[j];
    return Vec3D(out[0], out[1], out[2]);
}

// ---- materials / lights ----------------------------------------------------------------------------
static Material mk_mat(uint32_t kind, float p0, float p1, const Texture& t) {
    Material m;
    m.m.kind = kind; m.m.p0 = p0; m.m.p1 = p1;
    m.m.color[0] = t.color.x; m.m.color[1] = t.color.y; m.m.color[2] = t.color.z;
    m.m.tex_kind = t.kind; m.m.tex_param = t.param;   // image textures: the index is assigned when the scene is flattened
    m.image = t.image;
    if (t.kind == RTAMD_TEX_IMAGE && !t.image) throw std::runtime_error("image texture without an image");
    return m;
}
Material NullMaterial() { return mk_mat(RTAMD_MAT_NULL, 0.f, 0.f, Texture(COLOR_WHITE)); }
Material Metal() { return mk_mat(RTAMD_MAT_METAL, 0.f, 0.f, Texture(COLOR_WHITE)); }
Material Glossy(float e, float k, Texture t) { return mk_mat(RTAMD_MAT_GLOSSY, e, k, t); }
Material Transparent(float ior) { return mk_mat(RTAMD_MAT_TRANSPARENT, ior, 0.f, Texture(COLOR_WHITE)); }
Material Lambartian(Texture t) { return mk_mat(RTAMD_MAT_LAMBERT, 0.f, 0.f, t); }

static Light mk_light(uint32_t kind, Vec3D v, float param, Color emission) {
    Light l;
    l.l.kind = kind;
    l.l.vec[0] = v.x; l.l.vec[1] = v.y; l.l.vec[2] = v.z;
    l.l.param = param;
    l.l.emission[0] = emission.x; l.l.emission[1] = emission.y; l.l.emission[2] = emission.z;
    return l;
}
Light AmbientLight(Color color, float intensity) { return mk_light(RTAMD_LIGHT_AMBIENT, Vec3D::zero(), 0.f, color * intensity); }
Light DirectionLight(Vec3D dir, float spread, Color color, float intensity) {  // light.rs:87-95
    return mk_light(RTAMD_LIGHT_DIRECTION, dir.normalize(), spread > 0.0f ? spread : 0.0f, color * intensity);
}
Light AmbientOcclusion(float dist, Color color, float intensity) { return mk_light(RTAMD_LIGHT_AO, Vec3D::zero(), std::fabs(dist), color * intensity); }
Light PointLight(Vec3D pos, float radius, Color color, float intensity) { return mk_light(RTAMD_LIGHT_POINT, pos, radius, color * intensity); }

std::shared_ptr<Image> Image::from_rgb(const uint8_t* rgb, uint32_t w, uint32_t h) {
    auto img = std::make_shared<Image>();
    img->width = w; img->height = h;
    img->rgb.assign(rgb, rgb + (size_t)w * h * 3);
    return img;
}
std::shared_ptr<Image> Image::open_ppm(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string magic;
    f >> magic;
    if (magic != "P6") throw std::runtime_error(path + ": only binary PPM (P6) images are supported");
    auto skip = [&]() {
        for (;;) {
            int c = f.peek();
            if (c == '#') { std::string l; std::getline(f, l); }
            else if (std::isspace(c)) f.get();
            else break;
        }
    };
    uint32_t w, h, maxv;
    skip(); f >> w; skip(); f >> h; skip(); f >> maxv;
    f.get();
    if (!f || maxv != 255) throw std::runtime_error(path + ": bad PPM header");
    auto img = std::make_shared<Image>();
    img->width = w; img->height = h;
    img->rgb.resize((size_t)w * h * 3);
    f.read((char*)img->rgb.data(), (std::streamsize)img->rgb.size());
    if (!f) throw std::runtime_error(path + ": truncated PPM");
    return img;
}

// ---- BVH build (bvh.rs:10-171) ---------------------------------------------------------------------
namespace {
struct Builder {
    std::vector<AABB> box;          // permuted together with `order`
    std::vector<uint32_t> order;
    std::vector<Vec3D> centers;     // scratch for the range being partitioned
    std::vector<rtamd_bvh_node> nodes;
    float hit_cost;

    static float half_area(const AABB& b) {
        Vec3D d = b.max - b.min;
        return d.x * d.y + d.y * d.z + d.z * d.x;
    }
    static size_t bucket_of(float ratio) {  // `(ratio * 16).floor() as usize`.min(15): saturating cast, NaN -> 0
        float f = std::floor(ratio * 16.0f);
        if (!(f == f) || f <= 0.0f) return 0;
        if (f >= 15.0f) return 15;
        return (size_t)f;
    }
    // Returns the size of the left part; objects whose centroid is >= the split plane go left.
    size_t partition(size_t begin, size_t end) {
        const size_t n = end - begin;
        centers.resize(n);
        for (size_t k = 0; k < n; k++) centers[k] = (box[begin + k].min + box[begin + k].max) * 0.5f;
        float best_cost = INF, best_mid = 0.0f;
        int best_axis = 0;
        for (int axis = 0; axis < 3; axis++) {
            float mn = INF, mx = -INF;
            for (size_t k = 0; k < n; k++) {
                float c = centers[k].get(axis);
                mn = sel_min(mn, c);
                mx = sel_max(mx, c);
            }
            size_t count[16] = {0};
            AABB bb[16];
            for (size_t k = 0; k < n; k++) {
                float ratio = (centers[k].get(axis) - mn) / (mx - mn);
                size_t b = bucket_of(ratio);
                count[b]++;
                bb[b] = bb[b].union_with(box[begin + k]);
            }
            for (int i = 1; i < 16; i++) {
                size_t l = 0, r = 0;
                for (int b = 0; b < i; b++) l += count[b];
                for (int b = i; b < 16; b++) r += count[b];
                if (l == 0 || r == 0) continue;
                AABB lb, rb;
                for (int b = 0; b < i; b++) lb = lb.union_with(bb[b]);
                for (int b = i; b < 16; b++) rb = rb.union_with(bb[b]);
                float cost = (float)l * half_area(lb) + (float)r * half_area(rb);
                if (cost < best_cost) {
                    best_cost = cost; best_axis = axis;
                    best_mid = ((float)i / 16.0f) * (mx - mn) + mn;
                }
            }
        }
        size_t i = 0, j = n;
        for (;;) {
            while (i < n && centers[i].get(best_axis) >= best_mid) i++;
            while (j > 0 && centers[j - 1].get(best_axis) < best_mid) j--;
            if (i + 1 < j) {
                std::swap(box[begin + i], box[begin + j - 1]);
                std::swap(order[begin + i], order[begin + j - 1]);
                std::swap(centers[i], centers[j - 1]);
            } else {
                break;
            }
        }
        return i;
    }
    AABB bounds(size_t b, size_t e) const {
        AABB r;
        for (size_t k = b; k < e; k++) r = r.union_with(box[k]);
        return r;
    }
    void node(size_t begin, size_t end) {
        AABB bb = bounds(begin, end);
        size_t index = nodes.size();
        nodes.push_back(rtamd_bvh_node{{0, 0, 0}, {0, 0, 0}, 0xFFFFFFFFu, 0xFFFFFFFFu});
        size_t mid = begin + partition(begin, end);
        size_t l = mid - begin, r = end - mid;
        if (l > 0 && r > 0) {
            float ls = bounds(begin, mid).surface_area();
            float rs = bounds(begin, mid).surface_area();  // the reference folds the left range here too (bvh.rs:129-132)
            float s = bb.surface_area();
            float split_cost = 1.0f + hit_cost * (ls / s) * (float)l + hit_cost * (rs / s) * (float)r;
            float nosplit_cost = hit_cost * (float)(l + r);
            if (split_cost < nosplit_cost) {
                node(begin, mid);
                node(mid, end);
            }
        }
        rtamd_bvh_node nd;
        nd.min[0] = bb.min.x; nd.min[1] = bb.min.y; nd.min[2] = bb.min.z;
        nd.max[0] = bb.max.x; nd.max[1] = bb.max.y; nd.max[2] = bb.max.z;
        nd.skip = (uint32_t)nodes.size();
        nd.first = (uint32_t)begin;
        nodes[index] = nd;
    }
};
}  // namespace

bool& bvh_print_stats_enabled() {
    static bool on = false;
    return on;
}
// AABBTree::print_stats (bvh.rs:179-204): the line the reference prints after every AABBTree::new.
void print_stats(const BvhTree& t) {
    uint32_t s[5];
    bvh_statistics(t, s);
    std::printf("BVH statistics: %u objs, %u nodes, %u leafs, min/max/avg leaf: %u/%u/%g\n", s[0], s[1], s[2], s[3], s[4], (double)((float)s[0] / (float)s[2]));
    std::fflush(stdout);
}
BvhTree build_aabb_tree(const std::vector<AABB>& boxes, float hit_cost) {
    Builder b;
    b.box = boxes;
    b.order.resize(boxes.size());
    for (size_t k = 0; k < boxes.size(); k++) b.order[k] = (uint32_t)k;
    b.hit_cost = hit_cost;
    b.node(0, boxes.size());
    rtamd_bvh_node sentinel;
    AABB e;
    sentinel.min[0] = e.min.x; sentinel.min[1] = e.min.y; sentinel.min[2] = e.min.z;
    sentinel.max[0] = e.max.x; sentinel.max[1] = e.max.y; sentinel.max[2] = e.max.z;
    sentinel.skip = (uint32_t)b.nodes.size();
    sentinel.first = (uint32_t)boxes.size();
    b.nodes.push_back(sentinel);
    BvhTree t;
    t.nodes = std::move(b.nodes);
    t.order = std::move(b.order);
    if (bvh_print_stats_enabled()) print_stats(t);   // bvh.rs:173-177
    return t;
}
BvhTree build_aabb_tree(Device& dev, const std::vector<AABB>& boxes, float hit_cost) {
    if (boxes.empty()) return build_aabb_tree(boxes, hit_cost);
    std::vector<float> flat(boxes.size() * 6);
    for (size_t k = 0; k < boxes.size(); k++) {
        flat[6 * k] = boxes[k].min.x; flat[6 * k + 1] = boxes[k].min.y; flat[6 * k + 2] = boxes[k].min.z;
        flat[6 * k + 3] = boxes[k].max.x; flat[6 * k + 4] = boxes[k].max.y; flat[6 * k + 5] = boxes[k].max.z;
    }
    BvhTree t;
    t.nodes.resize(2 * boxes.size() + 1);
    t.order.resize(boxes.size());
    uint32_t n_nodes = 0;
    dev.check(rtamd_build_bvh(dev.ctx, flat.data(), (uint32_t)boxes.size(), hit_cost, t.nodes.data(), &n_nodes, t.order.data()), "rtamd_build_bvh");
    t.nodes.resize(n_nodes);
    if (bvh_print_stats_enabled()) print_stats(t);
    return t;
}
void bvh_statistics(const BvhTree& t, uint32_t out[5]) {
    uint32_t n = (uint32_t)t.nodes.size(), max_leaf = 0, min_leaf = (uint32_t)t.order.size(), leafs = 0;
    for (uint32_t i = 0; i + 1 < n; i++) {
        uint32_t size = t.nodes[i + 1].first - t.nodes[i].first;
        if (size > 0) { min_leaf = std::min(min_leaf, size); max_leaf = std::max(max_leaf, size); leafs++; }
    }
    out[0] = (uint32_t)t.order.size(); out[1] = n; out[2] = leafs; out[3] = min_leaf; out[4] = max_leaf;
}

// ---- geometry ----------------------------------------------------------------------------------------
uint32_t FlatScene::image_index(const std::shared_ptr<const Image>& img) {
    for (size_t k = 0; k < image_refs.size(); k++)
        if (image_refs[k].get() == img.get()) return (uint32_t)k;
    image_refs.push_back(img);
    images.push_back(rtamd_image{img->rgb.data(), img->width, img->height});
    return (uint32_t)(images.size() - 1);
}
uint32_t FlatScene::mesh_index(const std::shared_ptr<const Mesh>& m) {
    auto it = mesh_lookup.find(m.get());
    if (it != mesh_lookup.end()) return it->second;
    mesh_lookup.emplace(m.get(), (uint32_t)mesh_refs.size());
    mesh_refs.push_back(m);
    rtamd_mesh d;
    d.nodes = m->tree.nodes.data(); d.n_nodes = (uint32_t)m->tree.nodes.size();
    d.tri_indices = m->tri_indices.data(); d.n_tris = (uint32_t)(m->tri_indices.size() / 3);
    d.vertices = m->vertices.data(); d.normals = m->normals.data(); d.n_verts = (uint32_t)(m->vertices.size() / 3);
    meshes.push_back(d);
    return (uint32_t)(meshes.size() - 1);
}
static void emit_object(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t kind, uint32_t index, uint32_t material) {
    rtamd_object o;
    o.geom_kind = kind; o.geom_index = index; o.material = material;
    o.op_first = (uint32_t)fs.ops.size(); o.op_count = (uint32_t)ops.size();
    fs.ops.insert(fs.ops.end(), ops.begin(), ops.end());
    fs.objects.push_back(o);
}

std::shared_ptr<Mesh> Mesh::from_vertices(const std::vector<Vec3D>& vertices, const std::vector<std::array<uint32_t, 3>>& faces, Device* builder) {
    const size_t n = vertices.size();
    for (const auto& f : faces)
        for (uint32_t i : f)
            if (i >= n) throw std::runtime_error("index " + std::to_string(i) + " is out of bounds");  // mesh.rs:28-34
    std::vector<Vec3D> normals(n, Vec3D::zero());  // mesh.rs:54-78: area-weighted sum, then normalise
    for (const auto& f : faces) {
        Vec3D a = vertices[f[0]], b = vertices[f[1]], c = vertices[f[2]];
        Vec3D nrm = Vec3D::cross(b - a, c - a);
        normals[f[0]] = normals[f[0]] + nrm;
        normals[f[1]] = normals[f[1]] + nrm;
        normals[f[2]] = normals[f[2]] + nrm;
    }
    for (auto& v : normals) v = v.normalize();
    auto mesh = std::make_shared<Mesh>();
    for (size_t k = 0; k < n; k++) {
        mesh->vertices.insert(mesh->vertices.end(), {vertices[k].x, vertices[k].y, vertices[k].z});
        mesh->normals.insert(mesh->normals.end(), {normals[k].x, normals[k].y, normals[k].z});
    }
    std::vector<AABB> boxes(faces.size());  // mesh.rs:149-157
    for (size_t k = 0; k < faces.size(); k++)
        boxes[k] = AABB::from_point(vertices[faces[k][0]]).union_point(vertices[faces[k][1]]).union_point(vertices[faces[k][2]]);
    mesh->tree = builder ? build_aabb_tree(*builder, boxes, 0.1f) : build_aabb_tree(boxes, 0.1f);  // mesh.rs:50
    mesh->face_of_prim = mesh->tree.order;
    for (uint32_t face : mesh->tree.order)
        mesh->tri_indices.insert(mesh->tri_indices.end(), {faces[face][0], faces[face][1], faces[face][2]});
    return mesh;
}
std::shared_ptr<Mesh> Mesh::from_triangles(const std::vector<std::array<Vec3D, 3>>& tris) {  // mesh.rs:83-116
    std::vector<Vec3D> vertices;
    std::vector<std::array<uint32_t, 3>> faces;
    std::map<std::array<uint32_t, 3>, uint32_t> mapping;   // keyed by the bit pattern, as the reference's transmute to [u8; 12]
    for (const auto& tri : tris) {
        std::array<uint32_t, 3> face{};
        for (int i = 0; i < 3; i++) {
            std::array<uint32_t, 3> key;
            const float v[3] = {tri[i].x, tri[i].y, tri[i].z};
            std::memcpy(key.data(), v, 12);
            auto it = mapping.find(key);
            if (it == mapping.end()) {
                it = mapping.emplace(key, (uint32_t)vertices.size()).first;
                vertices.push_back(tri[i]);
            }
            face[i] = it->second;
        }
        faces.push_back(face);
    }
    return from_vertices(vertices, faces);
}
AABB Mesh::bounding_box() const {
    const rtamd_bvh_node& r = tree.nodes[0];
    return AABB(Vec3D(r.min[0], r.min[1], r.min[2]), Vec3D(r.max[0], r.max[1], r.max[2]));
}
void Mesh::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const {
    emit_object(fs, ops, RTAMD_GEOM_MESH, fs.mesh_index(shared_from_this()), material);
}
void UnitSphere::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const { emit_object(fs, ops, RTAMD_GEOM_UNIT_SPHERE, 0, material); }
void UnitCuboid::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const { emit_object(fs, ops, RTAMD_GEOM_UNIT_CUBOID, 0, material); }
Cuboid::Cuboid(Vec3D a, Vec3D b) {
    center = (a + b) * 0.5f;
    Vec3D e = a - center;
    extent = Vec3D(std::fabs(e.x), std::fabs(e.y), std::fabs(e.z));
    inv_extent = Vec3D(1.0f / extent.x, 1.0f / extent.y, 1.0f / extent.z);
}
void Cuboid::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const {
    rtamd_cuboid c;
    c.center[0] = center.x; c.center[1] = center.y; c.center[2] = center.z;
    c.extent[0] = extent.x; c.extent[1] = extent.y; c.extent[2] = extent.z;
    c.inv_extent[0] = inv_extent.x; c.inv_extent[1] = inv_extent.y; c.inv_extent[2] = inv_extent.z;
    fs.cuboids.push_back(c);
    emit_object(fs, ops, RTAMD_GEOM_CUBOID, (uint32_t)fs.cuboids.size() - 1, material);
}
void Triangle::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const {
    rtamd_triangle t;
    t.a[0] = a.x; t.a[1] = a.y; t.a[2] = a.z; t.b[0] = b.x; t.b[1] = b.y; t.b[2] = b.z; t.c[0] = c.x; t.c[1] = c.y; t.c[2] = c.z;
    fs.triangles.push_back(t);
    emit_object(fs, ops, RTAMD_GEOM_TRIANGLE, (uint32_t)fs.triangles.size() - 1, material);
}

AABB Translate::bounding_box() const {
    AABB b = obj->bounding_box();
    return AABB(b.min + offset, b.max + offset);
}
void Translate::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const {
    rtamd_xform_op op{};
    op.kind = RTAMD_OP_TRANSLATE;
    op.v[0] = offset.x; op.v[1] = offset.y; op.v[2] = offset.z;
    ops.push_back(op);
    obj->flatten(fs, ops, material);
    ops.pop_back();
}
Scale::Scale(GeomPtr o, float s) : obj(o), scale(s), inv_scale(1.0f / s) {
    if (!(s > 0.0f)) throw std::runtime_error("scale must be positive");  // transform.rs:87
}
AABB Scale::bounding_box() const {
    AABB b = obj->bounding_box();
    return AABB(b.min * scale, b.max * scale);
}
void Scale::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const {
    rtamd_xform_op op{};
    op.kind = RTAMD_OP_SCALE;
    op.v[0] = scale; op.v[1] = inv_scale;
    ops.push_back(op);
    obj->flatten(fs, ops, material);
    ops.pop_back();
}
AABB Rotate::bounding_box() const {  // transform.rs:201-220
    AABB old = obj->bounding_box();
    Vec3D corners[2] = {old.min, old.max};
    AABB nb;
    for (int i = 0; i < 8; i++) {
        Vec3D p(corners[i % 2].x, corners[(i / 2) % 2].y, corners[(i / 4) % 2].z);
        nb = nb.union_point(mat.apply(p));
    }
    return nb;
}
void Rotate::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const {
    rtamd_xform_op op{};
    op.kind = RTAMD_OP_ROTATE;
    std::memcpy(op.v, mat.m, sizeof(mat.m));
    ops.push_back(op);
    obj->flatten(fs, ops, material);
    ops.pop_back();
}
BoundingBox::BoundingBox(GeomPtr o) : obj(o) {  // aggregate.rs:62-69
    AABB bbox = o->bounding_box();
    Vec3D delta = bbox.max - bbox.min;
    Vec3D pad = Vec3D::fill(delta.x + delta.y * delta.z) * 0.001f;
    box = AABB::from_points(bbox.min - pad, bbox.max + pad);
}
void BoundingBox::flatten(FlatScene& fs, std::vector<rtamd_xform_op>& ops, uint32_t material) const {
    rtamd_xform_op op{};
    op.kind = RTAMD_OP_BOUNDS;
    op.v[0] = box.min.x; op.v[1] = box.min.y; op.v[2] = box.min.z;
    op.v[3] = box.max.x; op.v[4] = box.max.y; op.v[5] = box.max.z;
    ops.push_back(op);
    obj->flatten(fs, ops, material);
    ops.pop_back();
}
std::shared_ptr<Transform> Transform::translate(Vec3D off) const {
    auto t = std::make_shared<Transform>(*this);
    t->offset = offset + off;
    return t;
}
std::shared_ptr<Transform> Transform::scale(float s) const {  // transform.rs:249-253
    auto t = std::make_shared<Transform>(*this);
    t->scale_ = scale_ * s;
    t->offset = offset * s;
    return t;
}
std::shared_ptr<Transform> Transform::rotate(Vec3D axis, float angle) const {  // transform.rs:255-260
    auto t = std::make_shared<Transform>(*this);
    Vec3D off = mat.transpose_apply(offset);
    t->mat = Mat3D::multiply(Mat3D::new_rotation(axis, angle), mat);
    t->offset = t->mat.apply(off);
    return t;
}
std::shared_ptr<Transform> Transform::reflect(Vec3D axis) const {  // transform.rs:274-279, Rotate::reflect :156-160
    auto t = std::make_shared<Transform>(*this);
    Vec3D off = mat.transpose_apply(offset);
    t->mat = Mat3D::multiply(Mat3D::new_reflection(axis), mat);
    t->offset = t->mat.apply(off);
    return t;
}
GeomPtr Transform::chain() const {
    GeomPtr r = std::make_shared<Rotate>(obj, mat);
    GeomPtr s = std::make_shared<Scale>(r, scale_);
    return std::make_shared<Translate>(s, offset);
}
GeomPtr Sphere(Vec3D center, float radius) {
    GeomPtr unit = std::make_shared<UnitSphere>();
    GeomPtr sc = std::make_shared<Scale>(unit, radius);
    return std::make_shared<Translate>(sc, center);
}

std::shared_ptr<ObjectTree> ObjectTree::make(std::vector<Object> objs, float hit_cost) {
    std::vector<AABB> boxes(objs.size());
    for (size_t k = 0; k < objs.size(); k++) boxes[k] = objs[k].bounding_box();
    auto t = std::make_shared<ObjectTree>();
    t->tree = build_aabb_tree(boxes, hit_cost);
    t->objs.reserve(objs.size());
    for (uint32_t src : t->tree.order) t->objs.push_back(objs[src]);
    return t;
}
AABB ObjectTree::bounding_box() const {
    const rtamd_bvh_node& r = tree.nodes[0];
    return AABB(Vec3D(r.min[0], r.min[1], r.min[2]), Vec3D(r.max[0], r.max[1], r.max[2]));
}
GeometryList GeometryList::from_vec(std::vector<Object> v) {
    GeometryList l;
    for (auto& o : v) l.push(o);
    return l;
}
size_t GeometryList::accelerate(size_t min_run) {
    std::vector<Item> out;
    size_t built = 0;
    for (size_t k = 0; k < items.size();) {
        size_t e = k;
        while (e < items.size() && !items[e].is_tree) e++;
        if (e - k >= min_run && e - k >= 2) {
            std::vector<Object> run;
            for (size_t q = k; q < e; q++) run.push_back(items[q].object);
            out.push_back(Item{true, Object(), ObjectTree::make(std::move(run), 1.0f)});
            built++;
        } else {
            for (size_t q = k; q < e; q++) out.push_back(items[q]);
        }
        if (e < items.size()) out.push_back(items[e++]);
        k = e;
    }
    items = std::move(out);
    return built;
}
AABB GeometryList::bounding_box() const {
    AABB b;
    for (const auto& it : items) b = b.union_with(it.is_tree ? it.tree->bounding_box() : it.object.bounding_box());
    return b;
}

// ---- camera ----------------------------------------------------------------------------------------
Camera::Camera(size_t w, size_t h) : width(w), height(h), pos(Vec3D::zero()), dir(Vec3D::z_axis()), horizontal(Vec3D::x_axis()), vertical(Vec3D::y_axis()) {
    look_towards(Vec3D::z_axis(), Vec3D::y_axis());
    perspective(60.0f);
}
Camera& Camera::look_towards(Vec3D d, Vec3D up) {
    d = d.normalize();
    Vec3D horz = Vec3D::cross(up, d).normalize();
    Vec3D vert = Vec3D::cross(d, horz).normalize();
    dir = d;
    horizontal = horz * horizontal.norm();
    vertical = vert * vertical.norm();
    return *this;
}
Camera& Camera::perspective(float fov) {
    float fac = std::tan(to_radians(fov / 2.0f));
    float aspect = (float)height / (float)width;
    horizontal = horizontal * (fac / horizontal.norm());
    vertical = vertical * (fac / vertical.norm() * aspect);
    return *this;
}
rtamd_camera Camera::flat() const {
    rtamd_camera c;
    c.pos[0] = pos.x; c.pos[1] = pos.y; c.pos[2] = pos.z;
    c.dir[0] = dir.x; c.dir[1] = dir.y; c.dir[2] = dir.z;
    c.horizontal[0] = horizontal.x; c.horizontal[1] = horizontal.y; c.horizontal[2] = horizontal.z;
    c.vertical[0] = vertical.x; c.vertical[1] = vertical.y; c.vertical[2] = vertical.z;
    c.width = (uint32_t)width; c.height = (uint32_t)height;
    return c;
}

// ---- device / scene --------------------------------------------------------------------------------
Device::Device(int device_id) {
    int rc = rtamd_create(device_id, &ctx);
    if (rc != RTAMD_OK || !ctx) throw GpuError("rtamd_create failed (code " + std::to_string(rc) + "): no usable GPU / HIP runtime");
}
Device::~Device() { rtamd_destroy(ctx); }
void Device::check(int rc, const char* what) const {
    if (rc != RTAMD_OK) throw GpuError(std::string(what) + ": " + rtamd_last_error(ctx));
}

void Scene::flatten() {
    if (!linear_root) root.accelerate(accelerate_min);
    auto fs = std::make_shared<FlatScene>();
    auto add_material = [&](const Material& m) {
        rtamd_material flat = m.m;
        if (flat.tex_kind == RTAMD_TEX_IMAGE) flat.tex_param = (int32_t)fs->image_index(m.image);
        fs->materials.push_back(flat);
        return (uint32_t)fs->materials.size() - 1;
    };
    std::vector<rtamd_xform_op> ops;
    for (const auto& it : root.items) {
        if (it.is_tree) {
            rtamd_object_tree t;
            t.first_object = (uint32_t)fs->objects.size();
            for (const Object& o : it.tree->objs) {
                size_t before = fs->objects.size();
                o.geometry->flatten(*fs, ops, add_material(o.material));
                if (fs->objects.size() != before + 1) throw std::runtime_error("an object inside an object tree must wrap exactly one primitive");
            }
            t.n_objects = (uint32_t)it.tree->objs.size();
            t.nodes = it.tree->tree.nodes.data();
            t.n_nodes = (uint32_t)it.tree->tree.nodes.size();
            fs->trees.push_back(t);
            fs->root.push_back(rtamd_root_item{RTAMD_ROOT_OBJECT_TREE, (uint32_t)fs->trees.size() - 1});
        } else {
            size_t before = fs->objects.size();
            it.object.geometry->flatten(*fs, ops, add_material(it.object.material));
            if (fs->objects.size() != before + 1) throw std::runtime_error("a root object must wrap exactly one primitive");
            fs->root.push_back(rtamd_root_item{RTAMD_ROOT_OBJECT, (uint32_t)before});
        }
    }
    for (const Light& l : lights) fs->lights.push_back(l.l);
    flat_ = fs;
}
void Scene::upload(Device& dev) {
    flatten();
    upload_flat(dev);
}
void Scene::upload_flat(Device& dev) {
    if (!flat_) flatten();
    const FlatScene* fs = flat_.get();
    rtamd_scene_desc d{};
    d.meshes = fs->meshes.data(); d.n_meshes = (uint32_t)fs->meshes.size();
    d.cuboids = fs->cuboids.data(); d.n_cuboids = (uint32_t)fs->cuboids.size();
    d.triangles = fs->triangles.data(); d.n_triangles = (uint32_t)fs->triangles.size();
    d.ops = fs->ops.data(); d.n_ops = (uint32_t)fs->ops.size();
    d.objects = fs->objects.data(); d.n_objects = (uint32_t)fs->objects.size();
    d.trees = fs->trees.data(); d.n_trees = (uint32_t)fs->trees.size();
    d.root = fs->root.data(); d.n_root = (uint32_t)fs->root.size();
    d.materials = fs->materials.data(); d.n_materials = (uint32_t)fs->materials.size();
    d.lights = fs->lights.data(); d.n_lights = (uint32_t)fs->lights.size();
    if (skybox) { d.skybox_rgb = skybox->rgb.data(); d.sky_w = skybox->width; d.sky_h = skybox->height; }
    d.images = fs->images.data(); d.n_images = (uint32_t)fs->images.size();
    dev.check(rtamd_upload_scene(dev.ctx, &d), "rtamd_upload_scene");
}
void Scene::hit(Device& dev, const rtamd_rays& rays, size_t n, const rtamd_hits& out) const {
    dev.check(rtamd_trace_closest(dev.ctx, &rays, n, &out), "rtamd_trace_closest");
}
void Scene::is_hit(Device& dev, const rtamd_rays& rays, size_t n, uint8_t* out) const {
    dev.check(rtamd_trace_any(dev.ctx, &rays, n, out), "rtamd_trace_any");
}

// ---- render ----------------------------------------------------------------------------------------
namespace render {
std::vector<float> parallel_render(Device& dev, const Scene& scene, const WhittedIntegrator& integ, rtamd_stats* stats, bool quiet) {
    const size_t width = scene.camera.width, height = scene.camera.height;
    std::vector<float> buffer(width * height * 3);
    rtamd_camera cam = scene.camera.flat();
    rtamd_integrator ig = integ.flat();
    rtamd_region region{0, 0, (uint32_t)width, (uint32_t)height, (uint32_t)width, (uint32_t)height, 0, 1};  // one tile = the image
    rtamd_stats local;
    auto before = std::chrono::steady_clock::now();
    dev.check(rtamd_render(dev.ctx, &cam, &ig, &region, buffer.data(), stats ? stats : &local), "rtamd_render");
    double time = std::chrono::duration<double>(std::chrono::steady_clock::now() - before).count();
    if (!quiet) {
        int minutes = (int)std::floor(time / 60.0);
        int seconds = (int)std::ceil(std::fmod(time, 60.0));
        std::printf("Rendered %zux%zu=%zu pixels in %02d:%02d (%.3f sec/pixel)\n", width, height, width * height, minutes, seconds,
                    time / ((double)width * (double)height));  // render.rs:45-53
    }
    return buffer;
}
std::vector<float> parallel_render(const std::vector<Device*>& devs, const Scene& scene, const WhittedIntegrator& integ, rtamd_stats* stats, bool quiet) {
    if (devs.empty()) throw GpuError("parallel_render: no device");
    const size_t width = scene.camera.width, height = scene.camera.height;
    std::vector<float> buffer(width * height * 3);
    rtamd_camera cam = scene.camera.flat();
    rtamd_integrator ig = integ.flat();
    std::vector<rtamd_context*> ctxs;
    for (Device* d : devs) ctxs.push_back(d->ctx);
    auto before = std::chrono::steady_clock::now();
    devs[0]->check(rtamd_render_multi(ctxs.data(), (int)ctxs.size(), &cam, &ig, 0, 0, (uint32_t)width, (uint32_t)height, 64, 64, buffer.data(), stats),
                   "rtamd_render_multi");
    double time = std::chrono::duration<double>(std::chrono::steady_clock::now() - before).count();
    if (!quiet) {
        int minutes = (int)std::floor(time / 60.0);
        int seconds = (int)std::ceil(std::fmod(time, 60.0));
        std::printf("Rendered %zux%zu=%zu pixels in %02d:%02d (%.3f sec/pixel)\n", width, height, width * height, minutes, seconds,
                    time / ((double)width * (double)height));  // render.rs:45-53
    }
    return buffer;
}
std::vector<uint8_t> parallel_render_image(Device& dev, const Scene& scene, const WhittedIntegrator& integ) {
    std::vector<float> buffer = parallel_render(dev, scene, integ);
    std::vector<uint8_t> px(buffer.size());
    for (size_t k = 0; k < buffer.size(); k++) {  // render.rs:62-72
        float f = std::floor(buffer[k] * 256.0f);
        if (!(f == f)) f = 0.0f;
        f = f > 0.0f ? f : 0.0f;
        f = f < 255.0f ? f : 255.0f;
        px[k] = (uint8_t)f;
    }
    return px;
}
void save_ppm(const std::string& path, const std::vector<uint8_t>& rgb, size_t w, size_t h) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    f << "P6\n" << w << " " << h << "\n255\n";
    f.write((const char*)rgb.data(), (std::streamsize)rgb.size());
}
void save_png(const std::string& path, const std::vector<uint8_t>& rgb, size_t w, size_t h) {
    if (rgb.size() != w * h * 3 || w == 0 || h == 0 || w > 0x7FFFFFFFu || h > 0x7FFFFFFFu) throw std::runtime_error("save_png: bad image size");
    static uint32_t table[256];
    static bool have = false;
    if (!have) {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
        have = true;
    }
    auto be32 = [](std::string& o, uint32_t v) { o.push_back((char)(v >> 24)); o.push_back((char)(v >> 16)); o.push_back((char)(v >> 8)); o.push_back((char)v); };
    auto chunk = [&](std::string& file, const char type[4], const std::string& data) {
        be32(file, (uint32_t)data.size());
        std::string body(type, 4);
        body += data;
        uint32_t c = 0xFFFFFFFFu;
        for (unsigned char ch : body) c = table[(c ^ ch) & 0xFFu] ^ (c >> 8);
        file += body;
        be32(file, c ^ 0xFFFFFFFFu);
    };
    // scanlines with filter type 0, wrapped in a zlib stream of stored (uncompressed) deflate blocks
    std::string raw;
    raw.reserve((w * 3 + 1) * h);
    for (size_t y = 0; y < h; y++) {
        raw.push_back('\0');
        raw.append((const char*)rgb.data() + y * w * 3, w * 3);
    }
    std::string z;
    z.push_back((char)0x78); z.push_back((char)0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back((char)(pos + n == raw.size() ? 1 : 0));
        z.push_back((char)(n & 0xFF)); z.push_back((char)(n >> 8));
        z.push_back((char)(~n & 0xFF)); z.push_back((char)((~n >> 8) & 0xFF));
        z.append(raw, pos, n);
        for (size_t k = 0; k < n; k++) { a = (a + (unsigned char)raw[pos + k]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    }
    be32(z, (b << 16) | a);
    std::string file("\x89PNG\r\n\x1a\n", 8);
    std::string ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.push_back((char)8); ihdr.push_back((char)2); ihdr.push_back('\0'); ihdr.push_back('\0'); ihdr.push_back('\0');
    chunk(file, "IHDR", ihdr);
    chunk(file, "IDAT", z);
    chunk(file, "IEND", std::string());
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    f.write(file.data(), (std::streamsize)file.size());
}
}  // namespace render

// ---- PLY loader (loader/ply.rs) -----------------------------------------------------------------------
namespace loader {
namespace {
std::vector<std::string> tokens(const std::string& s) {
    std::vector<std::string> out;
    std::istringstream is(s);
    std::string t;
    while (is >> t) out.push_back(t);
    return out;
}
std::string scalar_type(const std::string& t) {
    if (t == "char" || t == "int8") return "int8";
    if (t == "uchar" || t == "uint8") return "uint8";
    if (t == "short" || t == "int16") return "int16";
    if (t == "ushort" || t == "uint16") return "uint16";
    if (t == "int" || t == "int32") return "int32";
    if (t == "uint" || t == "uint32") return "uint32";
    if (t == "float" || t == "float32") return "float32";
    if (t == "double" || t == "float64") return "float64";
    return "";
}
}  // namespace

namespace {
size_t scalar_size(const std::string& t) {
    if (t == "int8" || t == "uint8") return 1;
    if (t == "int16" || t == "uint16") return 2;
    if (t == "int32" || t == "uint32" || t == "float32") return 4;
    return 8;  // float64
}
// One scalar of a binary PLY body as a double (every PLY scalar type is exactly representable) and, for float32, its bits.
bool read_scalar(std::istream& f, const std::string& type, bool big_endian, double& value, float& as_f32) {
    unsigned char b[8];
    const size_t n = scalar_size(type);
    f.read((char*)b, (std::streamsize)n);
    if (!f) return false;
    if (big_endian) std::reverse(b, b + n);
    as_f32 = 0.0f;
    if (type == "int8") { int8_t v; std::memcpy(&v, b, 1); value = v; }
    else if (type == "uint8") { value = b[0]; }
    else if (type == "int16") { int16_t v; std::memcpy(&v, b, 2); value = v; }
    else if (type == "uint16") { uint16_t v; std::memcpy(&v, b, 2); value = v; }
    else if (type == "int32") { int32_t v; std::memcpy(&v, b, 4); value = v; }
    else if (type == "uint32") { uint32_t v; std::memcpy(&v, b, 4); value = v; }
    else if (type == "float32") { std::memcpy(&as_f32, b, 4); value = as_f32; }
    else { std::memcpy(&value, b, 8); }
    return true;
}
}  // namespace

void load_ply(const std::string& path, std::vector<Vec3D>& vertices, std::vector<std::array<uint32_t, 3>>& faces) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Error while reading file: " + path);
    size_t lineno = 0;
    std::vector<std::string> cur;
    std::string line;
    auto next = [&]() {
        cur.clear();
        if (std::getline(f, line)) { cur = tokens(line); lineno++; }
    };
    auto err = [&](const std::string& m) { return std::runtime_error("Parse error at line " + std::to_string(lineno) + ": " + m); };
    next();
    if (cur != std::vector<std::string>{"ply"}) throw err("expecting 'ply'");
    next();
    // ply.rs:89-91 accepts `format ascii 1.0` only; the two binary forms are this loader's extension
    enum { ASCII, BINARY_LE, BINARY_BE } format = ASCII;
    if (cur == std::vector<std::string>{"format", "binary_little_endian", "1.0"}) format = BINARY_LE;
    else if (cur == std::vector<std::string>{"format", "binary_big_endian", "1.0"}) format = BINARY_BE;
    else if (cur != std::vector<std::string>{"format", "ascii", "1.0"}) throw err("expected 'format ascii 1.0'");
    struct Element { std::string name; size_t count; std::vector<std::pair<std::string, std::string>> props; };
    std::vector<Element> elements;
    next();
    while (!cur.empty()) {
        if (cur[0] == "comment") { next(); continue; }
        if (cur[0] != "element") break;
        if (cur.size() < 3) throw err("failed to parse number");
        Element e;
        e.name = cur[1];
        try { e.count = std::stoull(cur[2]); } catch (...) { throw err("failed to parse number"); }
        for (;;) {
            next();
            if (cur.empty()) break;
            if (cur[0] == "comment") continue;
            if (cur[0] != "property") break;
            std::string type;
            if (cur.size() == 3) type = scalar_type(cur[1]);
            else if (cur.size() == 5 && cur[1] == "list") {
                std::string a = scalar_type(cur[2]), b = scalar_type(cur[3]);
                if (!a.empty() && !b.empty()) type = "list " + a + " " + b;
            }
            if (type.empty()) throw err("invalid type");
            e.props.push_back({cur.back(), type});
        }
        elements.push_back(e);
    }
    if (cur != std::vector<std::string>{"end_header"}) throw err("expected 'end_header'");
    if (format != ASCII) {
        // The body follows the header byte for byte: elements in header order, properties in declaration order.
        const bool be = format == BINARY_BE;
        for (const Element& e : elements) {
            const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
            int xi = -1, yi = -1, zi = -1, index = -1;
            for (size_t k = 0; k < e.props.size(); k++) {
                const auto& pr = e.props[k];
                if (is_vertex && pr.second == "float32") {
                    if (pr.first == "x") xi = (int)k;
                    if (pr.first == "y") yi = (int)k;
                    if (pr.first == "z") zi = (int)k;
                }
                if (is_face && index < 0 && (pr.first == "vertex_indices" || pr.first == "vertex_index") && pr.second.rfind("list ", 0) == 0 &&
                    pr.second.find("float") == std::string::npos)
                    index = (int)k;
            }
            if (is_vertex && xi < 0) throw err("vertex has no x property");
            if (is_vertex && yi < 0) throw err("vertex has no y property");
            if (is_vertex && zi < 0) throw err("vertex has no z property");
            if (is_face && index < 0) throw err("face has not vertex_indices property");
            if (!is_vertex && !is_face) std::fprintf(stderr, "WARN: ignoring element \"%s\"\n", e.name.c_str());
            std::vector<uint32_t> idx;
            for (size_t k = 0; k < e.count; k++) {
                float xyz[3] = {0.f, 0.f, 0.f};
                for (size_t q = 0; q < e.props.size(); q++) {
                    const std::string& type = e.props[q].second;
                    double v;
                    float fv;
                    if (type.rfind("list ", 0) != 0) {
                        if (!read_scalar(f, type, be, v, fv)) throw err("unexpected end of the binary body");
                        if ((int)q == xi) xyz[0] = fv;
                        if ((int)q == yi) xyz[1] = fv;
                        if ((int)q == zi) xyz[2] = fv;
                        continue;
                    }
                    const size_t sp = type.find(' ', 5);
                    const std::string count_type = type.substr(5, sp - 5), item_type = type.substr(sp + 1);
                    // (a list's length is an integer type; a float one could hold a NaN or a value no size_t can)
                    if (count_type.rfind("float", 0) == 0) throw err("list length of a floating-point type");
                    if (!read_scalar(f, count_type, be, v, fv)) throw err("unexpected end of the binary body");
                    if (!(v >= 0.0 && v <= 16777216.0)) throw err("invalid list length");
                    const size_t n = (size_t)v;
                    const bool take = is_face && (int)q == index;
                    if (take && n < 3) throw err("invalid number of vertices");
                    idx.clear();
                    for (size_t w = 0; w < n; w++) {
                        if (!read_scalar(f, item_type, be, v, fv)) throw err("unexpected end of the binary body");
                        if (!take) continue;
                        if (v < 0 || v >= (double)vertices.size())
                            throw err("invalid vertex identifier " + std::to_string((long long)v) + ", only " + std::to_string(vertices.size()) + " vertices");
                        idx.push_back((uint32_t)v);
                    }
                    if (take)
                        for (size_t w = 2; w < idx.size(); w++) faces.push_back({idx[0], idx[w - 1], idx[w]});  // fan, ply.rs:246-248
                }
                if (is_vertex) vertices.push_back(Vec3D(xyz[0], xyz[1], xyz[2]));
            }
        }
        return;
    }
    for (const Element& e : elements) {
        if (e.name == "vertex") {
            int xi = -1, yi = -1, zi = -1;
            for (size_t k = 0; k < e.props.size(); k++) {
                if (e.props[k].second != "float32") continue;
                if (e.props[k].first == "x") xi = (int)k;
                if (e.props[k].first == "y") yi = (int)k;
                if (e.props[k].first == "z") zi = (int)k;
            }
            if (xi < 0) throw err("vertex has no x property");
            if (yi < 0) throw err("vertex has no y property");
            if (zi < 0) throw err("vertex has no z property");
            for (size_t k = 0; k < e.count; k++) {
                next();
                if (cur.size() != e.props.size()) throw err("expected " + std::to_string(e.props.size()) + " elements");
                char* end;
                float x = std::strtof(cur[xi].c_str(), &end); if (*end) throw err("failed to parse x coordinate");
                float y = std::strtof(cur[yi].c_str(), &end); if (*end) throw err("failed to parse y coordinate");
                float z = std::strtof(cur[zi].c_str(), &end); if (*end) throw err("failed to parse z coordinate");
                vertices.push_back(Vec3D(x, y, z));
            }
        } else if (e.name == "face") {
            int index = -1;
            for (size_t k = 0; k < e.props.size(); k++)
                if ((e.props[k].first == "vertex_indices" || e.props[k].first == "vertex_index") && e.props[k].second == "list uint8 int32") { index = (int)k; break; }
            if (index < 0) throw err("face has not vertex_indices property");
            std::vector<uint32_t> idx;
            for (size_t k = 0; k < e.count; k++) {
                next();
                if ((size_t)index >= cur.size()) throw err("failed to parse integer");
                size_t n;
                try { n = std::stoull(cur[index]); } catch (...) { throw err("failed to parse integer"); }
                if (n < 3 || index + n >= cur.size()) throw err("invalid number of vertices");
                idx.clear();
                for (size_t q = index + 1; q < index + 1 + n; q++) {
                    unsigned long long v;
                    try { v = std::stoull(cur[q]); } catch (...) { throw err("failed to parse integer"); }
                    if (v >= vertices.size()) throw err("invalid vertex identifier " + cur[q] + ", only " + std::to_string(vertices.size()) + " vertices");
                    idx.push_back((uint32_t)v);
                }
                for (size_t q = 2; q < idx.size(); q++) faces.push_back({idx[0], idx[q - 1], idx[q]});  // fan, ply.rs:246-248
            }
        } else {
            std::fprintf(stderr, "WARN: ignoring element \"%s\"\n", e.name.c_str());
        }
    }
}
std::shared_ptr<Mesh> load_ply_as_mesh(const std::string& path) {
    std::vector<Vec3D> v;
    std::vector<std::array<uint32_t, 3>> f;
    load_ply(path, v, f);
    return Mesh::from_vertices(v, f);
}
}  // namespace loader

}  // namespace rth
