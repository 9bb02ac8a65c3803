// TEST INFRASTRUCTURE -- C entry points of the CPU oracle (ctypes-friendly).
// Scene scripts restate src/main.rs:22-97 of the reference (asset paths supplied by the caller)
// and add the two synthetic scenes used by the larger benchmark configurations.
#include <cstring>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <functional>
#include <thread>

#include "rt_oracle.hpp"

using namespace orc;

namespace {

struct OrcScene {
    Scene scene;
    WhittedIntegrator integ;
    std::shared_ptr<Mesh> mesh;                        // the (single) shared mesh, if any
    std::shared_ptr<Mesh> mesh2;                       // a second mesh some synthetic scenes hold
    std::shared_ptr<AABBTree<Object>> object_tree;     // object tree, if any
    std::vector<const Object*> objects;                // flattened object order (global ids)
    V3 cam_pos, cam_focus, cam_up;                     // camera script arguments (main.rs:58-63)
    std::string error;
    void make_camera(size_t w, size_t h) {
        Camera cam(w, h);
        cam.position(cam_pos).look_at(cam_focus, cam_up);
        scene.camera = cam;
    }
};

const float PI_F = 3.14159265358979323846f;

std::shared_ptr<Image> make_sky(const uint8_t* rgb, int w, int h) {
    auto img = std::make_shared<Image>();
    if (rgb && w > 0 && h > 0) {
        img->w = w; img->h = h;
        img->rgb.assign(rgb, rgb + (size_t)w * h * 3);
    } else {  // 1x1 grey sky
        img->w = 1; img->h = 1;
        img->rgb = {128, 128, 128};
    }
    return img;
}

std::vector<Light> main_lights() {  // main.rs:67-76
    std::vector<Light> lights;
    Light amb; amb.kind = LIGHT_AMBIENT; amb.emission = V3::fill(1.0f) * 0.1f;
    Light dl; dl.kind = LIGHT_DIRECTION; dl.vec = normalize(V3(-2.5f, -3.0f, -3.0f)); dl.param = 0.01f;
    dl.emission = V3(1.f, 1.f, 1.f) * 0.4f;
    Light ao; ao.kind = LIGHT_AO; ao.param = std::fabs(1e12f); ao.emission = V3::fill(1.0f) * 0.5f;
    lights.push_back(amb); lights.push_back(dl); lights.push_back(ao);
    return lights;
}

Material mat_transparent(float ior) { Material m; m.kind = MAT_TRANSPARENT; m.p0 = ior; return m; }
Material mat_metal() { Material m; m.kind = MAT_METAL; return m; }
Texture tex_color(Color c) { Texture t; t.color = c; return t; }
Texture tex_uv() { Texture t; t.kind = TEX_UV; return t; }
Texture tex_checkerboard(int repeats) { Texture t; t.kind = TEX_CHECKERBOARD; t.repeats = repeats; return t; }
Texture tex_image(std::shared_ptr<const Image> img) { Texture t; t.kind = TEX_IMAGE; t.image = img; return t; }
Material mat_lambert(Texture t) { Material m; m.kind = MAT_LAMBERT; m.tex = t; return m; }
Material mat_lambert(Color c) { return mat_lambert(tex_color(c)); }
Material mat_glossy(float e, float k, Texture t) { Material m; m.kind = MAT_GLOSSY; m.p0 = e; m.p1 = k; m.tex = t; return m; }
Material mat_glossy(float e, float k, Color c) { return mat_glossy(e, k, tex_color(c)); }

// splitmix64: the generator of the synthetic scenes (shared definition with the product's host code:
// state += 0x9E3779B97F4A7C15; z = (z ^ z>>30) * 0xBF58476D1CE4E5B9; z = (z ^ z>>27) * 0x94D049BB133111EB; z ^ z>>31;
// uniform float = (z >> 40) * 2^-24).
struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    float uniform() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
};

void finalize_object_ids(OrcScene* s, const std::vector<std::shared_ptr<Object>>& root_objects) {
    // Global object ids: object-tree leaves first (in leaf order), then the plain root objects.
    int32_t id = 0;
    if (s->object_tree) {
        for (Object& o : s->object_tree->objs) { o.index = id++; s->objects.push_back(&o); }
    }
    for (const auto& o : root_objects) { o->index = id++; s->objects.push_back(o.get()); }
}

}  // namespace

extern "C" {

// ---- scenes -------------------------------------------------------------------------------------
// The reference's hard-coded scene (main.rs:22-83) with the asset paths supplied by the caller.
void* orc_scene_main(const char* ply_path, const uint8_t* sky_rgb, int sky_w, int sky_h, int cam_w, int cam_h) {
    auto* s = new OrcScene();
    std::vector<V3> verts;
    std::vector<std::array<uint32_t, 3>> faces;
    if (!load_ply(ply_path, verts, faces, s->error)) { fprintf(stderr, "orc: %s\n", s->error.c_str()); delete s; return nullptr; }
    s->mesh = Mesh::from_vertices(verts, faces);
    GeomPtr tmesh = TransformBuilder(s->mesh).scale(20.0f).rotate(V3(1.f, 0.f, 0.f), 90.0f * (PI_F / 180.0f)).build();
    AABB bb = tmesh->bounding_box();
    GeomPtr cube = std::make_shared<Cuboid>(V3(-5.0f, -5.0f, bb.min[2]), V3(5.0f, 5.0f, bb.min[2] - 1.0f));
    V3 c = (bb.min + bb.max) / 2.0f;

    std::vector<std::shared_ptr<Object>> objs;
    objs.push_back(std::make_shared<Object>(tmesh, mat_transparent(1.0f / 0.6f)));
    objs.push_back(std::make_shared<Object>(std::make_shared<Translate>(tmesh, V3::fill(0.0f) + V3(0.0f, -2.5f, 0.0f)), mat_metal()));
    objs.push_back(std::make_shared<Object>(std::make_shared<Translate>(tmesh, V3::fill(0.0f) + V3(0.0f, 2.5f, 0.0f)), mat_lambert(V3(1.f, 1.f, 1.f))));
    objs.push_back(std::make_shared<Object>(cube, mat_glossy(1e3f, 0.3f, V3(1.f, 1.f, 1.f))));
    auto list = std::make_shared<GeometryList>();
    for (auto& o : objs) list->items.push_back(o);
    finalize_object_ids(s, objs);

    V3 pos(-6.0f, -6.0f, 3.0f);
    V3 up = -V3(0.f, 0.f, 1.f);
    V3 focus = c + V3(0.0f, 0.0f, 0.1f);
    s->cam_pos = pos; s->cam_focus = focus; s->cam_up = up;
    s->make_camera((size_t)cam_w, (size_t)cam_h);

    s->scene.root = list;
    s->scene.skybox = make_sky(sky_rgb, sky_w, sky_h);
    s->scene.lights = main_lights();
    return s;
}

// Synthetic: n spheres (each its own Object, alternating Lambertian / Metal) in one AABBTree<Object>
// (hit_cost 1.0) + a floor cuboid.  Centres uniform in [-12,12]^2 x [0.3,4.0], radii in [0.08,0.25].
// `scale` multiplies every length of the scene in f32 after it is drawn (1.0: every value as drawn).
void* orc_scene_spheres_scaled(int n, uint64_t seed, float scale, const uint8_t* sky_rgb, int sky_w, int sky_h, int cam_w, int cam_h) {
    auto* s = new OrcScene();
    SplitMix64 g{seed};
    std::vector<Object> objs;
    objs.reserve(n);
    for (int i = 0; i < n; i++) {
        float cx = -12.0f + 24.0f * g.uniform();
        float cy = -12.0f + 24.0f * g.uniform();
        float cz = 0.3f + 3.7f * g.uniform();
        float r = 0.08f + 0.17f * g.uniform();
        objs.emplace_back(make_sphere(V3(cx * scale, cy * scale, cz * scale), r * scale), (i % 2 == 0) ? mat_lambert(V3(1.f, 1.f, 1.f)) : mat_metal());
    }
    s->object_tree = std::make_shared<AABBTree<Object>>(std::move(objs), 1.0f);
    std::vector<std::shared_ptr<Object>> root_objs;
    root_objs.push_back(std::make_shared<Object>(std::make_shared<Cuboid>(V3(-15.0f * scale, -15.0f * scale, 0.0f), V3(15.0f * scale, 15.0f * scale, -1.0f * scale)),
                                                 mat_glossy(1e3f, 0.3f, V3(1.f, 1.f, 1.f))));
    auto list = std::make_shared<GeometryList>();
    list->items.push_back(s->object_tree);
    for (auto& o : root_objs) list->items.push_back(o);
    finalize_object_ids(s, root_objs);
    s->cam_pos = V3(-18.0f * scale, -18.0f * scale, 9.0f * scale); s->cam_focus = V3(0.0f, 0.0f, 1.0f * scale); s->cam_up = -V3(0.f, 0.f, 1.f);
    s->make_camera((size_t)cam_w, (size_t)cam_h);
    s->scene.root = list;
    s->scene.skybox = make_sky(sky_rgb, sky_w, sky_h);
    s->scene.lights = main_lights();
    return s;
}
void* orc_scene_spheres(int n, uint64_t seed, const uint8_t* sky_rgb, int sky_w, int sky_h, int cam_w, int cam_h) {
    return orc_scene_spheres_scaled(n, seed, 1.0f, sky_rgb, sky_w, sky_h, cam_w, cam_h);
}

// Synthetic: every texture and the materials main.rs leaves out, in a plain GeometryList (same construction as the
// product's host-side scene of that name): UV-textured mesh, checkerboard sphere, glossy image-textured sphere (tex,
// or the skybox image), Glass sphere, standalone checkerboard triangle, checkerboard floor.
void* orc_scene_textured(const char* ply_path, const uint8_t* sky_rgb, int sky_w, int sky_h, const uint8_t* tex_rgb, int tex_w, int tex_h, int cam_w,
                         int cam_h) {
    auto* s = new OrcScene();
    std::vector<V3> verts;
    std::vector<std::array<uint32_t, 3>> faces;
    if (!load_ply(ply_path, verts, faces, s->error)) { fprintf(stderr, "orc: %s\n", s->error.c_str()); delete s; return nullptr; }
    s->mesh = Mesh::from_vertices(verts, faces);
    GeomPtr tmesh = TransformBuilder(s->mesh).scale(20.0f).rotate(V3(1.f, 0.f, 0.f), 90.0f * (PI_F / 180.0f)).build();
    AABB bb = tmesh->bounding_box();
    V3 c = (bb.min + bb.max) / 2.0f;
    auto sky = make_sky(sky_rgb, sky_w, sky_h);
    std::shared_ptr<const Image> timg = (tex_rgb && tex_w > 0 && tex_h > 0) ? make_sky(tex_rgb, tex_w, tex_h) : sky;
    const bool have_image = (tex_rgb && tex_w > 0 && tex_h > 0) || (sky_rgb && sky_w > 0 && sky_h > 0);
    std::vector<std::shared_ptr<Object>> objs;
    objs.push_back(std::make_shared<Object>(tmesh, mat_lambert(tex_uv())));
    objs.push_back(std::make_shared<Object>(make_sphere(V3(c[0], c[1] - 2.6f, bb.min[2] + 0.9f), 0.9f), mat_lambert(tex_checkerboard(8))));
    objs.push_back(std::make_shared<Object>(make_sphere(V3(c[0], c[1] + 2.6f, bb.min[2] + 0.9f), 0.9f),
                                            mat_glossy(20.0f, 0.4f, have_image ? tex_image(timg) : tex_checkerboard(6))));
    objs.push_back(std::make_shared<Object>(make_sphere(V3(c[0] - 2.2f, c[1] - 0.6f, bb.min[2] + 0.6f), 0.6f), mat_transparent(1.5f)));  // Glass
    objs.push_back(std::make_shared<Object>(std::make_shared<Triangle>(V3(c[0] + 1.5f, c[1] - 3.0f, bb.min[2] + 0.01f), V3(c[0] + 1.5f, c[1] + 3.0f, bb.min[2] + 0.01f),
                                                                       V3(c[0] + 1.5f, c[1], bb.min[2] + 3.0f)),
                                            mat_lambert(tex_checkerboard(4))));
    objs.push_back(std::make_shared<Object>(std::make_shared<Cuboid>(V3(-5.0f, -5.0f, bb.min[2]), V3(5.0f, 5.0f, bb.min[2] - 1.0f)),
                                            mat_glossy(1e3f, 0.3f, tex_checkerboard(10))));
    // a BoundingBox prefilter around a translated sphere, and one around a flat (zero-thickness in z) triangle, whose
    // padded box is still flat in z only if dx + dy * dz is 0: here it is not, the triangle stays visible
    objs.push_back(std::make_shared<Object>(std::make_shared<BoundingBox>(make_sphere(V3(c[0] - 1.0f, c[1] + 3.6f, bb.min[2] + 0.5f), 0.5f)), mat_metal()));
    objs.push_back(std::make_shared<Object>(std::make_shared<Translate>(
                                                std::make_shared<BoundingBox>(std::make_shared<Triangle>(V3(0.f, 0.f, 0.f), V3(1.2f, 0.f, 0.f), V3(0.f, 1.2f, 0.f))),
                                                V3(c[0] - 3.0f, c[1] - 3.0f, bb.min[2] + 0.02f)),
                                            mat_lambert(tex_uv())));
    {   // Mesh::from_triangles: a tetrahedron given as four independent triangles (4 shared vertices)
        V3 p0(0.f, 0.f, 0.f), p1(1.f, 0.f, 0.f), p2(0.f, 1.f, 0.f), p3(0.3f, 0.3f, 1.1f);
        std::vector<std::array<V3, 3>> tris = {{p0, p2, p1}, {p0, p1, p3}, {p1, p2, p3}, {p2, p0, p3}};
        s->mesh2 = Mesh::from_triangles(tris);
        objs.push_back(std::make_shared<Object>(std::make_shared<Translate>(s->mesh2, V3(c[0] - 2.8f, c[1] + 1.6f, bb.min[2])), mat_lambert(tex_checkerboard(3))));
    }
    // Transform::reflect + a rotation about a general axis (mat3d.rs:47-80 as written, slips included), then moved beside the floor's far corner
    objs.push_back(std::make_shared<Object>(TransformBuilder(s->mesh).scale(9.0f).rotate(V3(1.f, 0.f, 0.f), 90.0f * (PI_F / 180.0f)).reflect(V3(0.f, 1.f, 0.f))
                                                .rotate(V3(0.3f, 0.2f, 1.0f), 0.7f).translate(V3(c[0] + 2.6f, c[1] - 3.4f, bb.min[2] + 0.1f)).build(),
                                            mat_lambert(tex_checkerboard(5))));
    auto list = std::make_shared<GeometryList>();
    for (auto& o : objs) list->items.push_back(o);
    finalize_object_ids(s, objs);
    s->cam_pos = V3(-6.0f, -6.0f, 3.0f); s->cam_focus = c + V3(0.0f, 0.0f, 0.1f); s->cam_up = -V3(0.f, 0.f, 1.f);
    s->make_camera((size_t)cam_w, (size_t)cam_h);
    s->scene.root = list;
    s->scene.skybox = sky;
    s->scene.lights = main_lights();
    return s;
}

// Synthetic: nx*ny instances of the (scaled, rotated) mesh on a grid of pitch 4.0, materials cycling
// Transparent / Metal / Lambertian by (ix+iy)%3, in one AABBTree<Object> (hit_cost 1.0) + floor.
// flat != 0: every instance is a mesh of its own whose vertices are carried to world space in f32 the way a hit point
// leaves the wrappers (rotate, scale, the Transform's own offset, the instance offset: transform.rs:183,108,61), with its
// own normals and tree (Mesh::from_vertices) and no ray transform.
void* orc_scene_grid2(const char* ply_path, int nx, int ny, int flat, const uint8_t* sky_rgb, int sky_w, int sky_h, int cam_w, int cam_h) {
    auto* s = new OrcScene();
    std::vector<V3> verts;
    std::vector<std::array<uint32_t, 3>> faces;
    if (!load_ply(ply_path, verts, faces, s->error)) { fprintf(stderr, "orc: %s\n", s->error.c_str()); delete s; return nullptr; }
    s->mesh = Mesh::from_vertices(verts, faces);
    TransformBuilder tb = TransformBuilder(s->mesh).scale(20.0f).rotate(V3(1.f, 0.f, 0.f), 90.0f * (PI_F / 180.0f));
    GeomPtr tmesh = tb.build();
    AABB bb = tmesh->bounding_box();
    const float pitch = 4.0f;
    float x0 = -0.5f * pitch * (float)(nx - 1), y0 = -0.5f * pitch * (float)(ny - 1);
    std::vector<Object> objs;
    objs.reserve((size_t)nx * ny);
    for (int iy = 0; iy < ny; iy++)
        for (int ix = 0; ix < nx; ix++) {
            V3 off(x0 + pitch * (float)ix, y0 + pitch * (float)iy, 0.0f);
            Material m;
            switch ((ix + iy) % 3) {
                case 0: m = mat_transparent(1.0f / 0.6f); break;
                case 1: m = mat_metal(); break;
                default: m = mat_lambert(V3(1.f, 1.f, 1.f)); break;
            }
            objs.emplace_back(flat ? GeomPtr() : GeomPtr(std::make_shared<Translate>(tmesh, V3::fill(0.0f) + off)), m);
        }
    if (flat >= 2) {   // the instances of each material merged into one world-space mesh, in a plain root list with the floor
        std::vector<V3> mv[3];
        std::vector<std::array<uint32_t, 3>> mf[3];
        for (size_t k = 0; k < objs.size(); k++) {
            const int ix = (int)(k % (size_t)nx), iy = (int)(k / (size_t)nx), m = (ix + iy) % 3;
            const V3 off = V3::fill(0.0f) + V3(x0 + pitch * (float)ix, y0 + pitch * (float)iy, 0.0f);
            const uint32_t base = (uint32_t)mv[m].size();
            for (size_t v = 0; v < verts.size(); v++) mv[m].push_back(((tb.mat.apply(verts[v]) * tb.scale_) + tb.offset) + off);
            for (const auto& f : faces) mf[m].push_back({base + f[0], base + f[1], base + f[2]});
        }
        float hx = 0.5f * pitch * (float)nx + 2.0f, hy = 0.5f * pitch * (float)ny + 2.0f;
        std::vector<std::shared_ptr<Object>> root_objs;
        const Material mats[3] = {mat_transparent(1.0f / 0.6f), mat_metal(), mat_lambert(V3(1.f, 1.f, 1.f))};
        for (int m = 0; m < 3; m++)
            if (!mv[m].empty()) {
                auto mesh = Mesh::from_vertices(mv[m], mf[m]);
                if (root_objs.empty()) s->mesh = mesh;
                root_objs.push_back(std::make_shared<Object>(mesh, mats[m]));
            }
        root_objs.push_back(std::make_shared<Object>(std::make_shared<Cuboid>(V3(-hx, -hy, bb.min[2]), V3(hx, hy, bb.min[2] - 1.0f)),
                                                     mat_glossy(1e3f, 0.3f, V3(1.f, 1.f, 1.f))));
        auto list = std::make_shared<GeometryList>();
        for (auto& o : root_objs) list->items.push_back(o);
        finalize_object_ids(s, root_objs);
        float ext = hx > hy ? hx : hy;
        s->cam_pos = V3(-1.2f * ext, -1.2f * ext, 0.9f * ext); s->cam_focus = V3(0.0f, 0.0f, 0.0f); s->cam_up = -V3(0.f, 0.f, 1.f);
        s->make_camera((size_t)cam_w, (size_t)cam_h);
        s->scene.root = list;
        s->scene.skybox = make_sky(sky_rgb, sky_w, sky_h);
        s->scene.lights = main_lights();
        return s;
    }
    if (flat) {
        std::atomic<size_t> next{0};
        auto work = [&]() {
            std::vector<V3> wv(verts.size());
            for (size_t k = next++; k < objs.size(); k = next++) {
                const int ix = (int)(k % (size_t)nx), iy = (int)(k / (size_t)nx);
                const V3 off = V3::fill(0.0f) + V3(x0 + pitch * (float)ix, y0 + pitch * (float)iy, 0.0f);
                for (size_t v = 0; v < verts.size(); v++) wv[v] = ((tb.mat.apply(verts[v]) * tb.scale_) + tb.offset) + off;
                objs[k].geometry = Mesh::from_vertices(wv, faces);
            }
        };
        size_t nt = std::min<size_t>(16, std::max<size_t>(1, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    s->object_tree = std::make_shared<AABBTree<Object>>(std::move(objs), 1.0f);
    float hx = 0.5f * pitch * (float)nx + 2.0f, hy = 0.5f * pitch * (float)ny + 2.0f;
    std::vector<std::shared_ptr<Object>> root_objs;
    root_objs.push_back(std::make_shared<Object>(std::make_shared<Cuboid>(V3(-hx, -hy, bb.min[2]), V3(hx, hy, bb.min[2] - 1.0f)),
                                                 mat_glossy(1e3f, 0.3f, V3(1.f, 1.f, 1.f))));
    auto list = std::make_shared<GeometryList>();
    list->items.push_back(s->object_tree);
    for (auto& o : root_objs) list->items.push_back(o);
    finalize_object_ids(s, root_objs);
    float ext = hx > hy ? hx : hy;
    s->cam_pos = V3(-1.2f * ext, -1.2f * ext, 0.9f * ext); s->cam_focus = V3(0.0f, 0.0f, 0.0f); s->cam_up = -V3(0.f, 0.f, 1.f);
    s->make_camera((size_t)cam_w, (size_t)cam_h);
    s->scene.root = list;
    s->scene.skybox = make_sky(sky_rgb, sky_w, sky_h);
    s->scene.lights = main_lights();
    return s;
}

void* orc_scene_grid(const char* ply_path, int nx, int ny, const uint8_t* sky_rgb, int sky_w, int sky_h, int cam_w, int cam_h) {
    return orc_scene_grid2(ply_path, nx, ny, 0, sky_rgb, sky_w, sky_h, cam_w, cam_h);
}

// Synthetic: n small spheres and cuboids (alternating, splitmix64(seed): centre in [-6,6]^2 x [0.2,2.5], size in
// [0.1,0.4], drawn cx, cy, cz, size) in a PLAIN GeometryList followed by a floor cuboid.  accelerate != 0 applies the
// product's top-level extension: the run of plain objects becomes one AABBTree<Object> (hit_cost 1.0) in place.
void* orc_scene_scatter(int n, uint64_t seed, int accelerate, const uint8_t* sky_rgb, int sky_w, int sky_h, int cam_w, int cam_h) {
    auto* s = new OrcScene();
    SplitMix64 g{seed};
    std::vector<Object> run;
    for (int i = 0; i < n; i++) {
        float cx = -6.0f + 12.0f * g.uniform();
        float cy = -6.0f + 12.0f * g.uniform();
        float cz = 0.2f + 2.3f * g.uniform();
        float r = 0.1f + 0.3f * g.uniform();
        GeomPtr geom = (i % 2 == 0) ? make_sphere(V3(cx, cy, cz), r) : GeomPtr(std::make_shared<Cuboid>(V3(cx - r, cy - r, cz - r), V3(cx + r, cy + r, cz + r)));
        Material m = (i % 3 == 0) ? mat_metal() : ((i % 3 == 1) ? mat_lambert(V3(1.f, 1.f, 1.f)) : mat_lambert(tex_checkerboard(4)));
        run.emplace_back(geom, m);
    }
    auto floor = std::make_shared<Object>(std::make_shared<Cuboid>(V3(-8.0f, -8.0f, 0.0f), V3(8.0f, 8.0f, -1.0f)), mat_glossy(1e3f, 0.3f, V3(1.f, 1.f, 1.f)));
    auto list = std::make_shared<GeometryList>();
    std::vector<std::shared_ptr<Object>> root_objs;
    if (accelerate && n >= 2) {
        s->object_tree = std::make_shared<AABBTree<Object>>(std::move(run), 1.0f);
        list->items.push_back(s->object_tree);
    } else {
        for (auto& o : run) root_objs.push_back(std::make_shared<Object>(o));
    }
    root_objs.push_back(floor);
    for (auto& o : root_objs) list->items.push_back(o);
    finalize_object_ids(s, root_objs);
    s->cam_pos = V3(-9.0f, -9.0f, 6.0f); s->cam_focus = V3(0.0f, 0.0f, 0.8f); s->cam_up = -V3(0.f, 0.f, 1.f);
    s->make_camera((size_t)cam_w, (size_t)cam_h);
    s->scene.root = list;
    s->scene.skybox = make_sky(sky_rgb, sky_w, sky_h);
    s->scene.lights = main_lights();
    return s;
}

// Synthetic: a heightfield mesh of n x n cells over [-6,6]^2 (vertex heights 0.6 * uniform from splitmix64(seed), drawn
// row by row; two triangles per cell; in every cell with (ix + iy) % 7 == 0 the first triangle is listed twice, so
// that leaves hold coplanar duplicates), Lambertian, translated to z = -1; with_bunny != 0 adds the reference scene's
// scaled and rotated mesh (Transparent) and a copy at y + 2.5 (Metal) in front of it; a glossy floor cuboid below.
// The heightfield comes first in the root list, i.e. it is mesh 0 of the flattened scene.
void* orc_scene_terrain(const char* ply_path, int n, uint64_t seed, int with_bunny, const uint8_t* sky_rgb, int sky_w, int sky_h,
                        int cam_w, int cam_h) {
    auto* s = new OrcScene();
    SplitMix64 g{seed};
    std::vector<V3> tv;
    std::vector<std::array<uint32_t, 3>> tf;
    const float cell = 12.0f / (float)n;
    for (int iy = 0; iy <= n; iy++)
        for (int ix = 0; ix <= n; ix++) tv.push_back(V3(-6.0f + cell * (float)ix, -6.0f + cell * (float)iy, 0.6f * g.uniform()));
    for (int iy = 0; iy < n; iy++)
        for (int ix = 0; ix < n; ix++) {
            uint32_t a = (uint32_t)(iy * (n + 1) + ix), b = a + 1, c = a + (uint32_t)(n + 1), d = c + 1;
            tf.push_back({a, b, d});
            if ((ix + iy) % 7 == 0) tf.push_back({a, b, d});
            tf.push_back({a, d, c});
        }
    auto terrain = Mesh::from_vertices(tv, tf);
    s->mesh = terrain;
    std::vector<std::shared_ptr<Object>> objs;
    objs.push_back(std::make_shared<Object>(std::make_shared<Translate>(terrain, V3::fill(0.0f) + V3(0.0f, 0.0f, -1.0f)), mat_lambert(V3(1.f, 1.f, 1.f))));
    if (with_bunny) {
        std::vector<V3> verts;
        std::vector<std::array<uint32_t, 3>> faces;
        if (!load_ply(ply_path, verts, faces, s->error)) { fprintf(stderr, "orc: %s\n", s->error.c_str()); delete s; return nullptr; }
        auto bunny = Mesh::from_vertices(verts, faces);
        GeomPtr tmesh = TransformBuilder(bunny).scale(20.0f).rotate(V3(1.f, 0.f, 0.f), 90.0f * (PI_F / 180.0f)).build();
        objs.push_back(std::make_shared<Object>(tmesh, mat_transparent(1.0f / 0.6f)));
        objs.push_back(std::make_shared<Object>(std::make_shared<Translate>(tmesh, V3::fill(0.0f) + V3(0.0f, 2.5f, 0.0f)), mat_metal()));
    }
    objs.push_back(std::make_shared<Object>(std::make_shared<Cuboid>(V3(-7.0f, -7.0f, -1.5f), V3(7.0f, 7.0f, -2.5f)), mat_glossy(1e3f, 0.3f, V3(1.f, 1.f, 1.f))));
    auto list = std::make_shared<GeometryList>();
    for (auto& o : objs) list->items.push_back(o);
    finalize_object_ids(s, objs);
    s->cam_pos = V3(-8.0f, -8.0f, 5.0f); s->cam_focus = V3(0.0f, 0.0f, 0.0f); s->cam_up = -V3(0.f, 0.f, 1.f);
    s->make_camera((size_t)cam_w, (size_t)cam_h);
    s->scene.root = list;
    s->scene.skybox = make_sky(sky_rgb, sky_w, sky_h);
    s->scene.lights = main_lights();
    // two point lights (light.rs:34-79): a ball of radius 0.5 and a true point
    Light p1; p1.kind = LIGHT_POINT; p1.vec = V3(2.0f, -3.0f, 4.0f); p1.param = std::fabs(0.5f); p1.emission = V3(1.0f, 0.9f, 0.8f) * 12.0f;
    Light p2; p2.kind = LIGHT_POINT; p2.vec = V3(-4.0f, 3.0f, 2.5f); p2.param = std::fabs(0.0f); p2.emission = V3(0.6f, 0.7f, 1.0f) * 6.0f;
    s->scene.lights.push_back(p1); s->scene.lights.push_back(p2);
    return s;
}

void orc_scene_free(void* p) { delete (OrcScene*)p; }

// Re-run the camera script (Camera::new(w,h).position(..).look_at(..)) at another resolution.
void orc_scene_set_camera_size(void* p, int w, int h) { ((OrcScene*)p)->make_camera((size_t)w, (size_t)h); }

// camera record: pos(3) dir(3) horizontal(3) vertical(3) width height
void orc_scene_camera(void* p, float out[12], uint32_t wh[2]) {
    OrcScene* s = (OrcScene*)p;
    const Camera& c = s->scene.camera;
    for (int i = 0; i < 3; i++) { out[i] = c.pos[i]; out[3 + i] = c.dir[i]; out[6 + i] = c.horizontal[i]; out[9 + i] = c.vertical[i]; }
    wh[0] = (uint32_t)c.width; wh[1] = (uint32_t)c.height;
}

// ---- structure dumps ----------------------------------------------------------------------------
// which: 0 = mesh triangle tree, 1 = object tree.  nodes as 8 x u32 words (6 float bit patterns, skip, first).
int orc_bvh_size(void* p, int which, uint32_t out[2]) {
    OrcScene* s = (OrcScene*)p;
    if (which == 0 && s->mesh) { out[0] = (uint32_t)s->mesh->tree->nodes.size(); out[1] = (uint32_t)s->mesh->tree->objs.size(); return 0; }
    if (which == 1 && s->object_tree) { out[0] = (uint32_t)s->object_tree->nodes.size(); out[1] = (uint32_t)s->object_tree->objs.size(); return 0; }
    return -1;
}
int orc_bvh_dump(void* p, int which, uint32_t* nodes8, uint32_t* leaf_ids, uint32_t stats[5]) {
    OrcScene* s = (OrcScene*)p;
    const std::vector<BvhNode>* nodes = nullptr;
    if (which == 0 && s->mesh) {
        nodes = &s->mesh->tree->nodes;
        for (size_t k = 0; k < s->mesh->tree->objs.size(); k++) leaf_ids[k] = s->mesh->tree->objs[k].face;
        s->mesh->tree->stats(stats);
    } else if (which == 1 && s->object_tree) {
        nodes = &s->object_tree->nodes;
        for (size_t k = 0; k < s->object_tree->objs.size(); k++) leaf_ids[k] = (uint32_t)k;
        s->object_tree->stats(stats);
    } else {
        return -1;
    }
    for (size_t k = 0; k < nodes->size(); k++) {
        const BvhNode& nd = (*nodes)[k];
        float f[6] = {nd.bb.min[0], nd.bb.min[1], nd.bb.min[2], nd.bb.max[0], nd.bb.max[1], nd.bb.max[2]};
        std::memcpy(&nodes8[k * 8], f, 24);
        nodes8[k * 8 + 6] = nd.skip;
        nodes8[k * 8 + 7] = nd.first;
    }
    return 0;
}
// mesh data: vertices then normals (3 floats each); returns vertex count
int orc_mesh_data(void* p, float* out, int capacity_floats) {
    OrcScene* s = (OrcScene*)p;
    if (!s->mesh) return -1;
    const auto& d = s->mesh->data->data;
    if (out && (size_t)capacity_floats >= d.size() * 3)
        for (size_t k = 0; k < d.size(); k++) { out[3 * k] = d[k][0]; out[3 * k + 1] = d[k][1]; out[3 * k + 2] = d[k][2]; }
    return (int)(d.size() / 2);
}

// ---- ray batches ---------------------------------------------------------------------------------
// rays: SoA ox,oy,oz,dx,dy,dz,tmax (7 arrays of n floats).
static void run_parallel(size_t n, int threads, const std::function<void(size_t, size_t)>& fn, Counters* total) {
    if (threads < 1) threads = 1;
    std::atomic<size_t> next(0);
    const size_t chunk = 4096;
    std::vector<Counters> per(threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
            g_cnt = Counters();
            for (;;) {
                size_t b = next.fetch_add(chunk);
                if (b >= n) break;
                size_t e = b + chunk < n ? b + chunk : n;
                fn(b, e);
            }
            per[t] = g_cnt;
        });
    }
    for (auto& th : pool) th.join();
    if (total) for (auto& c : per) total->add(c);
}
static void export_counters(const Counters& c, uint64_t out[12]) {
    if (!out) return;
    out[0] = c.nodes_visited; out[1] = c.tris_tested; out[2] = c.spheres_tested; out[3] = c.cuboids_tested;
    out[4] = c.closest_rays; out[5] = c.any_rays; out[6] = c.degenerate_tris; out[7] = 0;
    out[8] = c.nodes_visited_any; out[9] = c.tris_tested_any; out[10] = c.spheres_tested_any; out[11] = c.cuboids_tested_any;
}

// Scene::root.hit(ray, t_max) per ray (integrator.rs:64).  prim = 0xFFFFFFFF and obj = -1 on a miss.
int orc_trace_closest(void* p, size_t n, const float* const rays[7], int32_t* obj, uint32_t* prim, float* t,
                      float* u, float* v, float* t_local, float* pos3, float* norm3, int threads, uint64_t counters[12]) {
    OrcScene* s = (OrcScene*)p;
    Counters total;
    run_parallel(n, threads, [&](size_t b, size_t e) {
        for (size_t k = b; k < e; k++) {
            Ray ray(V3(rays[0][k], rays[1][k], rays[2][k]), V3(rays[3][k], rays[4][k], rays[5][k]));
            HitResult h;
            g_cnt.closest_rays++;
            bool ok = s->scene.root->hit(ray, rays[6][k], h);
            obj[k] = ok ? h.obj : -1;
            prim[k] = ok ? h.prim : 0xFFFFFFFFu;
            t[k] = ok ? h.t : 0.0f;
            u[k] = ok ? h.uv[0] : 0.0f;
            v[k] = ok ? h.uv[1] : 0.0f;
            if (t_local) t_local[k] = ok ? h.t_local : 0.0f;
            if (pos3) for (int c = 0; c < 3; c++) pos3[3 * k + c] = ok ? h.pos[c] : 0.0f;
            if (norm3) for (int c = 0; c < 3; c++) norm3[3 * k + c] = ok ? h.norm[c] : 0.0f;
        }
    }, &total);
    export_counters(total, counters);
    return 0;
}
// Scene::root.is_hit(ray, t_max) per ray (integrator.rs:124).
int orc_trace_any(void* p, size_t n, const float* const rays[7], uint8_t* out, int threads, uint64_t counters[12]) {
    OrcScene* s = (OrcScene*)p;
    Counters total;
    run_parallel(n, threads, [&](size_t b, size_t e) {
        for (size_t k = b; k < e; k++) {
            Ray ray(V3(rays[0][k], rays[1][k], rays[2][k]), V3(rays[3][k], rays[4][k], rays[5][k]));
            g_cnt.any_rays++;
            g_cnt.in_any = true;
            out[k] = s->scene.root->is_hit(ray, rays[6][k]) ? 1 : 0;
            g_cnt.in_any = false;
        }
    }, &total);
    export_counters(total, counters);
    return 0;
}
// Camera::generate_ray for pixel centres of the rect (x0,y0,w,h), row-major (scene.rs:84-90).
int orc_primary_rays(void* p, int x0, int y0, int w, int h, float* const rays[7]) {
    OrcScene* s = (OrcScene*)p;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t k = (size_t)y * w + x;
            Ray r = s->scene.camera.generate_ray((float)(x0 + x), (float)(y0 + y));
            rays[0][k] = r.pos[0]; rays[1][k] = r.pos[1]; rays[2][k] = r.pos[2];
            rays[3][k] = r.dir[0]; rays[4][k] = r.dir[1]; rays[5][k] = r.dir[2];
            rays[6][k] = 1e12f;
        }
    return 0;
}

// ---- render --------------------------------------------------------------------------------------
// parallel_render (render.rs:10-56) restricted to the rect: worker threads pull 100-pixel chunks from
// a shared counter; params = {max_depth, shadow_rays, scatter_rays, antialiasing}.  out is row-major
// w*h*3 floats.  Returns the wall time of the pixel loop in seconds.
double orc_render(void* p, const int params[4], int x0, int y0, int w, int h, float* out, int threads, uint64_t counters[12]) {
    OrcScene* s = (OrcScene*)p;
    WhittedIntegrator integ;
    integ.max_depth = params[0]; integ.shadow_rays = params[1]; integ.scatter_rays = params[2]; integ.antialiasing = params[3];
    if (threads < 1) threads = 1;
    size_t n = (size_t)w * h;
    std::atomic<size_t> next(0);
    std::vector<Counters> per(threads);
    std::vector<std::thread> pool;
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
            g_cnt = Counters();
            for (;;) {
                size_t b = next.fetch_add(100);
                if (b >= n) break;
                size_t e = b + 100 < n ? b + 100 : n;
                for (size_t idx = b; idx < e; idx++) {
                    size_t x = idx % (size_t)w, y = idx / (size_t)w;
                    Color c = integ.calculate_pixel(s->scene, (size_t)x0 + x, (size_t)y0 + y);
                    out[3 * idx] = c[0]; out[3 * idx + 1] = c[1]; out[3 * idx + 2] = c[2];
                }
            }
            per[t] = g_cnt;
        });
    }
    for (auto& th : pool) th.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Counters total;
    for (auto& c : per) total.add(c);
    export_counters(total, counters);
    return sec;
}
// Diagnostic: the closest-hit rays of one pixel in the order the recursion traces them (see g_path_log): up to max_records
// records of 11 floats; returns the number of records the pixel produced.  rgb receives the pixel.
int orc_pixel_path(void* p, const int params[4], int cx, int cy, float* records, int max_records, float rgb[3]) {
    OrcScene* s = (OrcScene*)p;
    WhittedIntegrator integ;
    integ.max_depth = params[0]; integ.shadow_rays = params[1]; integ.scatter_rays = params[2]; integ.antialiasing = params[3];
    std::vector<float> log;
    g_path_log = &log;
    const Color c = integ.calculate_pixel(s->scene, (size_t)cx, (size_t)cy);
    g_path_log = nullptr;
    rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
    const int n = (int)(log.size() / 11);
    std::memcpy(records, log.data(), sizeof(float) * 11 * (size_t)(n < max_records ? n : max_records));
    return n;
}
// render.rs:62-72 quantiser
void orc_quantize(const float* rgb, size_t n, uint8_t* out) {
    for (size_t k = 0; k < n; k++) {
        float f = std::floor(rgb[k] * 256.0f);
        f = (f != f) ? 0.0f : f;
        f = f > 0.0f ? f : 0.0f;   // .max(0.0)
        f = f < 255.0f ? f : 255.0f;  // .min(255.0)
        out[k] = (uint8_t)f;
    }
}

// ---- small known-answer probes ---------------------------------------------------------------------
void orc_kat_moller_trumbore(const float tri[9], const float pos[3], const float dir[3], float out[3]) {
    Ray r(V3(pos[0], pos[1], pos[2]), V3(dir[0], dir[1], dir[2]));
    moller_trumbore(V3(tri[0], tri[1], tri[2]), V3(tri[3], tri[4], tri[5]), V3(tri[6], tri[7], tri[8]), r, out);
}
int orc_kat_triangle_hit(const float tri[9], const float pos[3], const float dir[3], float t_max, float out_tuv[3]) {
    Ray r(V3(pos[0], pos[1], pos[2]), V3(dir[0], dir[1], dir[2]));
    Triangle tr(V3(tri[0], tri[1], tri[2]), V3(tri[3], tri[4], tri[5]), V3(tri[6], tri[7], tri[8]));
    HitResult h;
    if (!tr.hit(r, t_max, h)) return 0;
    out_tuv[0] = h.t; out_tuv[1] = h.uv[0]; out_tuv[2] = h.uv[1];
    return 1;
}
int orc_kat_aabb(const float mn[3], const float mx[3], const float pos[3], const float dir[3], float out[2]) {
    AABB b(V3(mn[0], mn[1], mn[2]), V3(mx[0], mx[1], mx[2]));
    V3 inv = 1.0f / V3(dir[0], dir[1], dir[2]);
    bool neg[3] = {std::signbit(inv[0]), std::signbit(inv[1]), std::signbit(inv[2])};
    float t0, t1;
    bool ok = b.fast_intersect_ray(V3(pos[0], pos[1], pos[2]), inv, neg, t0, t1);
    out[0] = t0; out[1] = t1;
    return ok ? 1 : 0;
}
// unit sphere optionally wrapped in Scale(scale) and Translate(center); returns hit flag, out = t, pos(3), any-hit flag
int orc_kat_sphere(const float center[3], float radius, const float pos[3], const float dir[3], float t_max, float out[5]) {
    GeomPtr g = make_sphere(V3(center[0], center[1], center[2]), radius);
    Ray r(V3(pos[0], pos[1], pos[2]), V3(dir[0], dir[1], dir[2]));
    HitResult h;
    bool ok = g->hit(r, t_max, h);
    out[0] = ok ? h.t : 0.0f;
    for (int c = 0; c < 3; c++) out[1 + c] = ok ? h.pos[c] : 0.0f;
    out[4] = g->is_hit(r, t_max) ? 1.0f : 0.0f;
    return ok ? 1 : 0;
}
int orc_kat_cuboid(const float a[3], const float b[3], const float pos[3], const float dir[3], float t_max, float out[8]) {
    Cuboid c(V3(a[0], a[1], a[2]), V3(b[0], b[1], b[2]));
    Ray r(V3(pos[0], pos[1], pos[2]), V3(dir[0], dir[1], dir[2]));
    HitResult h;
    bool ok = c.hit(r, t_max, h);
    out[0] = ok ? h.t : 0.0f;
    for (int k = 0; k < 3; k++) { out[1 + k] = ok ? h.pos[k] : 0.0f; out[4 + k] = ok ? h.norm[k] : 0.0f; }
    out[7] = c.is_hit(r, t_max) ? 1.0f : 0.0f;
    return ok ? 1 : 0;
}
// Two-triangle mesh probe: builds Mesh::from_vertices and returns the leaf-order prim index hit.
int orc_kat_mesh_hit(const float* verts, int n_verts, const uint32_t* faces, int n_faces, const float pos[3],
                     const float dir[3], float t_max, uint32_t* prim_leaf, uint32_t* face, float tuv[3]) {
    std::vector<V3> v;
    std::vector<std::array<uint32_t, 3>> f;
    for (int k = 0; k < n_verts; k++) v.push_back(V3(verts[3 * k], verts[3 * k + 1], verts[3 * k + 2]));
    for (int k = 0; k < n_faces; k++) f.push_back({faces[3 * k], faces[3 * k + 1], faces[3 * k + 2]});
    auto mesh = Mesh::from_vertices(v, f);
    Ray r(V3(pos[0], pos[1], pos[2]), V3(dir[0], dir[1], dir[2]));
    HitResult h;
    if (!mesh->hit(r, t_max, h)) return 0;
    *prim_leaf = h.prim;
    *face = mesh->tree->objs[h.prim].face;
    tuv[0] = h.t; tuv[1] = h.uv[0]; tuv[2] = h.uv[1];
    return 1;
}
void orc_kat_rng(uint64_t seed, int n, uint32_t* out_u32, float* out_f32) {
    SmallRng a = SmallRng::seed_from_u64(seed);
    SmallRng b = a;
    for (int k = 0; k < n; k++) { out_u32[k] = a.next_u32(); out_f32[k] = b.gen_f32(); }
}
int orc_kat_ply(const char* path, uint32_t counts[2], uint32_t* first_faces, int n_first) {
    std::vector<V3> v;
    std::vector<std::array<uint32_t, 3>> f;
    std::string err;
    if (!load_ply(path, v, f, err)) return -1;
    counts[0] = (uint32_t)v.size(); counts[1] = (uint32_t)f.size();
    for (int k = 0; k < n_first && (size_t)k < f.size(); k++) { first_faces[3 * k] = f[k][0]; first_faces[3 * k + 1] = f[k][1]; first_faces[3 * k + 2] = f[k][2]; }
    return 0;
}
void orc_kat_rotation(const float axis[3], float angle, float out9[9]) {
    Mat3 m = Mat3::new_rotation(V3(axis[0], axis[1], axis[2]), angle);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out9[3 * i + j] = m.m[i][j];
}
// Material::sample_at of Lambartian(texture) at (u, v): kind 0 colour (white), 1 UV, 2 checkerboard(param), 3 image (rgb, w x h).
void orc_kat_texture(int kind, int param, const uint8_t* rgb, int w, int h, float u, float v, float out[3]) {
    Texture t;
    if (kind == TEX_UV) t = tex_uv();
    else if (kind == TEX_CHECKERBOARD) t = tex_checkerboard(param);
    else if (kind == TEX_IMAGE) t = tex_image(make_sky(rgb, w, h));
    Color c = mat_lambert(t).sample_at(u, v);
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
}
void orc_kat_sky(void* p, const float dir[3], float out[3]) {
    OrcScene* s = (OrcScene*)p;
    Ray r(V3(0, 0, 0), V3(dir[0], dir[1], dir[2]));
    Color c = s->scene.calculate_background(r);
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
}

}  // extern "C"
