// Scene scripts: the reference's hard-coded scene (src/main.rs:22-83) and the two synthetic scenes
// used for the larger benchmark configurations.
#include "rt_host.hpp"

#include <algorithm>
#include <atomic>
#include <thread>

namespace rth {

static std::vector<Light> main_lights() {  // main.rs:67-76
    return {AmbientLight(Vec3D::one(), 0.1f),
            DirectionLight(vec3d(-2.5f, -3.0f, -3.0f), 0.01f, COLOR_WHITE, 0.4f),
            AmbientOcclusion(1e12f, Vec3D::one(), 0.5f)};
}

Scene scene_main(const std::string& ply_path, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h) {
    std::vector<Object> list;
    auto mesh = loader::load_ply_as_mesh(ply_path);
    GeomPtr tmesh = Transform::make(mesh)->scale(20.0f)->rotate_x(to_radians(90.0f));  // main.rs:25-27

    AABB bb = tmesh->bounding_box();
    GeomPtr cube = Cuboid::make(vec3d(-5.0f, -5.0f, bb.min.z), vec3d(5.0f, 5.0f, bb.min.z - 1.0f));
    Vec3D c = (bb.min + bb.max) / 2.0f;

    list.push_back(Object::with_material(tmesh, Transparent(1.0f / 0.6f)));
    list.push_back(Object::with_material(Translate::make(tmesh)->translate_y(-2.5f), Metal()));
    list.push_back(Object::with_material(Translate::make(tmesh)->translate_y(2.5f), Lambartian(COLOR_WHITE)));
    list.push_back(Object::with_material(cube, Glossy(1e3f, 0.3f, COLOR_WHITE)));

    Vec3D pos(-6.0f, -6.0f, 3.0f);
    Vec3D up = -Vec3D::z_axis();
    Vec3D focus = c + Vec3D(0.0f, 0.0f, 0.1f);
    Camera cam(cam_w, cam_h);
    cam.position(pos).look_at(focus, up);

    Scene scene;
    scene.root = GeometryList::from_vec(list);
    scene.camera = cam;
    scene.skybox = skybox;
    scene.lights = main_lights();
    return scene;
}

// splitmix64 -> uniform float in [0,1): (z >> 40) * 2^-24.
namespace {
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
}  // namespace

// n spheres, each its own Object (even index: white Lambertian, odd: Metal), in one AABBTree<Object>
// with hit_cost 1.0, above a glossy floor cuboid.  Centres uniform in [-12,12]^2 x [0.3,4.0], radii
// uniform in [0.08,0.25]; per sphere the generator is drawn in the order cx, cy, cz, r.  `scale` multiplies every length of
// the scene (centres, radii, floor, camera position and focus) in f32 after it is drawn: 1.0 leaves every value as it is.
Scene scene_spheres(int n, uint64_t seed, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h, float scale) {
    SplitMix64 g{seed};
    std::vector<Object> objs;
    objs.reserve((size_t)n);
    for (int i = 0; i < n; i++) {
        float cx = -12.0f + 24.0f * g.uniform();
        float cy = -12.0f + 24.0f * g.uniform();
        float cz = 0.3f + 3.7f * g.uniform();
        float r = 0.08f + 0.17f * g.uniform();
        objs.push_back(Object::with_material(Sphere(Vec3D(cx * scale, cy * scale, cz * scale), r * scale), (i % 2 == 0) ? Lambartian(COLOR_WHITE) : Metal()));
    }
    Scene scene;
    scene.root.push(ObjectTree::make(std::move(objs), 1.0f));
    scene.root.push(Object::with_material(Cuboid::make(vec3d(-15.0f * scale, -15.0f * scale, 0.0f), vec3d(15.0f * scale, 15.0f * scale, -1.0f * scale)),
                                          Glossy(1e3f, 0.3f, COLOR_WHITE)));
    Camera cam(cam_w, cam_h);
    cam.position(Vec3D(-18.0f * scale, -18.0f * scale, 9.0f * scale)).look_at(Vec3D(0.0f, 0.0f, 1.0f * scale), -Vec3D::z_axis());
    scene.camera = cam;
    scene.skybox = skybox;
    scene.lights = main_lights();
    return scene;
}

// nx * ny instances of the scaled + rotated mesh on a grid of pitch 4.0 (row-major, x fastest),
// materials cycling Transparent / Metal / Lambertian by (ix + iy) % 3, in one AABBTree<Object> with
// hit_cost 1.0, above a glossy floor cuboid.
// flat: the world-space form of the same scene.  Every instance is a mesh of its own whose vertices are carried to world
// space in f32 the way a hit point leaves the wrappers (rotate, scale, the Transform's own offset, the instance offset:
// transform.rs:183,108,61), with its own normals and tree (Mesh::from_vertices) and no ray transform: nx * ny * 948
// world-space triangles in nx * ny trees (230 x 230: 50.1 M triangles, 2.4 GB of triangle records).
// flat 2: the instances of each material merged into one world-space mesh (three meshes of nx * ny * 948 / 3 triangles,
// vertices of instance k at [453 k', 453 k' + 453) in instance order) in a plain root list with the floor: no object
// tree, no instancing; the three trees are built by `builder` (the GPU builder) when one is given.
Scene scene_grid(const std::string& ply_path, int nx, int ny, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h, int flat, Device* builder) {
    std::vector<Vec3D> verts;
    std::vector<std::array<uint32_t, 3>> faces;
    loader::load_ply(ply_path, verts, faces);
    auto mesh = Mesh::from_vertices(verts, faces);
    auto tform = Transform::make(mesh)->scale(20.0f)->rotate_x(to_radians(90.0f));
    GeomPtr tmesh = tform;
    AABB bb = tmesh->bounding_box();
    const float pitch = 4.0f;
    float x0 = -0.5f * pitch * (float)(nx - 1), y0 = -0.5f * pitch * (float)(ny - 1);
    std::vector<Object> objs;
    objs.reserve((size_t)nx * ny);
    for (int iy = 0; iy < ny; iy++)
        for (int ix = 0; ix < nx; ix++) {
            Vec3D off(x0 + pitch * (float)ix, y0 + pitch * (float)iy, 0.0f);
            Material m = ((ix + iy) % 3 == 0) ? Transparent(1.0f / 0.6f) : (((ix + iy) % 3 == 1) ? Metal() : Lambartian(COLOR_WHITE));
            objs.push_back(Object::with_material(flat ? GeomPtr() : GeomPtr(Translate::make(tmesh)->translate(off)), m));
        }
    if (flat == 2) {
        std::vector<Vec3D> mv[3];
        std::vector<std::array<uint32_t, 3>> mf[3];
        for (size_t k = 0; k < objs.size(); k++) {
            const int ix = (int)(k % (size_t)nx), iy = (int)(k / (size_t)nx), m = (ix + iy) % 3;
            const Vec3D off = Vec3D::zero() + Vec3D(x0 + pitch * (float)ix, y0 + pitch * (float)iy, 0.0f);
            const uint32_t base = (uint32_t)mv[m].size();
            for (size_t v = 0; v < verts.size(); v++) mv[m].push_back(((tform->mat.apply(verts[v]) * tform->scale_) + tform->offset) + off);
            for (const auto& f : faces) mf[m].push_back({base + f[0], base + f[1], base + f[2]});
        }
        float hx = 0.5f * pitch * (float)nx + 2.0f, hy = 0.5f * pitch * (float)ny + 2.0f;
        Scene scene;
        const Material mats[3] = {Transparent(1.0f / 0.6f), Metal(), Lambartian(COLOR_WHITE)};
        for (int m = 0; m < 3; m++)
            if (!mv[m].empty()) scene.root.push(Object::with_material(Mesh::from_vertices(mv[m], mf[m], builder), mats[m]));
        scene.root.push(Object::with_material(Cuboid::make(vec3d(-hx, -hy, bb.min.z), vec3d(hx, hy, bb.min.z - 1.0f)), Glossy(1e3f, 0.3f, COLOR_WHITE)));
        float ext = hx > hy ? hx : hy;
        Camera cam(cam_w, cam_h);
        cam.position(Vec3D(-1.2f * ext, -1.2f * ext, 0.9f * ext)).look_at(Vec3D(0.0f, 0.0f, 0.0f), -Vec3D::z_axis());
        scene.camera = cam;
        scene.skybox = skybox;
        scene.lights = main_lights();
        return scene;
    }
    if (flat) {   // the instances' meshes, built on up to 16 host threads
        std::atomic<size_t> next{0};
        auto work = [&]() {
            std::vector<Vec3D> wv(verts.size());
            for (size_t k = next++; k < objs.size(); k = next++) {
                const int ix = (int)(k % (size_t)nx), iy = (int)(k / (size_t)nx);
                const Vec3D off = Vec3D::zero() + Vec3D(x0 + pitch * (float)ix, y0 + pitch * (float)iy, 0.0f);
                for (size_t v = 0; v < verts.size(); v++) wv[v] = ((tform->mat.apply(verts[v]) * tform->scale_) + tform->offset) + off;
                objs[k].geometry = Mesh::from_vertices(wv, faces);
            }
        };
        const size_t nt = std::min<size_t>(16, std::max<size_t>(1, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    float hx = 0.5f * pitch * (float)nx + 2.0f, hy = 0.5f * pitch * (float)ny + 2.0f;
    Scene scene;
    scene.root.push(ObjectTree::make(std::move(objs), 1.0f));
    scene.root.push(Object::with_material(Cuboid::make(vec3d(-hx, -hy, bb.min.z), vec3d(hx, hy, bb.min.z - 1.0f)), Glossy(1e3f, 0.3f, COLOR_WHITE)));
    float ext = hx > hy ? hx : hy;
    Camera cam(cam_w, cam_h);
    cam.position(Vec3D(-1.2f * ext, -1.2f * ext, 0.9f * ext)).look_at(Vec3D(0.0f, 0.0f, 0.0f), -Vec3D::z_axis());
    scene.camera = cam;
    scene.skybox = skybox;
    scene.lights = main_lights();
    return scene;
}

// Heightfield of n x n cells over [-6,6]^2 (vertex heights 0.6 * uniform, drawn row by row; two triangles per cell;
// every cell with (ix + iy) % 7 == 0 lists its first triangle twice: coplanar duplicates inside a leaf), Lambertian,
// translated to z = -1; with_bunny adds the reference scene's mesh (Transparent) and a copy at y + 2.5 (Metal);
// glossy floor cuboid below.  A scene with two different meshes, one of them as large as n asks for.
Scene scene_terrain(const std::string& ply_path, int n, uint64_t seed, bool with_bunny, std::shared_ptr<const Image> skybox,
                    size_t cam_w, size_t cam_h) {
    SplitMix64 g{seed};
    std::vector<Vec3D> tv;
    std::vector<std::array<uint32_t, 3>> tf;
    const float cell = 12.0f / (float)n;
    for (int iy = 0; iy <= n; iy++)
        for (int ix = 0; ix <= n; ix++) tv.push_back(Vec3D(-6.0f + cell * (float)ix, -6.0f + cell * (float)iy, 0.6f * g.uniform()));
    for (int iy = 0; iy < n; iy++)
        for (int ix = 0; ix < n; ix++) {
            uint32_t a = (uint32_t)(iy * (n + 1) + ix), b = a + 1, c = a + (uint32_t)(n + 1), d = c + 1;
            tf.push_back({a, b, d});
            if ((ix + iy) % 7 == 0) tf.push_back({a, b, d});
            tf.push_back({a, d, c});
        }
    auto terrain = Mesh::from_vertices(tv, tf);
    std::vector<Object> list;
    list.push_back(Object::with_material(Translate::make(terrain)->translate(Vec3D(0.0f, 0.0f, -1.0f)), Lambartian(COLOR_WHITE)));
    if (with_bunny) {
        auto mesh = loader::load_ply_as_mesh(ply_path);
        GeomPtr tmesh = Transform::make(mesh)->scale(20.0f)->rotate_x(to_radians(90.0f));
        list.push_back(Object::with_material(tmesh, Transparent(1.0f / 0.6f)));
        list.push_back(Object::with_material(Translate::make(tmesh)->translate_y(2.5f), Metal()));
    }
    list.push_back(Object::with_material(Cuboid::make(vec3d(-7.0f, -7.0f, -1.5f), vec3d(7.0f, 7.0f, -2.5f)), Glossy(1e3f, 0.3f, COLOR_WHITE)));
    Camera cam(cam_w, cam_h);
    cam.position(Vec3D(-8.0f, -8.0f, 5.0f)).look_at(Vec3D(0.0f, 0.0f, 0.0f), -Vec3D::z_axis());
    Scene scene;
    scene.root = GeometryList::from_vec(list);
    scene.camera = cam;
    scene.skybox = skybox;
    scene.lights = main_lights();
    // two point lights (light.rs:34-79): a ball of radius 0.5 and a true point
    scene.lights.push_back(PointLight(Vec3D(2.0f, -3.0f, 4.0f), 0.5f, Vec3D(1.0f, 0.9f, 0.8f), 12.0f));
    scene.lights.push_back(PointLight(Vec3D(-4.0f, 3.0f, 2.5f), 0.0f, Vec3D(0.6f, 0.7f, 1.0f), 6.0f));
    return scene;
}

// Every texture and the materials main.rs does not use, in a plain GeometryList: the scaled + rotated mesh with a
// UV texture (barycentric u, v), a checkerboard sphere, a glossy sphere textured with `tex` (the skybox image when the
// caller has none), a Glass sphere, a standalone triangle with a coarse checkerboard and a checkerboard floor.
Scene scene_textured(const std::string& ply_path, std::shared_ptr<const Image> skybox, std::shared_ptr<const Image> tex, size_t cam_w, size_t cam_h) {
    std::vector<Object> list;
    auto mesh = loader::load_ply_as_mesh(ply_path);
    GeomPtr tmesh = Transform::make(mesh)->scale(20.0f)->rotate_x(to_radians(90.0f));
    AABB bb = tmesh->bounding_box();
    Vec3D c = (bb.min + bb.max) / 2.0f;
    list.push_back(Object::with_material(tmesh, Lambartian(UVTexture())));
    list.push_back(Object::with_material(Sphere(Vec3D(c.x, c.y - 2.6f, bb.min.z + 0.9f), 0.9f), Lambartian(Checkerboard(8))));
    list.push_back(Object::with_material(Sphere(Vec3D(c.x, c.y + 2.6f, bb.min.z + 0.9f), 0.9f), Glossy(20.0f, 0.4f, (tex || skybox) ? Texture(tex ? tex : skybox) : Checkerboard(6))));
    list.push_back(Object::with_material(Sphere(Vec3D(c.x - 2.2f, c.y - 0.6f, bb.min.z + 0.6f), 0.6f), Glass()));
    list.push_back(Object::with_material(std::make_shared<Triangle>(Vec3D(c.x + 1.5f, c.y - 3.0f, bb.min.z + 0.01f), Vec3D(c.x + 1.5f, c.y + 3.0f, bb.min.z + 0.01f),
                                                                    Vec3D(c.x + 1.5f, c.y, bb.min.z + 3.0f)),
                                         Lambartian(Checkerboard(4))));
    list.push_back(Object::with_material(Cuboid::make(vec3d(-5.0f, -5.0f, bb.min.z), vec3d(5.0f, 5.0f, bb.min.z - 1.0f)), Glossy(1e3f, 0.3f, Checkerboard(10))));
    // BoundingBox prefilters (around a sphere, and inside a Translate around a flat triangle) and Mesh::from_triangles
    list.push_back(Object::with_material(BoundingBox::make(Sphere(Vec3D(c.x - 1.0f, c.y + 3.6f, bb.min.z + 0.5f), 0.5f)), Metal()));
    list.push_back(Object::with_material(
        std::make_shared<Translate>(BoundingBox::make(std::make_shared<Triangle>(Vec3D(0.f, 0.f, 0.f), Vec3D(1.2f, 0.f, 0.f), Vec3D(0.f, 1.2f, 0.f))),
                                    Vec3D(c.x - 3.0f, c.y - 3.0f, bb.min.z + 0.02f)),
        Lambartian(UVTexture())));
    {
        Vec3D p0(0.f, 0.f, 0.f), p1(1.f, 0.f, 0.f), p2(0.f, 1.f, 0.f), p3(0.3f, 0.3f, 1.1f);
        auto tetra = Mesh::from_triangles({{p0, p2, p1}, {p0, p1, p3}, {p1, p2, p3}, {p2, p0, p3}});
        list.push_back(Object::with_material(std::make_shared<Translate>(tetra, Vec3D(c.x - 2.8f, c.y + 1.6f, bb.min.z)), Lambartian(Checkerboard(3))));
    }
    // Transform::reflect + a rotation about a general axis (mat3d.rs:47-80 as written, slips included), beside the floor's far corner
    list.push_back(Object::with_material(Transform::make(mesh)->scale(9.0f)->rotate_x(to_radians(90.0f))->reflect_y()->rotate(Vec3D(0.3f, 0.2f, 1.0f), 0.7f)
                                             ->translate(Vec3D(c.x + 2.6f, c.y - 3.4f, bb.min.z + 0.1f)),
                                         Lambartian(Checkerboard(5))));
    Scene scene;
    for (auto& o : list) scene.root.push(o);
    Camera cam(cam_w, cam_h);
    cam.position(Vec3D(-6.0f, -6.0f, 3.0f)).look_at(c + Vec3D(0.0f, 0.0f, 0.1f), -Vec3D::z_axis());
    scene.camera = cam;
    scene.skybox = skybox;
    scene.lights = main_lights();
    return scene;
}

// n small spheres and cuboids (alternating, splitmix64(seed): centre in [-6,6]^2 x [0.2,2.5], size in [0.1,0.4], drawn
// cx, cy, cz, size) in a PLAIN GeometryList followed by a floor cuboid; accelerate applies GeometryList::accelerate.
Scene scene_scatter(int n, uint64_t seed, bool accelerate, std::shared_ptr<const Image> skybox, size_t cam_w, size_t cam_h) {
    SplitMix64 g{seed};
    Scene scene;
    for (int i = 0; i < n; i++) {
        float cx = -6.0f + 12.0f * g.uniform();
        float cy = -6.0f + 12.0f * g.uniform();
        float cz = 0.2f + 2.3f * g.uniform();
        float r = 0.1f + 0.3f * g.uniform();
        GeomPtr geom = (i % 2 == 0) ? Sphere(Vec3D(cx, cy, cz), r) : Cuboid::make(vec3d(cx - r, cy - r, cz - r), vec3d(cx + r, cy + r, cz + r));
        Material m = (i % 3 == 0) ? Metal() : ((i % 3 == 1) ? Lambartian(COLOR_WHITE) : Lambartian(Checkerboard(4)));
        scene.root.push(Object::with_material(geom, m));
    }
    if (accelerate) scene.root.accelerate(2);
    else scene.linear_root = true;   // the reference's linear walk, whatever the length of the list
    scene.root.push(Object::with_material(Cuboid::make(vec3d(-8.0f, -8.0f, 0.0f), vec3d(8.0f, 8.0f, -1.0f)), Glossy(1e3f, 0.3f, COLOR_WHITE)));
    Camera cam(cam_w, cam_h);
    cam.position(Vec3D(-9.0f, -9.0f, 6.0f)).look_at(Vec3D(0.0f, 0.0f, 0.8f), -Vec3D::z_axis());
    scene.camera = cam;
    scene.skybox = skybox;
    scene.lights = main_lights();
    return scene;
}

}  // namespace rth
