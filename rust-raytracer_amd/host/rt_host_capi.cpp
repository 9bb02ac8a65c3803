// C facade over the host library so that Python (ctypes) can drive it: build one of the scene
// scripts, upload it to a GPU context, inspect its flattened arrays, trace and render.
#include <chrono>
#include <cstdio>
#include <cstring>

#include "rt_host.hpp"

using namespace rth;

namespace {
struct Handle {
    Scene scene;
    std::unique_ptr<Device> dev;
    std::vector<std::unique_ptr<Device>> more;   // further contexts of rth_render_multi
    std::string error;
};
thread_local std::string g_error;
std::shared_ptr<const Image> sky_from(const uint8_t* rgb, int w, int h) {
    if (rgb && w > 0 && h > 0) return Image::from_rgb(rgb, (uint32_t)w, (uint32_t)h);
    return nullptr;
}
template <class F>
void* guarded(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        g_error = e.what();
        std::fprintf(stderr, "rth: %s\n", e.what());
        return nullptr;
    }
}
}  // namespace

extern "C" {

const char* rth_last_error() { return g_error.c_str(); }

void* rth_scene_main(const char* ply, const uint8_t* sky, int sw, int sh, int cam_w, int cam_h) {
    return guarded([&]() -> void* { auto* h = new Handle(); h->scene = scene_main(ply, sky_from(sky, sw, sh), (size_t)cam_w, (size_t)cam_h); return h; });
}
void* rth_scene_spheres(int n, uint64_t seed, const uint8_t* sky, int sw, int sh, int cam_w, int cam_h) {
    return guarded([&]() -> void* { auto* h = new Handle(); h->scene = scene_spheres(n, seed, sky_from(sky, sw, sh), (size_t)cam_w, (size_t)cam_h); return h; });
}
// The same scene with every length multiplied by `scale` (a tree of tiny objects far from the rays that meet it).
void* rth_scene_spheres_scaled(int n, uint64_t seed, float scale, const uint8_t* sky, int sw, int sh, int cam_w, int cam_h) {
    return guarded([&]() -> void* { auto* h = new Handle(); h->scene = scene_spheres(n, seed, sky_from(sky, sw, sh), (size_t)cam_w, (size_t)cam_h, scale); return h; });
}
void* rth_scene_grid(const char* ply, int nx, int ny, const uint8_t* sky, int sw, int sh, int cam_w, int cam_h) {
    return guarded([&]() -> void* { auto* h = new Handle(); h->scene = scene_grid(ply, nx, ny, sky_from(sky, sw, sh), (size_t)cam_w, (size_t)cam_h); return h; });
}
void* rth_scene_grid2(const char* ply, int nx, int ny, int flat, const uint8_t* sky, int sw, int sh, int cam_w, int cam_h) {
    // flat: 0 instanced, 1 one world-space mesh per instance, 2 / 3 one merged world-space mesh per material with its tree
    // built on the host (2) or on GPU 0 (3)
    return guarded([&]() -> void* {
        auto* h = new Handle();
        std::unique_ptr<Device> builder;
        if (flat == 3) builder.reset(new Device(0));
        h->scene = scene_grid(ply, nx, ny, sky_from(sky, sw, sh), (size_t)cam_w, (size_t)cam_h, flat >= 2 ? 2 : flat, builder.get());
        return h;
    });
}
void* rth_scene_scatter(int n, uint64_t seed, int accelerate, const uint8_t* sky, int sw, int sh, int cam_w, int cam_h) {
    return guarded([&]() -> void* { auto* h = new Handle(); h->scene = scene_scatter(n, seed, accelerate != 0, sky_from(sky, sw, sh), (size_t)cam_w, (size_t)cam_h); return h; });
}
void* rth_scene_terrain(const char* ply, int n, uint64_t seed, int with_bunny, const uint8_t* sky, int sw, int sh, int cam_w, int cam_h) {
    return guarded([&]() -> void* { auto* h = new Handle(); h->scene = scene_terrain(ply, n, seed, with_bunny != 0, sky_from(sky, sw, sh), (size_t)cam_w, (size_t)cam_h); return h; });
}
// tex = NULL: the textured sphere takes the skybox image
void* rth_scene_textured(const char* ply, const uint8_t* sky, int sw, int sh, const uint8_t* tex, int tw, int th, int cam_w, int cam_h) {
    return guarded([&]() -> void* {
        auto* h = new Handle();
        std::shared_ptr<const Image> timg;
        if (tex && tw > 0 && th > 0) timg = Image::from_rgb(tex, (uint32_t)tw, (uint32_t)th);
        h->scene = scene_textured(ply, sky_from(sky, sw, sh), timg, (size_t)cam_w, (size_t)cam_h);
        return h;
    });
}
void rth_scene_free(void* p) { delete (Handle*)p; }

// Camera::new(w, h).position(pos).look_towards(dir, up) at another resolution, keeping the pose:
// re-derives the image-plane vectors exactly like the builder does (scene.rs:35-78).
void rth_scene_set_camera_size(void* p, int w, int h, const float pos[3], const float lookat[3], const float up[3]) {
    Handle* hd = (Handle*)p;
    Camera cam((size_t)w, (size_t)h);
    cam.position(Vec3D(pos[0], pos[1], pos[2])).look_at(Vec3D(lookat[0], lookat[1], lookat[2]), Vec3D(up[0], up[1], up[2]));
    hd->scene.camera = cam;
}
void rth_scene_camera(void* p, rtamd_camera* out) { *out = ((Handle*)p)->scene.camera.flat(); }

// Opens the GPU (once) and uploads the flattened scene.  Returns the rtamd context or NULL.
void* rth_scene_upload(void* p, int device_id) {
    Handle* h = (Handle*)p;
    return guarded([&]() -> void* {
        if (!h->dev) h->dev.reset(new Device(device_id));
        h->scene.upload(*h->dev);
        return h->dev->ctx;
    });
}

// Host-side flattening without a GPU: fills the flat arrays so that tests can compare tree shape with
// the CPU oracle.  which: 0 = first mesh tree, 1 = first object tree.
int rth_scene_tree_size(void* p, int which, uint32_t out[2]) {
    Handle* h = (Handle*)p;
    const BvhTree* t = nullptr;
    for (const auto& it : h->scene.root.items) {
        if (which == 1 && it.is_tree) { t = &it.tree->tree; break; }
    }
    if (which == 0) {
        // find the first mesh by flattening into a scratch FlatScene
        FlatScene fs;
        std::vector<rtamd_xform_op> ops;
        for (const auto& it : h->scene.root.items) {
            if (it.is_tree) { for (const auto& o : it.tree->objs) { o.geometry->flatten(fs, ops, 0); if (!fs.mesh_refs.empty()) break; } }
            else it.object.geometry->flatten(fs, ops, 0);
            if (!fs.mesh_refs.empty()) break;
        }
        if (fs.mesh_refs.empty()) return -1;
        out[0] = (uint32_t)fs.mesh_refs[0]->tree.nodes.size();
        out[1] = (uint32_t)fs.mesh_refs[0]->tree.order.size();
        return 0;
    }
    if (!t) return -1;
    out[0] = (uint32_t)t->nodes.size(); out[1] = (uint32_t)t->order.size();
    return 0;
}
// AABBTree::new prints its statistics line (bvh.rs:173-204) from now on (on != 0) or not (the default)
void rth_set_print_stats(int on) { rth::bvh_print_stats_enabled() = on != 0; }
int rth_scene_tree_dump(void* p, int which, uint32_t* nodes8, uint32_t* order, uint32_t stats[5]) {
    Handle* h = (Handle*)p;
    const BvhTree* t = nullptr;
    std::shared_ptr<const Mesh> keep;
    if (which == 1) {
        for (const auto& it : h->scene.root.items) if (it.is_tree) { t = &it.tree->tree; break; }
    } else {
        FlatScene fs;
        std::vector<rtamd_xform_op> ops;
        for (const auto& it : h->scene.root.items) {
            if (it.is_tree) { for (const auto& o : it.tree->objs) { o.geometry->flatten(fs, ops, 0); if (!fs.mesh_refs.empty()) break; } }
            else it.object.geometry->flatten(fs, ops, 0);
            if (!fs.mesh_refs.empty()) break;
        }
        if (!fs.mesh_refs.empty()) { keep = fs.mesh_refs[0]; t = &keep->tree; }
    }
    if (!t) return -1;
    static_assert(sizeof(rtamd_bvh_node) == 32, "node layout");
    std::memcpy(nodes8, t->nodes.data(), t->nodes.size() * 32);
    std::memcpy(order, t->order.data(), t->order.size() * 4);
    bvh_statistics(*t, stats);
    return 0;
}

// render::parallel_render on the uploaded scene; out = width*height*3 floats.
int rth_render(void* p, const int params[4], float* out, rtamd_stats* stats) {
    Handle* h = (Handle*)p;
    if (!h->dev) { g_error = "scene not uploaded"; return -1; }
    try {
        WhittedIntegrator ig;
        ig.max_depth = params[0]; ig.shadow_rays = params[1]; ig.scatter_rays = params[2]; ig.antialiasing = params[3];
        std::vector<float> buf = render::parallel_render(*h->dev, h->scene, ig, stats, true);
        std::memcpy(out, buf.data(), buf.size() * 4);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// render::parallel_render over n_ctx contexts: context k lives on GPU device_ids[k] (the same id may repeat: several
// contexts on one GPU).  The scene is uploaded to every context on first use.  out = width*height*3 floats.
int rth_render_multi(void* p, int n_ctx, const int* device_ids, const int params[4], float* out, rtamd_stats* stats) {
    Handle* h = (Handle*)p;
    try {
        if (n_ctx < 1) throw std::runtime_error("no context");
        if (!h->dev) { h->dev.reset(new Device(device_ids[0])); h->scene.upload(*h->dev); }
        while ((int)h->more.size() < n_ctx - 1) {
            h->more.emplace_back(new Device(device_ids[h->more.size() + 1]));
            h->scene.upload_flat(*h->more.back());
        }
        std::vector<Device*> devs{h->dev.get()};
        for (int k = 0; k < n_ctx - 1; k++) devs.push_back(h->more[k].get());
        WhittedIntegrator ig;
        ig.max_depth = params[0]; ig.shadow_rays = params[1]; ig.scatter_rays = params[2]; ig.antialiasing = params[3];
        std::vector<float> buf = render::parallel_render(devs, h->scene, ig, stats, true);
        std::memcpy(out, buf.data(), buf.size() * 4);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// AABBTree::new over n boxes (6 floats each) on the host (device < 0) or on GPU `device`; nodes_out: 2 n + 1 records.
int rth_build_tree(int device, const float* boxes, uint32_t n, float hit_cost, rtamd_bvh_node* nodes_out, uint32_t* order_out, uint32_t* n_nodes_out, double* seconds) {
    try {
        std::vector<AABB> bx(n);
        for (uint32_t k = 0; k < n; k++) bx[k] = AABB(Vec3D(boxes[6 * k], boxes[6 * k + 1], boxes[6 * k + 2]), Vec3D(boxes[6 * k + 3], boxes[6 * k + 4], boxes[6 * k + 5]));
        BvhTree t;
        std::unique_ptr<Device> dev;
        if (device >= 0) dev.reset(new Device(device));
        auto t0 = std::chrono::steady_clock::now();
        t = device >= 0 ? build_aabb_tree(*dev, bx, hit_cost) : build_aabb_tree(bx, hit_cost);
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::memcpy(nodes_out, t.nodes.data(), t.nodes.size() * sizeof(rtamd_bvh_node));
        std::memcpy(order_out, t.order.data(), t.order.size() * 4);
        *n_nodes_out = (uint32_t)t.nodes.size();
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// Loader and image-writer probes for the tests.
int rth_ply_info(const char* path, uint32_t counts[2], float* first_vertices, uint32_t* first_faces, int n_first) {
    try {
        std::vector<Vec3D> v;
        std::vector<std::array<uint32_t, 3>> f;
        loader::load_ply(path, v, f);
        counts[0] = (uint32_t)v.size(); counts[1] = (uint32_t)f.size();
        for (int k = 0; k < n_first && k < (int)v.size(); k++) { first_vertices[3 * k] = v[k].x; first_vertices[3 * k + 1] = v[k].y; first_vertices[3 * k + 2] = v[k].z; }
        for (int k = 0; k < n_first && k < (int)f.size(); k++) { first_faces[3 * k] = f[k][0]; first_faces[3 * k + 1] = f[k][1]; first_faces[3 * k + 2] = f[k][2]; }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
int rth_save_png(const char* path, const uint8_t* rgb, int w, int h) {
    try {
        render::save_png(path, std::vector<uint8_t>(rgb, rgb + (size_t)w * h * 3), (size_t)w, (size_t)h);
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
void rth_kat_reflection(const float axis[3], float out[9]) {
    Mat3D m = Mat3D::new_reflection(Vec3D(axis[0], axis[1], axis[2]));
    std::memcpy(out, m.m, 36);
}

}  // extern "C"
