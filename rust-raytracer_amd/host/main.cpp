// Command-line renderer: the counterpart of the reference binary (src/main.rs:22-101).
//   rtamd_main [antialiasing=5] [--ply assets/bunny.ply] [--sky sky.ppm] [--size W H] [--device N]
// renders the reference's hard-coded scene on one MI355X with shadow_rays = 1, scatter_rays = 1,
// max_depth = 5 and writes test_<antialiasing>.ppm.  The skybox must be a binary PPM (no JPEG
// decoder is linked); without --sky a 1x1 grey sky is used.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt_host.hpp"

int main(int argc, char** argv) {
    using namespace rth;
    int aa = 5, device = 0;
    size_t w = 1080, h = 960;
    std::string ply = "assets/bunny.ply", sky;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--ply") && i + 1 < argc) ply = argv[++i];
        else if (!std::strcmp(argv[i], "--sky") && i + 1 < argc) sky = argv[++i];
        else if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--size") && i + 2 < argc) { w = (size_t)std::atoi(argv[++i]); h = (size_t)std::atoi(argv[++i]); }
        else aa = std::atoi(argv[i]);
    }
    try {
        bvh_print_stats_enabled() = true;   // AABBTree::new prints its statistics (bvh.rs:173-177)
        std::shared_ptr<const Image> skybox;
        if (!sky.empty()) skybox = Image::open_ppm(sky);
        Scene scene = scene_main(ply, skybox, w, h);
        Device dev(device);
        scene.upload(dev);
        WhittedIntegrator integrator;       // main.rs:54-56
        integrator.antialiasing = 4;
        integrator.shadow_rays = 20;
        WhittedIntegrator fast = integrator;  // main.rs:93-97
        fast.shadow_rays = 1;
        fast.scatter_rays = 1;
        fast.antialiasing = aa;
        fast.max_depth = 5;
        std::printf("WhittedIntegrator { max_depth: %d, shadow_rays: %d, scatter_rays: %d, antialiasing: %d, gamma: %.1f }\n",
                    fast.max_depth, fast.shadow_rays, fast.scatter_rays, fast.antialiasing, fast.gamma);
        std::vector<uint8_t> img = render::parallel_render_image(dev, scene, fast);
        std::string out = "test_" + std::to_string(aa) + ".ppm";
        render::save_ppm(out, img, w, h);
        std::printf("wrote %s\n", out.c_str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
