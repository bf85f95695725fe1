// main.cpp — harness playing the role of the reference's main() (main.rs:155-221): pick a
// scene, build the BVH, upload once, then per Camera of cam_iter: render through the C ABI
// (the call that replaces main.rs:181-198), write output_%04d.ppm, print the frame time.
// The reference hard-codes scene/width/spp/depth (main.rs:28-29,159-167,171); here they are
// arguments:   vecchio_cli <scene> [width=900] [spp=1000] [max_depth=100] [frames=1] [seed=1]
// Texture images are read from ./assets (as in the reference) or $VECCHIO_ASSETS: <name>.ppm.gz, see host_api.h.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "host_api.h"

template <class T>
static T sym(void *h, const char *name) {
    void *p = dlsym(h, name);
    if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
    return reinterpret_cast<T>(p);
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <balls_demo|random_spheres_demo|random_spheres_iow|perlin_demo|bowser_demo|cornell_box|final_scene|"
                        "final_scene_nextweek|stress_spheres:N> [width] [spp] [max_depth] [frames] [seed]\n", argv[0]);
        return 2;
    }
    const char *name = argv[1];
    uint32_t width = argc > 2 ? (uint32_t)atoi(argv[2]) : 900;       // main.rs:171
    uint32_t spp = argc > 3 ? (uint32_t)atoi(argv[3]) : 1000;        // main.rs:28
    uint32_t depth = argc > 4 ? (uint32_t)atoi(argv[4]) : 100;       // main.rs:29
    int frames = argc > 5 ? atoi(argv[5]) : 1;
    uint64_t seed = argc > 6 ? strtoull(argv[6], nullptr, 10) : 1;

    std::string dir = argv[0];
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? "." : dir.substr(0, slash);
    void *h = dlopen((dir + "/libvecchio_amd.so").c_str(), RTLD_NOW);
    if (!h) { fprintf(stderr, "cannot load libvecchio_amd.so: %s\n", dlerror()); return 2; }
    auto p_create = sym<int (*)(const vk_scene_desc *, int, vk_scene **)>(h, "vk_scene_create");
    auto p_render = sym<int (*)(vk_scene *, const vk_camera *, const vk_render_params *, float *, vk_stats *)>(h, "vk_render");
    auto p_destroy = sym<void (*)(vk_scene *)>(h, "vk_scene_destroy");
    auto p_err = sym<const char *(*)()>(h, "vk_last_error");

    fprintf(stderr, "Generating scene...\n");                        // main.rs:157
    vkh_scene *hs = vkh_scene_build(name, seed);
    if (!hs) { fprintf(stderr, "%s\n", vkh_last_error()); return 1; }
    float aspect; uint32_t integ, bg; float bgc[3];
    vkh_scene_defaults(hs, &aspect, &integ, &bg, bgc);
    uint32_t height = (uint32_t)((float)width / aspect);              // main.rs:172
    vk_scene *scene = nullptr;
    if (p_create(vkh_scene_desc(hs), 0, &scene) != VK_OK) { fprintf(stderr, "vk_scene_create: %s\n", p_err()); return 1; }

    std::vector<float> pixels((size_t)width * height * 3, 0.0f);     // main.rs:173
    vk_render_params rp{};
    rp.width = width; rp.height = height; rp.samples_per_pixel = spp; rp.max_depth = depth; rp.seed = seed + 1;
    rp.integrator = integ; rp.background = bg; rp.background_color[0] = bgc[0]; rp.background_color[1] = bgc[1];
    rp.background_color[2] = bgc[2];
    rp.tile_rank = 0; rp.tile_world = 1;
    vk_camera cam;
    int file_idx = 0;
    while (file_idx < frames && vkh_scene_next_camera(hs, &cam)) {   // main.rs:176
        auto start = std::chrono::steady_clock::now();
        vk_stats st{};
        if (p_render(scene, &cam, &rp, pixels.data(), &st) != VK_OK) { fprintf(stderr, "vk_render: %s\n", p_err()); return 1; }
        char fn[64];
        snprintf(fn, sizeof fn, "output_%04d.ppm", file_idx);         // main.rs:201
        if (vkh_write_ppm(fn, pixels.data(), width, height)) { fprintf(stderr, "%s\n", vkh_last_error()); return 1; }
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        fprintf(stderr, "Wrote frame %s in %.3fs (kernel %.1f ms, %.1f Msamples/s)\n", fn, secs, st.kernel_ms,
                st.kernel_ms > 0 ? (double)st.samples / st.kernel_ms / 1e3 : 0.0);   // main.rs:215
        file_idx++;
    }
    p_destroy(scene);
    vkh_scene_free(hs);
    return 0;
}
