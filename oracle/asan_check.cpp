// asan_check.cpp — builds a few scenes with the C++ host mirror and renders small frames with
// the oracle under AddressSanitizer + UBSan (CPU only; GPU sanitizers are unavailable on this pool).
// Built and run by tests/test_sanitizers.py:  oracle.cpp + host sources + this file.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../vecchio_amd/host/host_api.h"
#include "oracle.h"

// tests/emu/emu.cpp (the lineariser of the device library + the kernel's per-lane code, built for the host)
extern "C" int emu_render(const vk_scene_desc *desc, const vk_camera *cam, const vk_render_params *p, float *rgb_out,
                          float *per_sample_out, int n_threads, uint64_t *steps_out, uint32_t *info_out);
extern "C" const char *emu_last_error(void);

int main() {
    const char *names[] = {"cornell_box", "final_scene", "random_spheres_iow", "random_spheres_demo", "perlin_demo", "balls_demo", "bowser_demo"};
    for (const char *name : names) {
        vkh_scene *hs = vkh_scene_build(name, 1);
        if (!hs) { fprintf(stderr, "%s: %s\n", name, vkh_last_error()); return 1; }
        float aspect; uint32_t integ, bg; float bgc[3];
        vkh_scene_defaults(hs, &aspect, &integ, &bg, bgc);
        vk_camera cam;
        if (!vkh_scene_next_camera(hs, &cam)) return 1;
        vk_render_params p;
        memset(&p, 0, sizeof(p));
        p.width = 24; p.height = (uint32_t)(24.0f / aspect); p.samples_per_pixel = 4; p.max_depth = 50; p.seed = 2;
        p.integrator = integ; p.background = bg; p.tile_rank = 0; p.tile_world = 1;
        std::vector<float> img((size_t)p.width * p.height * 3);
        oracle_counters cnt;
        int st = oracle_render(vkh_scene_desc(hs), &cam, &p, img.data(), 2, &cnt);
        if (st != 0) { fprintf(stderr, "%s: oracle status %d: %s\n", name, st, oracle_last_error()); return 1; }
        double sum = 0;
        for (float v : img) sum += v;
        printf("%s: %ux%u mean %.4f samples %llu\n", name, p.width, p.height, sum / img.size(), (unsigned long long)cnt.samples);
        // the device library's lineariser (vk_linearize.cpp: threaded layout, instances, Boxy recognition, and with
        // VK_SCENE_FAST_ACCEL the SAH re-treeing) + the kernel's per-lane logic, against the oracle's image
        for (uint32_t flags = 0; flags < 2; flags++) {
            vk_scene_desc d = *vkh_scene_desc(hs);
            d.flags = flags;
            std::vector<float> emu_img(img.size());
            uint64_t steps = 0; uint32_t info[4];
            st = emu_render(&d, &cam, &p, emu_img.data(), nullptr, 2, &steps, info);
            if (st != 0) { fprintf(stderr, "%s: emu status %d: %s\n", name, st, emu_last_error()); return 1; }
            double worst = 0;
            for (size_t i = 0; i < img.size(); i++) worst = fmax(worst, fabs((double)img[i] - (double)emu_img[i]));
            if (!(worst < 1e-4)) { fprintf(stderr, "%s: emulator (flags %u) differs from the oracle by %g\n", name, flags, worst); return 1; }
            printf("   lineariser flags %u: %u items, %llu steps, max |d| %.2e\n", flags, info[0], (unsigned long long)steps, worst);
        }
        vkh_scene_free(hs);
    }
    return 0;
}
