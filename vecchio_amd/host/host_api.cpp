// host_api.cpp — C entry points over the C++ host mirror (see host_api.h)
#include "host_api.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "vecchio_host.hpp"

using namespace vecchio;

struct vkh_scene {
    SceneConfig cfg;
    FlatBuilder fb;
    vk_scene_desc desc;
};

static thread_local std::string g_herr;

extern "C" {

const char *vkh_last_error(void) { return g_herr.c_str(); }

vkh_scene *vkh_scene_build(const char *name, uint64_t seed) {
    g_herr.clear();
    if (!name) { g_herr = "null scene name"; return nullptr; }
    try {
        seed_thread_rng(seed);
        std::string n(name);
        // "<scene>+sah": every BVHNode::new of the scene runs the SAH builder instead (same objects, same geometry)
        set_bvh_builder(BvhBuilder::REFERENCE);
        if (n.size() > 4 && n.compare(n.size() - 4, 4, "+sah") == 0) { n.resize(n.size() - 4); set_bvh_builder(BvhBuilder::SAH); }
        // "<scene>+treeseed:<n>": the same objects, but main()'s BVHNode::new (main.rs:168) draws its split axes from another stream:
        // the reference's tree is random (accel.rs:99-100, unseeded), so two runs of it walk different trees over the same world
        long tree_seed = -1;
        {
            size_t at = n.rfind("+treeseed:");
            if (at != std::string::npos) { tree_seed = atol(n.c_str() + at + 10); n.resize(at); }
        }
        auto s = new vkh_scene;
        if (n == "balls_demo") s->cfg = balls_demo();
        else if (n == "random_spheres_demo") s->cfg = random_spheres_demo();
        else if (n == "random_spheres_iow") s->cfg = random_spheres_iow(11);
        else if (n == "perlin_demo") s->cfg = perlin_demo();
        else if (n == "bowser_demo") s->cfg = bowser_demo();
        else if (n == "cornell_box") s->cfg = cornell_box();
        else if (n == "final_scene") s->cfg = final_scene();
        else if (n == "final_scene_nextweek") s->cfg = final_scene_nextweek();
        else if (n.rfind("stress_spheres:", 0) == 0) {
            int g = atoi(n.c_str() + 15);
            if (g < 1 || g > 2000) { delete s; g_herr = "stress_spheres grid_half out of range"; return nullptr; }
            s->cfg = random_spheres_iow(g);
        } else { delete s; g_herr = "Not a valid scene: " + n; return nullptr; }  // main.rs:166
        // main.rs:168-169
        if (tree_seed >= 0) seed_thread_rng((uint64_t)tree_seed * 0x9E3779B97F4A7C15ull + 0x5EEDull);
        auto world_bvh = BVHNode::build(s->cfg.world);
        s->fb.world = s->fb.hittable(world_bvh);
        for (auto &l : s->cfg.lights) s->fb.lights.push_back(s->fb.hittable(l));
        s->desc = s->fb.desc();
        return s;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return nullptr;
    }
}

void vkh_set_assets_dir(const char *dir) { set_assets_dir(dir ? dir : ""); }
void vkh_scene_free(vkh_scene *s) { delete s; }
const vk_scene_desc *vkh_scene_desc(vkh_scene *s) { return s ? &s->desc : nullptr; }
int vkh_scene_next_camera(vkh_scene *s, vk_camera *out) {
    if (!s || !out || !s->cfg.cam_iter) return 0;
    return s->cfg.cam_iter(*out) ? 1 : 0;
}
void vkh_scene_defaults(vkh_scene *s, float *aspect_ratio, uint32_t *integrator, uint32_t *background, float background_color[3]) {
    if (aspect_ratio) *aspect_ratio = s->cfg.aspect_ratio;
    if (integrator) *integrator = s->cfg.integrator;
    if (background) *background = s->cfg.background;
    if (background_color) { background_color[0] = s->cfg.background_color.x; background_color[1] = s->cfg.background_color.y;
        background_color[2] = s->cfg.background_color.z; }
}
void vkh_camera_new(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov, float aspect_ratio,
                    float aperture, float focus_dist, float time0, float time1, vk_camera *out) {
    *out = camera_new(Vec3(lookfrom[0], lookfrom[1], lookfrom[2]), Vec3(lookat[0], lookat[1], lookat[2]), Vec3(vup[0], vup[1], vup[2]),
                      vfov, aspect_ratio, aperture, focus_dist, time0, time1);
}

static inline float clampf(float x, float mn, float mx) { return x < mn ? mn : (x > mx ? mx : x); }  // vec3.rs:44-52
static inline uint32_t to_u32(float f) { return vk::sat_u32(f); }                                      // `as u32` saturates

void vkh_to_color(const float *rgb, uint32_t width, uint32_t height, uint8_t *out) {
    for (uint32_t row = 0; row < height; row++) {
        uint32_t y = height - 1 - row;  // main.rs:209
        for (uint32_t x = 0; x < width; x++)
            for (int c = 0; c < 3; c++) {
                float v = rgb[((size_t)y * width + x) * 3 + c];
                out[((size_t)row * width + x) * 3 + c] = (uint8_t)to_u32(256.0f * clampf(sqrtf(v), 0.0f, 0.999f));  // vec3.rs:54-61
            }
    }
}

int vkh_write_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height) {
    FILE *f = fopen(path, "w");
    if (!f) { g_herr = std::string("cannot create ") + path; return 1; }
    fprintf(f, "P3\n%u %u\n255\n", width, height);  // main.rs:205-207
    for (uint32_t row = 0; row < height; row++) {
        uint32_t y = height - 1 - row;
        for (uint32_t x = 0; x < width; x++) {
            const float *p = &rgb[((size_t)y * width + x) * 3];
            fprintf(f, "%u %u %u\n", to_u32(256.0f * clampf(sqrtf(p[0]), 0.0f, 0.999f)), to_u32(256.0f * clampf(sqrtf(p[1]), 0.0f, 0.999f)),
                    to_u32(256.0f * clampf(sqrtf(p[2]), 0.0f, 0.999f)));
        }
    }
    fclose(f);
    return 0;
}

}  // extern "C"
