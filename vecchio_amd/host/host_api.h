/* host_api.h — C entry points of libvecchio_host.so: the C++ stand-in for the reference's
 * Rust host side (scene.rs builders + BVHNode::new + Camera::new), exposed so Python tests
 * and bench.py can obtain the same flattened scenes the CLI harness renders.
 * Not part of the drop-in boundary (that is include/vecchio_amd.h).                     */
#ifndef VECCHIO_HOST_API_H
#define VECCHIO_HOST_API_H
#include "../../include/vecchio_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vkh_scene vkh_scene;

/* name: "balls_demo" | "random_spheres_demo" | "random_spheres_iow" | "perlin_demo" |
 *       "cornell_box" | "final_scene" | "final_scene_nextweek" | "bowser_demo" | "stress_spheres:<grid_half>"
 * Builds the world list with the seeded build stream, wraps it in BVHNode::new
 * (main.rs:168) and flattens it.  Returns NULL on failure (vkh_last_error()).           */
vkh_scene *vkh_scene_build(const char *name, uint64_t seed);
/* directory holding the decoded copies (<name>.ppm.gz / .ppm) of the reference's assets/<name>.png, read by the builders
 * that call ImageTexture::new (scene.rs:230,355-398,588,822).  Default: $VECCHIO_ASSETS, else "assets" (cwd-relative,
 * as in the reference).  A missing image fails the build, as File::open(..).unwrap() does (material.rs:270).  */
void vkh_set_assets_dir(const char *dir);
void vkh_scene_free(vkh_scene *s);
const vk_scene_desc *vkh_scene_desc(vkh_scene *s);
/* next Camera of config.cam_iter (main.rs:176); returns 0 when exhausted */
int vkh_scene_next_camera(vkh_scene *s, vk_camera *out);
/* aspect ratio and the integrator/background the scene's git tag hard-codes */
void vkh_scene_defaults(vkh_scene *s, float *aspect_ratio, uint32_t *integrator, uint32_t *background,
                        float background_color[3]);
/* Camera::new (main.rs:71-109) */
void vkh_camera_new(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                    float aspect_ratio, float aperture, float focus_dist, float time0, float time1,
                    vk_camera *out);
const char *vkh_last_error(void);

/* ASCII PPM writer of main.rs:200-214 (P3, rows top-down, Vec3::to_color vec3.rs:54-61),
 * from a float framebuffer with y = 0 at the bottom.                                    */
int vkh_write_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height);
/* Vec3::to_color on the host, output row 0 = TOP (for checking vk_to_color_device) */
void vkh_to_color(const float *rgb, uint32_t width, uint32_t height, uint8_t *rgb8_out);

#ifdef __cplusplus
}
#endif
#endif
