/*
 * oracle.h — C API of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * The oracle is a recursive, structure-following restatement of the reference's per-pixel
 * sample loop (see oracle.cpp for the file:line map).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library (libvecchio_amd.so)
 * never links, loads or calls it.
 *
 * PARITY PINNING: the reference is Rust and cannot be built here (no cargo/rustc), has no
 * tests, golden vectors or seed hook (rand::thread_rng everywhere).  The oracle is
 * therefore pinned only statistically, against the one deterministic-scene artefact the
 * reference ships (sample/therestofyourlife.png -> tests/golden/cornell_blocks.json), and
 * by closed-form unit tests.  Per-sample parity with a reference run is "parity unpinned".
 */
#ifndef VECCHIO_ORACLE_H
#define VECCHIO_ORACLE_H
#include "../include/vecchio_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_counters {
    uint64_t samples;        /* pixel-samples traced                                   */
    uint64_t segments;       /* world.hit() calls from ray_color (main.rs:130)         */
    uint64_t n_aabb;         /* AxisBB::hit evaluations (accel.rs:16-35)               */
    uint64_t n_sphere;       /* Sphere::hit evaluations (hittable.rs:65-95)            */
    uint64_t n_moving;       /* MovingSphere::hit (hittable.rs:154-184)                */
    uint64_t n_rect;         /* Rect::hit (hittable.rs:230-256)                        */
    uint64_t n_xform;        /* Translate/Rotate*::hit (hittable.rs:507,579,676,765)   */
    uint64_t n_medium;       /* ConstantMedium::hit (hittable.rs:453-493)              */
    uint64_t n_closest;      /* segments that found a closest hit (material record read) */
    uint64_t n_texel;        /* ImageTexture::value fetches (material.rs:283-303)      */
    uint64_t n_perlin;       /* Perlin::noise calls = octaves (material.rs:392-413)    */
    uint64_t n_draws;        /* u32 draws consumed                                     */
    uint64_t n_dropped;      /* samples dropped by the finite filter (main.rs:192-194) */
    uint64_t n_aabb_nonfinite;   /* of n_aabb: tests for segments whose ray is NaN / infinite (it passes every box)   */
    uint64_t n_sphere_nonfinite; /* of n_sphere: likewise (it fails every sphere)                                      */
} oracle_counters;

/* Render the whole image (or this call's tile partition) on n_threads host threads
 * (pixel-parallel with dynamic stealing — the rayon par_iter_mut stand-in, main.rs:181).
 * rgb_out: width*height*3 floats, y = 0 bottom row.  counters_out may be NULL.         */
int oracle_render(const vk_scene_desc *desc, const vk_camera *cam, const vk_render_params *params,
                  float *rgb_out, int n_threads, oracle_counters *counters_out);

/* As oracle_render, also storing every sample: per_sample_out[(pixel*spp + s)*4 + 0..2] =
 * radiance before the finite filter, [+3] = the sample's u32 draw count (bit pattern).   */
int oracle_render_samples(const vk_scene_desc *desc, const vk_camera *cam, const vk_render_params *params,
                          float *rgb_out, float *per_sample_out, int n_threads);

/* Trace ONE sample of ONE pixel; returns its radiance (before the finite filter) and the
 * number of u32 draws it consumed.  Debug aid for per-sample parity.                   */
int oracle_sample(const vk_scene_desc *desc, const vk_camera *cam, const vk_render_params *params,
                  uint32_t pixel, uint32_t sample, float rgb_out[3], uint32_t *draws_out);

/* closest-hit query through the oracle's object tree (unit tests): returns 1 on hit.
 * rec_out = {p.xyz, normal.xyz, t, u, v, front, material index}                         */
int oracle_hit(const vk_scene_desc *desc, const float origin[3], const float dir[3], float time,
               float tmin, float tmax, uint64_t seed, float rec_out[11]);

/* shared-math probes (unit tests against libm): op 0 sin,1 cos,2 log,3 asin,4 atan2(a,b),5 pow5 */
void oracle_math(int op, const float *a, const float *b, float *out, size_t n);
/* draw probes: kind 0 gen_f32, 1 gen_range(lo,hi), 2 gen_index(n) (as float) */
void oracle_draws(uint64_t seed, uint32_t pixel, uint32_t sample, int kind, float lo, float hi,
                  uint32_t n_index, float *out, size_t n);

const char *oracle_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
