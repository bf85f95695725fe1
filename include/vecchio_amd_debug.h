/*
 * vecchio_amd_debug.h — test and diagnostic entry points.
 *
 * Not part of the drop-in boundary (include/vecchio_amd.h): nothing here replaces a reference
 * interface, and the Rust shim does not bind it.  Used by tests/ and bench.py only.
 * vk_debug_render_samples is in libvecchio_amd.so: it is vk_render with the PRODUCTION kernel's per-sample dump switched on.
 * vk_debug_phase_stats and vk_debug_math need kernels of their own (the instrumented STATS builds of the megakernel, the arithmetic
 * probe): they are in libvecchio_amd_debug.so, the same sources compiled with -DVK_DEBUG_LIB, so that the product library's code
 * object holds production kernels only.  A vk_scene belongs to the library that created it.
 */
#ifndef VECCHIO_AMD_DEBUG_H
#define VECCHIO_AMD_DEBUG_H

#include "vecchio_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* as vk_render, also returning every sample: samples_out[(pixel*spp + s)*4 + 0..2] = radiance
 * before the finite filter (main.rs:192), [+3] = the sample's u32 draw count (bit pattern) */
int vk_debug_render_samples(vk_scene *scene, const vk_camera *cam, const vk_render_params *params,
                            float *rgb_out, float *samples_out);
/* render with the instrumented build of the sphere-only kernel and return the wave scheduler's
 * counters: [0] box steps (wave level) [1] lanes with box work summed over them [2] PRIM phases
 * [3] lanes with primitive work in them [4] SHADE+REFILL phases [5] lanes in them [6] rounds
 * [7] heavy-primitive phases; wave clocks spent in [8] BOX [9] light PRIM [10] heavy PRIM
 * [11] SHADE+REFILL phases, [12] total wave clocks, [13..15] SHADE split (material / refill / install), [16] the cooperative Perlin
 * turbulence ahead of the material code, [17] the cold-state load of the shading lanes; [18..23] reserved (0).
 * Sphere-only builds have no PRIM phases of their own (sphere tests run inside the box loop) and reuse four slots for the
 * box loop's exit tests: [2] exit tests, [7] live lanes, [9] lanes with a pending test, [10] lanes waiting for shading, each
 * summed over the exit tests.                                                                                   */
int vk_debug_phase_stats(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, uint64_t out[24]);
/* evaluate the shared host/device arithmetic ON THE DEVICE (host arrays in/out):
 * op 0 sin, 1 cos, 2 ln, 3 asin, 4 atan2(a,b), 5 pow5, 6 a/b, 7 sqrt(a), 8 draws, 9 a*b+a  */
int vk_debug_math(int device, int op, const float *a, const float *b, float *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* VECCHIO_AMD_DEBUG_H */
