/*
 * vecchio_amd.h — C ABI of the MI355X-native per-pixel sample loop.
 *
 * This is the drop-in boundary for ONE hot path of browserdotsys/vecchio: the closure at
 * reference src/main.rs:181-198 (pixel loop) and everything it calls (ray_color
 * main.rs:123-153, BVHNode::hit accel.rs:58-83, every Hittable::hit in hittable.rs, every
 * Material/Texture/PDF in material.rs / util.rs).  Everything above it (scene.rs builders,
 * BVHNode::new, the frame loop and the PPM writer of main.rs:155-221) stays on the host
 * side of this boundary.
 *
 * The reference has no FFI of its own (no extern "C" anywhere), so the entry points below
 * are "what a cgo/FFI binding for this path would bind": one scene upload
 * (vk_scene_create), one blocking call per camera frame (vk_render) placed where
 * main.rs:181-198 is today.  The Rust-side binding a maintainer would add is shown in
 * INTEGRATION.md and vecchio_amd/rust_shim/.
 *
 * Conventions
 *   - plain C, POD structs, caller owns every pointer it passes, library owns the handle.
 *   - every entry point returns an int status (VK_OK == 0); nothing unwinds across the
 *     boundary; vk_last_error() gives a thread-local message for the last failure.
 *   - the scene crosses the boundary as a *graph of tagged records* that mirrors the
 *     reference's trait objects 1:1 (one record per Arc<dyn Hittable/Material/Texture>),
 *     so that `flatten()` on the Rust side is a one-record push per object.  The library
 *     linearises that graph for the GPU itself (threaded pre-order BVH, SoA primitives).
 */
#ifndef VECCHIO_AMD_H
#define VECCHIO_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_ABI_VERSION 6   /* 6: VK_SCENE_RCCL_GATHER, vk_scene_info.gather, vk_gather_backends; 5: rebuilt trees by default only where their exactness is proven; VK_SCENE_EMPIRICAL_TREES; vk_scene_info.tree */

/* ---- status codes (reference convention is panic!/unwrap, main.rs:166,202) ---------- */
enum {
    VK_OK = 0,
    VK_ERR_BAD_ARG = 1,      /* null pointer, bad index, time0>=time1 (main.rs:118 would panic) */
    VK_ERR_UNSUPPORTED = 2,  /* scene graph shape the device path does not implement      */
    VK_ERR_HIP = 3,          /* a HIP runtime call failed                                  */
    VK_ERR_NO_DEVICE = 4,    /* no gfx950 device / HIP runtime unusable                    */
    VK_ERR_OOM = 5
};

/* ---- object references -------------------------------------------------------------
 * A vk_ref names one Arc<dyn Hittable>: kind in bits 31..28, FlipFace parity in bit 27
 * (FlipFace, hittable.rs:294-312, only negates `front`, so wrapping is an XOR of this
 * bit), index into the per-kind array in bits 26..0.                                    */
typedef uint32_t vk_ref;
enum {
    VK_KIND_NONE = 0,
    VK_KIND_BVH = 1,           /* accel.rs:52-56                 */
    VK_KIND_SPHERE = 2,        /* hittable.rs:46-51              */
    VK_KIND_MOVING_SPHERE = 3, /* hittable.rs:136-144            */
    VK_KIND_RECT = 4,          /* hittable.rs:199-210            */
    VK_KIND_LIST = 5,          /* Vec<Arc<HittableSS>>, hittable.rs:380 (Boxy::sides) */
    VK_KIND_MEDIUM = 6,        /* hittable.rs:436-440            */
    VK_KIND_TRANSLATE = 7,     /* hittable.rs:500-504            */
    VK_KIND_ROTATE = 8         /* hittable.rs:534-539,631,720    */
};
#define VK_REF_FLIP 0x08000000u
#define VK_REF_INDEX_MASK 0x07FFFFFFu
#define VK_MAKE_REF(kind, index) ((((uint32_t)(kind)) << 28) | ((uint32_t)(index) & VK_REF_INDEX_MASK))
#define VK_REF_KIND(r) ((r) >> 28)
#define VK_REF_INDEX(r) ((r) & VK_REF_INDEX_MASK)

/* ---- hittables ---------------------------------------------------------------------- */
typedef struct vk_bvh_node {   /* accel.rs:52-56 — 32 bytes, the canonical node record */
    float bb_min[3];
    float bb_max[3];
    vk_ref left;
    vk_ref right;
} vk_bvh_node;

typedef struct vk_sphere {     /* hittable.rs:47-51; radius may be negative (scene.rs:123-127) */
    float center[3];
    float radius;
    uint32_t material;
} vk_sphere;

typedef struct vk_moving_sphere { /* hittable.rs:137-144 */
    float center0[3];
    float center1[3];
    float time0, time1;
    float radius;
    uint32_t material;
} vk_moving_sphere;

typedef struct vk_rect {       /* hittable.rs:200-210; XY=(0,1,2) XZ=(0,2,1) YZ=(1,2,0) */
    float c0, c1, d0, d1, k;
    uint8_t axis0, axis1, axis2, _pad;
    uint32_t material;
} vk_rect;

typedef struct vk_list {       /* Vec<Arc<HittableSS>>: items[first .. first+count) */
    uint32_t first;
    uint32_t count;
} vk_list;

typedef struct vk_medium {     /* hittable.rs:436-440; material = the Isotropic phase function */
    vk_ref boundary;
    float neg_inv_density;
    uint32_t material;
} vk_medium;

typedef struct vk_translate {  /* hittable.rs:500-504 */
    vk_ref child;
    float offset[3];
} vk_translate;

typedef struct vk_rotate {     /* hittable.rs:534-539 (Y), 631-636 (X), 720-725 (Z) */
    vk_ref child;
    uint32_t axis;             /* 0 = RotateX, 1 = RotateY, 2 = RotateZ */
    float sin_theta, cos_theta;
} vk_rotate;

/* ---- materials (material.rs) -------------------------------------------------------- */
enum {
    VK_MAT_LAMBERTIAN = 0,   /* material.rs:45-109   texture */
    VK_MAT_METAL = 1,        /* material.rs:111-142  texture, param = fuzz */
    VK_MAT_DIELECTRIC = 2,   /* material.rs:144-207  param = ref_idx */
    VK_MAT_DIFFUSE_LIGHT = 3,/* material.rs:209-226  texture = emit */
    VK_MAT_ISOTROPIC = 4,    /* material.rs:436-465  texture */
    VK_MAT_SPEC_DIFFUSE = 5  /* material.rs:467-488  a = specular material, b = diffuse material, param = pct */
};
typedef struct vk_material {
    uint32_t kind;
    uint32_t texture;
    float param;
    uint32_t a, b;
} vk_material;

enum {
    VK_TEX_SOLID = 0,   /* material.rs:233-242 color            */
    VK_TEX_CHECKER = 1, /* material.rs:244-259 a = odd, b = even (texture indices) */
    VK_TEX_IMAGE = 2,   /* material.rs:261-304 a = image index  */
    VK_TEX_NOISE = 3    /* material.rs:416-434 a = perlin index, scale */
};
typedef struct vk_texture {
    uint32_t kind;
    float color[3];
    uint32_t a, b;
    float scale;
} vk_texture;

typedef struct vk_image {      /* decoded 8-bit RGB, row 0 = top (material.rs:261-279) */
    uint32_t width, height;
    const uint8_t *rgb;        /* width*height*3 bytes */
} vk_image;

typedef struct vk_perlin {     /* material.rs:306-311 */
    float ranvec[256][3];
    uint32_t perm_x[256], perm_y[256], perm_z[256];
} vk_perlin;

/* ---- the flattened scene ------------------------------------------------------------ */
typedef struct vk_scene_desc {
    uint32_t abi_version;      /* must be VK_ABI_VERSION */
    uint32_t n_bvh;            const vk_bvh_node *bvh;
    uint32_t n_spheres;        const vk_sphere *spheres;
    uint32_t n_moving_spheres; const vk_moving_sphere *moving_spheres;
    uint32_t n_rects;          const vk_rect *rects;
    uint32_t n_lists;          const vk_list *lists;
    uint32_t n_list_items;     const vk_ref *list_items;
    uint32_t n_media;          const vk_medium *media;
    uint32_t n_translates;     const vk_translate *translates;
    uint32_t n_rotates;        const vk_rotate *rotates;
    uint32_t n_materials;      const vk_material *materials;
    uint32_t n_textures;       const vk_texture *textures;
    uint32_t n_images;         const vk_image *images;
    uint32_t n_perlins;        const vk_perlin *perlins;
    vk_ref world;              /* main.rs:168 world_bvh */
    uint32_t n_lights;         const vk_ref *lights;  /* main.rs:169 config.lights */
    uint32_t flags;            /* VK_SCENE_* (ABI 2); 0 = results of the tree handed over (see below) */
} vk_scene_desc;

/* vk_scene_desc.flags.
 * 0 (default): every result is the one BVHNode::hit (accel.rs:58-83) gives on the tree handed over.  The library walks that tree,
 * with one exception: a world of spheres only may be walked on a tree REBUILT over the reference's leaf units, "exact re-treeing"
 * (DESIGN.md section 5).  Every object stays gated by the box the reference gates it with, grown by a bound on how far off its sphere
 * an f32 Sphere::hit (hittable.rs:65-95) can report a hit; the rebuilt walk then finds every hit the reference's walk can accept,
 * and whenever its winner is not certain to be the reference's too — a hit not safely behind its own box's entry, a ray from outside
 * the region the bound was derived for — the tree as handed over decides (the segment is walked again, or its sample is rendered by
 * a second launch).  That this reproduces BVHNode::hit is a THEOREM given the bound (the "gate lemma"; forward error analysis, K < 30
 * against the 32 used; tests/test_gate_lemma.py attacks it with 10^7 adversarial rays).  Three proven forms (vk_scene_info.tree).  The
 * NEAR form (ABI 6, VK_TREE_REBUILT_NEAR): every sphere behind its OWN box, which is sound for ray origins within a trusted radius of
 * the sphere (~144 radii); a segment's result is taken only if its hit lies within that reach of its origin or the ray provably runs
 * clear of every small sphere beyond it, and every other segment is decided by the tree as handed over (both trees stay in device
 * memory; docs/gate_lemma.md section 7).  It is the default where its reach spans the world's small spheres
 * and where the other form is too dear (BVHNode::new's long leaf boxes on the 1 M-sphere
 * stress scene: +85 %).  The UNIT form (VK_TREE_REBUILT_PROVEN): the reference's leaf units as gates, grown by the bound — where that is
 * cheap and the near form's reach does not span the world.  And, where the world's small spheres lie in a layer across y and the scene
 * is staged in LDS (the InOneWeekend scene), no tree at all — the GRID form (VK_TREE_REBUILT_GRID): every sphere that can hold a
 * candidate for the ray is found through a grid over the layer and tested, the winner checked like the other forms' (docs/gate_lemma.md
 * section 8; +33 % throughput over the tree handed over).  A world for which no form applies is walked as handed over.
 * VK_SCENE_REFERENCE_TREE: walk the tree handed over and nothing else.
 * VK_SCENE_EMPIRICAL_TREES: allow the rebuilt tree also where NEITHER proven form applies (since ABI 6: worlds with a sphere far
 * smaller than the rest; the environment's VK_GATE_PROOF=0 prefers it to the proven forms, for comparisons) — with the units' boxes as
 * handed over and the closest hit so far padded by 1/16 instead.  NOT proven: a hit that precedes its unit's box entry by more than 1/16 and,
 * in the reference's visiting order only, wins against a hit inside that gap is missed.  It takes a ray that grazes a sphere where the
 * sphere touches its box, within ~1e-5 of parallel to that face of a long box; tests/test_gate_lemma.py constructs one and shows the
 * wrong result.  Measured on natural frames: 0 differing pixels in 12.6 G samples (40 worlds, profiles/r03/exact_retree_seeds.log);
 * the 1 M-sphere stress scene runs 1.7x faster than on the tree handed over.
 * VK_SCENE_FAST_ACCEL: the library may rebuild the acceleration structure over subtrees whose objects are all
 * Sphere / Rect / Boxy / lists of those (no ConstantMedium, no transform, no negative-radius sphere), object by object:
 * BVHNode::hit's result does not depend on the tree over such objects in exact arithmetic, and exact ties in t are resolved as the
 * reference resolves them.  What it cannot reproduce is floating-point noise: an f32 Sphere::hit that reports a hit OUTSIDE the
 * sphere's own bounding box is found or not depending on which enclosing boxes a tree happens to have — in the reference as much
 * as here, whose own tree is random (accel.rs:99-100).  Measured: the InOneWeekend scene's full 1920x1080x1024-spp frame is
 * bit-identical with and without the flag; on the 1 M-sphere stress scene 0.19 % of the samples differ — as many as between two
 * reference-style trees over the same world (profiles/r03/tree_variation.log).  Does not match a seeded reference run sample for
 * sample.
 * VK_SCENE_RCCL_GATHER (ABI 6; vk_scene_create_multi only, no effect on a pixel): the tile slabs travel to devices[0] by RCCL — one
 * communicator per device (ncclCommInitAll), one grouped ncclSend / ncclRecv pair per device and frame, on the devices' own streams —
 * instead of hipMemcpyPeerAsync.  librccl.so is loaded when the flag is first used, never linked; if it cannot be loaded, or a device
 * is listed twice (one communicator rank per device), the scene falls back to peer copies and says so on stderr
 * (vk_scene_info.gather tells which). */
enum { VK_SCENE_FAST_ACCEL = 1, VK_SCENE_REFERENCE_TREE = 2, VK_SCENE_EMPIRICAL_TREES = 4, VK_SCENE_RCCL_GATHER = 8 };

/* ---- camera: the ten fields of main.rs:57-68, computed by Camera::new on the host --- */
typedef struct vk_camera {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
    float time0, time1;
} vk_camera;

/* ---- render parameters: the reference's compile-time constants, made arguments ------ */
enum {
    VK_INTEGRATOR_PDF = 0,     /* HEAD ray_color, main.rs:123-153 (scatter_with_pdf + mixture PDF) */
    VK_INTEGRATOR_SCATTER = 1  /* InOneWeekend/TheNextWeek tags: emitted + attenuation*L via
                                  Material::scatter (material.rs:21-28,85-90,118-132,150-175,442-446) */
};
enum {
    VK_BACKGROUND_SOLID = 0,   /* main.rs:124 (HEAD: black) */
    VK_BACKGROUND_SKY = 1      /* InOneWeekend gradient (1-t)*white + t*(0.5,0.7,1.0), t = 0.5*(unit(d).y+1) */
};
typedef struct vk_render_params {
    uint32_t width, height;        /* main.rs:171-172 */
    uint32_t samples_per_pixel;    /* main.rs:28 */
    uint32_t max_depth;            /* main.rs:29; depth starts at 1, path stops when depth > max_depth */
    uint64_t seed;                 /* replaces rand::thread_rng(): counter-based, keyed (seed,pixel,sample) */
    uint32_t integrator;           /* VK_INTEGRATOR_* */
    uint32_t background;           /* VK_BACKGROUND_* */
    float background_color[3];     /* for VK_BACKGROUND_SOLID */
    /* pixel-tile partition for multi-GPU: 8x8-pixel tiles are dealt round-robin; this call
     * renders tiles t with t % tile_world == tile_rank.  (0,1) or (0,0) = whole image. */
    uint32_t tile_rank, tile_world;
    /* what vk_render / vk_render_device write (ABI 2): VK_OUTPUT_F32 = the pixel means as above;
     * VK_OUTPUT_RGB8 = the reference's output stage fused behind the render: Vec3::to_color
     * (vec3.rs:54-61) per pixel and the PPM writer's top-down row order (main.rs:209) —
     * width*height*3 BYTES, row 0 = TOP.  A multi-GPU host gathers these (4x less xGMI traffic). */
    uint32_t output_format;
} vk_render_params;
enum { VK_OUTPUT_F32 = 0, VK_OUTPUT_RGB8 = 1 };

typedef struct vk_stats {
    uint64_t samples;          /* pixel-samples rendered by this call            */
    double seconds;            /* wall seconds of the call (incl. gather/copies) */
    double kernel_ms;          /* HIP-event time of the megakernel launch(es)    */
    uint32_t kernel_launches;
    uint32_t scene_in_lds;     /* 1 if the linear BVH + primitives were LDS-resident */
    /* ABI 3.  Pixel sums (`c += color`, main.rs:193) are kept as 64-bit fixed point with 2^-26 resolution so that they do not
     * depend on the order samples finish in: a sample component below 2^-26 (1.5e-8) adds 0, and a component beyond
     * +-min(1e10, 1.3e11 / samples_per_pixel) is CLAMPED to that (the sums saturate, they never wrap).  The reference adds such
     * a sample in f32; this counts the samples of the call that were clamped (0 on every BASELINE config): non-zero means the
     * frame's brightest pixels deviate from the reference's.  vk_render fills it; after vk_render_device use
     * vk_scene_last_clamped_samples().                                                                                     */
    uint64_t clamped_samples;
} vk_stats;

typedef struct vk_scene vk_scene;  /* opaque */

/* replaces: nothing (version handshake for the Rust shim) */
int vk_abi_version(void);
/* replaces: nothing (reference is single-device CPU); number of usable gfx950 devices */
int vk_device_count(void);
/* replaces: panic!/unwrap messages (main.rs:166,202); thread-local, never NULL */
const char *vk_last_error(void);
/* how a multi-device scene can move its tile slabs: bit 0 = peer copies (always), bit 1 = RCCL (librccl.so loads and exports what
 * VK_SCENE_RCCL_GATHER needs).  Touches no device.  replaces: nothing */
int vk_gather_backends(void);

/* replaces: the ownership hand-off at main.rs:168-169 (Arc::new(BVHNode::new(..)),
 * Arc::new(config.lights)): deep-copies the described graph, linearises it and uploads it
 * to `device`.  The scene DESCRIPTION is immutable afterwards and may be rendered many times
 * (RotatingCamera, scene.rs:65-91).  The handle also owns per-launch scratch (work counter,
 * chunk partials, tile order): AT MOST ONE render may be in flight per vk_scene — calls on one
 * scene must be made from one thread at a time and be stream-ordered (the reference's frame loop,
 * main.rs:176, is exactly that); concurrent frames need one vk_scene each.                */
int vk_scene_create(const vk_scene_desc *desc, int device, vk_scene **out);
/* same, uploaded to EVERY device in devices[0..n_devices) ("uploads to every participating
 * GPU").  vk_render / vk_render_device on such a scene deal this call's 8x8 tiles round-robin over the
 * devices, render each share on that device's own stream, move the tile slabs to devices[0] with peer
 * copies over xGMI — or, with VK_SCENE_RCCL_GATHER, with RCCL send / receive pairs — (one message per device: the path's only
 * exchange), de-interleave them
 * on devices[0] and, for vk_render, do ONE device-to-host copy.  The image is bit-identical to the
 * one-device image.  A device may be listed more than once (shares run concurrently on it).       */
int vk_scene_create_multi(const vk_scene_desc *desc, const int *devices, int n_devices, vk_scene **out);
void vk_scene_destroy(vk_scene *scene);

/* replaces: the closure body at main.rs:181-198 for one Camera yielded by cam_iter
 * (main.rs:176).  Blocking.  rgb_out is caller-owned, width*height*3 floats, index
 * (y*width + x)*3 with y = 0 the BOTTOM row (main.rs:182-183,209) — or, with
 * params->output_format == VK_OUTPUT_RGB8, width*height*3 bytes, top row first.  Pixels outside
 * this call's tile partition are left untouched.  max_depth == 0 renders the reference's result
 * for MAX_DEPTH = 0: every sample is (0,0,0) (main.rs:126-128).                            */
int vk_render(vk_scene *scene, const vk_camera *cam, const vk_render_params *params,
              float *rgb_out, vk_stats *stats_out);

/* same as vk_render but the framebuffer is a device pointer on the scene's device and the
 * work is enqueued on `hip_stream` (a hipStream_t, or NULL for the default stream) without
 * a host synchronisation; used by the multi-GPU host (one process per GPU) so the RCCL
 * gather can be enqueued behind it.  stats_out->kernel_ms is not filled.  For a multi-device
 * scene the pointer and the stream belong to devices[0].                                 */
int vk_render_device(vk_scene *scene, const vk_camera *cam, const vk_render_params *params,
                     void *d_rgb_out, void *hip_stream, vk_stats *stats_out);

/* replaces: Vec3::to_color (vec3.rs:54-61) applied per pixel at main.rs:211 — sqrt gamma,
 * clamp to [0,0.999], *256, truncate — plus the top-down row order of main.rs:209.
 * d_rgb: width*height*3 floats (y up); d_rgb8_out: width*height*3 bytes, row 0 = TOP.  */
int vk_to_color_device(vk_scene *scene, const void *d_rgb, uint32_t width, uint32_t height,
                       void *d_rgb8_out, void *hip_stream);

/* ---- tile slabs: the exchange format of the multi-GPU host.  Rank r of w renders the 8x8 tiles t = r, r + w, r + 2w ...
 * (vk_render_params.tile_rank / tile_world); its SLAB is those tiles packed one after the other, 64 pixel slots per tile (slot
 * (y % 8) * 8 + x % 8; slots outside the image are present and unused), 3 components per slot: floats (VK_OUTPUT_F32) or bytes through
 * Vec3::to_color (VK_OUTPUT_RGB8, vec3.rs:54-61: a quarter of the traffic).  A one-process-per-GPU host calls vk_render_device
 * (VK_OUTPUT_F32) and vk_pack_tiles_device on every rank, gathers the equal-sized slabs on rank 0 (RCCL: ONE message per GPU, the
 * path's only exchange) and calls vk_unpack_tiles_device once per rank there.  replaces: nothing (single address space, main.rs:181). */
/* bytes of one rank's slab; the largest over the ranks (they differ by at most one tile) when tile_rank >= tile_world */
size_t vk_tile_slab_bytes(uint32_t width, uint32_t height, uint32_t output_format, uint32_t tile_rank, uint32_t tile_world);
/* d_fb: this rank's f32 framebuffer (width*height*3 floats, y up) on the scene's device -> d_slab */
int vk_pack_tiles_device(vk_scene *scene, const void *d_fb, uint32_t width, uint32_t height, uint32_t output_format,
                         uint32_t tile_rank, uint32_t tile_world, void *d_slab, void *hip_stream);
/* d_slab of rank tile_rank -> its tiles of the full image d_img on the scene's device: f32, y up (VK_OUTPUT_F32), or bytes, top row
 * first (VK_OUTPUT_RGB8, main.rs:209) */
int vk_unpack_tiles_device(vk_scene *scene, const void *d_slab, uint32_t width, uint32_t height, uint32_t output_format,
                           uint32_t tile_rank, uint32_t tile_world, void *d_img, void *hip_stream);

/* introspection used by bench/tests: bytes of the linearised scene, item counts */
typedef struct vk_scene_info {
    uint32_t n_items;          /* 32-byte linear BVH records */
    uint32_t n_prims;          /* primitive records */
    uint32_t n_instances;
    uint64_t device_bytes;
    uint32_t lds_bytes;        /* bytes staged into LDS per workgroup (0 = not resident) */
    uint32_t features;         /* VKF_* mask of the kernel variant selected */
    uint32_t tree;             /* VK_TREE_*: what the world is walked on (ABI 5; see vk_scene_desc.flags) */
    /* frames for which a rebuilt tree is suspended (the tree as handed over is walked meanwhile): a frame that sends more than a quarter
     * of its samples to the tree as handed over anyway, or more than the queues between the two launches hold, pauses the rebuilt tree
     * for 32 frames, twice as long at every relapse; 0 = in use.  As of the last frame whose end the library has seen. */
    uint32_t tree_suspended_frames;
    uint32_t gather;           /* VK_GATHER_*: how a multi-device scene moves its tile slabs to devices[0] (ABI 6) */
} vk_scene_info;
enum { VK_GATHER_NONE = 0 /* one device */, VK_GATHER_PEER_COPY = 1 /* hipMemcpyPeerAsync over xGMI */, VK_GATHER_RCCL = 2 };
enum {
    VK_TREE_HANDED_OVER = 0,        /* the tree of the description, item for item */
    VK_TREE_REBUILT_PROVEN = 1,     /* exact re-treeing with grown gates: results proven to be the handed-over tree's */
    VK_TREE_REBUILT_EMPIRICAL = 2,  /* exact re-treeing without them (VK_SCENE_EMPIRICAL_TREES): measured, not proven */
    VK_TREE_REBUILT_FAST = 3,       /* VK_SCENE_FAST_ACCEL */
    VK_TREE_REBUILT_NEAR = 4,       /* exact re-treeing, near form (ABI 6): every sphere behind its own box, a segment's result taken only
                                       where no sphere beyond that box's trusted radius can matter, else walked again on the tree handed
                                       over: proven like VK_TREE_REBUILT_PROVEN; taken first where its reach spans the world's small
                                       spheres, and for worlds whose leaf units are too long for the unit form */
    VK_TREE_REBUILT_GRID = 5        /* exact re-treeing, grid form (ABI 6, round 5): no tree at all — a world of spheres whose small spheres
                                       lie in a layer across y is walked on a uniform grid over the layer; every sphere that can hold a
                                       candidate for the ray is tested, the winner is checked like the other forms' and the tree handed
                                       over decides where that fails: proven like them (docs/gate_lemma.md section 8) */
};
int vk_scene_get_info(const vk_scene *scene, vk_scene_info *out);

/* One part of a scene — a multi-device scene has one per entry of devices[], an ordinary scene one — for a host that wants to SAY what
 * ran where (bench.py --in-library): the device's index, name and PCI bus id, whether it can address devices[0]'s memory (if not, its
 * tile slab travels through host memory), and the HIP-event time of the part's launches in the last frame (waits for them; -1 before
 * the first frame).  replaces: nothing (ABI 6) */
typedef struct vk_part_info {
    uint32_t n_parts;
    int32_t device;
    char name[64];
    char pci_bus_id[32];
    uint32_t can_access_landing_device;
    double kernel_ms;
} vk_part_info;
int vk_scene_part_info(vk_scene *scene, int part, vk_part_info *out);

/* HIP-event time (ms) of the launches enqueued by the last vk_render / vk_render_device on
 * this scene, on the stream they were launched on; waits for their end event.  (For a
 * multi-device scene: the slowest device's time.)                                         */
int vk_scene_last_kernel_ms(vk_scene *scene, double *ms_out);

/* see vk_stats.clamped_samples; waits for the end of the last render enqueued on this scene */
int vk_scene_last_clamped_samples(vk_scene *scene, uint64_t *count_out);
/* Exact re-treeing (see vk_scene_desc.flags): samples of the last render that were rendered by the second launch, on the tree as
 * handed over; waits for the render's end. */
int vk_scene_last_requeued_samples(vk_scene *scene, uint64_t *count_out);

/* test/diagnostic entry points (vk_debug_*) are declared in vecchio_amd_debug.h */

#ifdef __cplusplus
}
#endif
#endif /* VECCHIO_AMD_H */
