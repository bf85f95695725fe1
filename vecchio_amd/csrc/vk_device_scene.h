// vk_device_scene.h — the linearised scene the megakernel traverses (layout in HBM / LDS).
//
// The host lineariser (vk_linearize.cpp) turns the vk_scene_desc graph (one record per
// reference trait object) into:
//
//   items[]   32-byte records in PRE-ORDER of the reference's BVH (accel.rs:58-83 always
//             descends left first and shrinks tmax, so a fixed-order, stack-free "threaded"
//             traversal visits exactly the nodes BVHNode::hit visits, in the same order):
//               INNER  {bbox, w0 = skip index (kind nibble 0), w1 = 0}
//                      box hit -> next item (= left subtree); box miss -> items[w0]
//               LEAF   {bbox, w0 = dref A, w1 = dref B or 0}   (both children are objects)
//               an object child next to a BVH child (e.g. the world leaves of final_scene,
//               scene.rs:760-769) has no box of its own in the reference: it becomes a LEAF
//               {box = (-3e38, 3e38)^3 (always hit), w0 = dref, w1 = 0}
//   spheres[] float4 {cx,cy,cz,r} + sphere_mat[]      (SoA-by-type primitive buffers)
//   moving[], rects[], lists[]+list_refs[], media[], instances[] (transform chains)
//   materials[], textures[], images, perlin tables, lights[]
//
// dref = kind(4) | flip(1) | index(27); flip is the FlipFace parity (hittable.rs:294-312).
#ifndef VK_DEVICE_SCENE_H
#define VK_DEVICE_SCENE_H
#include <stdint.h>

namespace vkd {

enum : uint32_t {
    DK_NONE = 0, DK_SPHERE = 1, DK_MOVING = 2, DK_RECT = 3, DK_LIST = 4, DK_MEDIUM = 5, DK_INSTANCE = 6, DK_BOX = 7,
    DK_PRIM_ITEM = 15
};
constexpr uint32_t DREF_FLIP = 0x08000000u;
constexpr uint32_t DREF_INDEX = 0x07FFFFFFu;
#define VKD_KIND(r) ((r) >> 28)
#define VKD_INDEX(r) ((r) & 0x07FFFFFFu)
#define VKD_MAKE(kind, idx) (((uint32_t)(kind) << 28) | ((uint32_t)(idx) & 0x07FFFFFFu))

struct alignas(16) DItem {      // 32 B: the canonical BVH node record (24 B box + two 4 B links)
    float mnx, mxx, mny, mxy;   // box bounds interleaved per axis (min,max pairs feed packed f32 math)
    float mnz, mxz;
    uint32_t w0, w1;
};

struct alignas(16) DSphere { float cx, cy, cz, r; };                       // 16 B (hittable.rs:47-51)
struct alignas(16) DMoving { float c0[3], t0, c1[3], t1, r; uint32_t mat, _p0, _p1; };  // 48 B
struct alignas(16) DRect { float c0, c1, d0, d1, k; uint32_t axes; uint32_t mat, _p; };  // 32 B; axes = a0 | a1<<2 | a2<<4
struct DList { uint32_t first, count; };
// A Vec of exactly the six rects Boxy::new builds (hittable.rs:325-353), recognised by the
// lineariser and stored as its two corners: 32 B instead of 6 x 32 B + 6 refs, LDS-cacheable.
// Face f (= index in Boxy::sides): 0 XY@p1.z, 1 Flip XY@p0.z, 2 XZ@p1.y, 3 Flip XZ@p0.y, 4 YZ@p1.x, 5 Flip YZ@p0.x
struct alignas(16) DBox { float p0[3]; float p1x; float p1y, p1z; uint32_t mat, _p; };
struct alignas(16) DMedium { uint32_t boundary; float neg_inv_density; uint32_t mat, _p; };

enum : uint32_t { OP_TRANSLATE = 0, OP_ROTATE_X = 1, OP_ROTATE_Y = 2, OP_ROTATE_Z = 3 };
struct DOp { uint32_t kind; float a, b, c; };      // translate: offset xyz; rotate: a = sin, b = cos
constexpr int MAX_OPS = 4;                          // ops per instance record (longer chains nest)
constexpr int MAX_DEPTH_INST = 4;                   // instance nesting levels
struct alignas(16) DInstance {
    uint32_t n_ops;
    uint32_t child_begin, child_end;   // item range of a BVH child (begin < end), else 0,0
    uint32_t child_ref;                // dref of a non-BVH child (prim/list/medium/instance), else 0
    int32_t parent;                    // enclosing instance or -1
    uint32_t home_next;                // item index to resume at in the parent's range
    uint32_t home_pend;                // dref still to process in the home LEAF item (its w1) or 0
    uint32_t depth;                    // number of enclosing instances (0 = directly in world)
    int32_t chain[MAX_DEPTH_INST];     // ancestors outermost-first, including self at [depth]
    uint32_t flip;                     // FlipFace parity of the reference to this instance (DREF_FLIP or 0)
    uint32_t _p[3];
    DOp ops[MAX_OPS];                  // applied in order to the ray (outermost wrapper first)
};

struct alignas(16) DMaterial {         // 32 B: material record with its solid colour baked in
    uint32_t kind;                     // VK_MAT_*
    uint32_t tex;                      // texture index (for non-solid textures)
    float param;                       // fuzz / ref_idx / pct
    uint32_t tex_kind;                 // VK_TEX_* of `tex` (SOLID -> use rgb below, no second fetch)
    float r, g, b;                     // solid colour
    uint32_t ab;                       // SPEC_DIFFUSE: specular | diffuse << 16
};
struct alignas(16) DTexture { uint32_t kind; float r, g, b; uint32_t a, b_; float scale; uint32_t _p; };
struct DImage { uint32_t width, height; uint64_t offset; };   // offset into image_bytes
struct DPerlin { float ranvec[256][3]; uint8_t perm_x[256], perm_y[256], perm_z[256]; };

enum : uint32_t {  // feature bits -> kernel variant selection
    VKF_MOVING = 1u, VKF_RECT = 2u, VKF_LIST = 4u, VKF_MEDIUM = 8u, VKF_INSTANCE = 16u,
    VKF_TEXTURES = 32u,   // any non-solid texture (checker/image/noise)
    VKF_SPEC_DIFFUSE = 64u,
    VKF_BOX = 256u,       // canonical Boxy lists stored as DBox
    VKF_ALL_SCENE = 0x17Fu,
    VKF_INTEG_PDF = 128u, // not a scene property: selects the HEAD integrator (main.rs:123-153) at compile time
    VKF_NOISE = 512u      // the scene has a Perlin noise texture (implies VKF_TEXTURES; not a variant selector: read at run time by
                          // vk_kernels.h cooperative_turb)
};

// The GRID form of exact re-treeing (vk_linearize.cpp rt_build_grid, vk_trace.h grid_step; docs/gate_lemma.md section 8): a world of
// spheres whose small spheres lie in a layer (the reference's random_spheres worlds) is walked on a uniform 2-D grid over the layer's
// plane (x, z; the layer's normal is y) instead of a tree.  Every small sphere is registered in the cells its box overlaps; the few
// others (a ground sphere, large spheres: at most 8) are tested for every segment.  A segment visits the cells within
// dl = k (s_exit + r2) + slack of its path through the layer's box — as far as an f32 Sphere::hit can report a hit off its sphere
// (the residual bound of the gate lemma) — so EVERY sphere that holds a candidate is tested, wherever the ray starts.
struct DGrid {
    uint32_t nu, nv;              // cells along x and z (0: no grid); cell c = ix * nv + iz
    float ou, ov, cell, inv_cell; // the grid's corner and cell size
    float lo[3], hi[3];           // the box around the registered spheres' surfaces
    float k, r2, slack;           // the dilation: dl = k (s + r2) + slack
    uint32_t n_always;            // refs[0, n_always): the spheres tested for every segment; cells' lists follow
    // ... of which the LAST n_gated (large spheres, but not the ground's size) only when the ray passes the box [alo, ahi] around them,
    // dilated like the layer's
    uint32_t n_gated; float alo[3], ahi[3];
};

// What the kernel sees.  All pointers are device (or, in the CPU emulator, host) addresses.
struct DScene {
    const DItem *items; uint32_t n_items; uint32_t n_world_items;   // world range = items[0, n_world_items); instance ranges follow
    const DSphere *spheres; const uint32_t *sphere_mat; uint32_t n_spheres;
    const DMoving *moving;
    const DRect *rects;
    const DBox *boxes; uint32_t n_boxes;
    const DList *lists; const uint32_t *list_refs;
    const DMedium *media;
    const DInstance *instances;
    const DMaterial *materials;
    const DMaterial *sphere_material;   // materials[sphere_mat[i]] for every sphere i (read by the sphere-only kernel variants)
    const DTexture *textures;
    const DImage *images; const uint8_t *image_bytes;
    const DPerlin *perlins;
    const uint32_t *lights; uint32_t n_lights;
    uint32_t features;
    // Tie table: null unless the lineariser rebuilt a draw-free subtree (vk_linearize.cpp).  Inside such a block objects are
    // visited in another order than the reference's, which is unobservable except when two of them are hit at EXACTLY the same
    // t; then the reference's choice is reproduced from their positions in ITS visiting order: entry = block << 20 | position,
    // indexed by object id = sphere index | tie_base_rect + rect | tie_base_box + box | tie_base_list + list.
    const uint32_t *tie_rank; uint32_t tie_base_rect, tie_base_box, tie_base_list;
    // The spheres whose material reads a Perlin noise texture (vk_kernels.h cooperative_turb finds the lanes that will evaluate one
    // by comparing their hit with this list instead of chasing sphere -> material -> texture): sphere index, texture, perlin table.
    // n_noise_spheres = ~0: more than fit here, look the material up.
    uint32_t n_noise_spheres; uint32_t noise_sphere[4], noise_tex[4], noise_perlin[4];
    // Exact re-treeing (vk_linearize.cpp rt_collect, vk_trace.h segment_unsafe): t_pad > 0 says that items[0, n_world_items) is a tree
    // REBUILT over the reference's leaf units, walked with the closest hit so far padded by (1 + t_pad); ref_items[0, n_ref_items) is
    // the tree as handed over, on which the rare sample whose result may depend on the visiting order is rendered again.
    // gate_scale = 1 / (1 + t_pad) and tmin_gate = T_MIN * gate_scale (rounded down); 1 and T_MIN when t_pad == 0.
    const DItem *ref_items; uint32_t n_ref_items; float t_pad, gate_scale, tmin_gate;
    // ... and the trusted origin ball: the gates of the rebuilt tree are proven sound for rays that start inside it (vk_linearize.h
    // rt_unit_growth); a segment that starts outside is decided on the tree as handed over (vk_trace.h segment_unsafe)
    float trust_c0[3], trust_r0sq;
    // every sphere's centre coordinates and radius are below 2^30 in magnitude: the sphere tests of the sphere-only kernel variants may
    // divide by |d|^2 through a shared reciprocal (vk_trace.h div_by_a)
    uint32_t fast_div;
    // Scenes traversed from global memory keep BOTH trees in items[]: [the tree as handed over | a sentinel no ray passes | the
    // rebuilt tree], walk_start = index of the rebuilt tree's first item (0: items[] is one tree).  A segment whose winner is early
    // is then walked again right away, from item 0, instead of its sample being queued (vk_trace.h begin_segment).
    uint32_t walk_start;
    // The near form of exact re-treeing (vk_linearize.cpp rt_grow_near): a segment walked on the rebuilt tree stands only if its hit lies
    // within `reach` of its origin (T |d| <= reach; a miss never does): then every sphere that could hold a closer candidate is within
    // the radius its own-box gate is sound for.  0 = no such condition (the unit form, whose gates are sound for the whole ball).
    float reach;
    // ... or if, beyond `reach`, the ray runs CLEAR of every small sphere.  A far sphere's candidate point lies within
    // delta = sqrt(32 u) (rho + R) of the sphere, hence of the box [small_clo, small_chi] around the small spheres' SURFACES, and no
    // farther from the origin than that box's far corner: delta <= M := clear_k (D_far(o) + clear_r2) + clear_slack.  If the ray is
    // outside that box grown by M for every s >= reach (a slab test), no sphere beyond rho_near holds a candidate at all, and a MISS,
    // or a far hit on a big sphere, stands too (vk_trace.h clear_of_small_spheres; docs/gate_lemma.md section 7).
    float small_clo[3], small_chi[3], clear_k, clear_r2, clear_slack;
    // ... and, decided per frame by the host: primary rays (depth 1) start on the tree as handed over (the camera is farther than
    // `reach` from everything, so their rebuilt walk could never stand)
    uint32_t primary_ref;
    // The grid form (DGrid above): cells[c] .. cells[c + 1] is cell c's range in refs[] (sphere references)
    DGrid grid; const uint32_t *grid_cells; const uint32_t *grid_refs;
    // exact re-treeing: unit_item[sphere] = the item of the tree as handed over whose box gates the sphere there (its leaf); null: none.
    // unit_tree = that tree's items (ref_items, or items[] when both trees share it: DScene::walk_start)
    const uint32_t *unit_item; const DItem *unit_tree;
};

// A view that walks the grid (grid.nu != 0) instead of the rebuilt tree: the grid form is sound for every ordinary ray wherever it
// starts, so the tree forms' conditions — the trusted ball, the near form's reach — do not apply (vk_trace.h segment_unsafe keeps the
// safe-winner test).  A view that walks the TREE of a world that also has a grid must keep them: drop_grid().
inline void use_grid(DScene &s) { s.trust_r0sq = __builtin_inff(); s.reach = 0.0f; s.primary_ref = 0u; }
inline void drop_grid(DScene &s) { s.grid.nu = 0u; s.grid.nv = 0u; }

}  // namespace vkd
#endif
