// vk_linearize.h — host-side conversion of the vk_scene_desc graph into the linear device
// scene of vk_device_scene.h (threaded pre-order BVH items, SoA primitives, flattened
// transform chains, baked material records).  Pure C++ (no HIP): also used by the CPU
// emulator under tests/emu.
#ifndef VK_LINEARIZE_H
#define VK_LINEARIZE_H
#include <string>
#include <vector>

#include "../../include/vecchio_amd.h"
#include "vk_device_scene.h"

namespace vkd {

struct LinearScene {
    std::vector<DItem> items;
    std::vector<DSphere> spheres;
    std::vector<uint32_t> sphere_mat;
    std::vector<DMoving> moving;
    std::vector<DRect> rects;
    std::vector<DBox> boxes;
    std::vector<DList> lists;
    std::vector<uint32_t> list_refs;
    std::vector<DMedium> media;
    std::vector<DInstance> instances;
    std::vector<DMaterial> materials;
    std::vector<DMaterial> sphere_material;   // see DScene (filled for scenes of spheres only)
    std::vector<DTexture> textures;
    std::vector<DImage> images;
    std::vector<uint8_t> image_bytes;
    std::vector<DPerlin> perlins;
    // see DScene
    uint32_t n_noise_spheres = 0; uint32_t noise_sphere[4] = {0, 0, 0, 0}, noise_tex[4] = {0, 0, 0, 0}, noise_perlin[4] = {0, 0, 0, 0};
    std::vector<uint32_t> lights;
    // tie table (empty when no subtree was rebuilt): per object id ([spheres][rects][boxes][lists]) its block (bits 31..20, 0 = not
    // in a rebuilt subtree) and its position in the reference's visiting order inside the block (bits 19..0)
    std::vector<uint32_t> tie_rank;
    uint32_t tie_base_rect = 0, tie_base_box = 0, tie_base_list = 0;
    std::vector<DItem> ref_items; float t_pad = 0.0f;      // see DScene
    std::vector<uint32_t> unit_item;                      // see DScene (one entry per sphere when ref_items is there)
    float trust_c0[3] = {0.0f, 0.0f, 0.0f}; float trust_r0 = 0.0f;      // exact re-treeing: the trusted origin ball (DScene)
    bool proven = false;        // exact re-treeing with grown gates: the gate lemma applies (vk_linearize.cpp rt_grow_units / rt_grow_near)
    // The NEAR form of exact re-treeing (vk_linearize.cpp rt_grow_near): every sphere gated by its OWN box, sound for origins within
    // near_radius of it; a segment's result stands only if its hit lies within `reach` of its origin (DScene::reach).  Needs both trees in
    // items[] (a scene traversed from global memory): vk_api.hip keeps such a scene out of LDS.
    bool near_form = false;
    bool near_spans = false;      // ... whose reach spans the small spheres' whole box: staged in LDS where it fits (vk_api.hip plan_residency)
    float reach = 0.0f, near_radius = 0.0f;
    // for the per-frame "primary rays straight to the tree as handed over" decision (vk_api.hip): the box around the small spheres and
    // the always-sound big ones (centre, radius), at most 8 (more: n_big = ~0 and the decision is "no")
    float small_lo[3] = {0, 0, 0}, small_hi[3] = {0, 0, 0};     // (box around the small spheres' surfaces)
    float clear_k = 0.0f, clear_r2 = 0.0f, clear_slack = 0.0f;      // DScene: the clearance test's constants
    uint32_t n_big = 0; float big[8][4] = {};
    // the GRID form (DGrid): grid.nu != 0 when the world is eligible (vk_linearize.cpp rt_build_grid)
    DGrid grid = {};
    std::vector<uint32_t> grid_cells, grid_refs;
    uint32_t features = 0;
    uint32_t n_prims = 0;
    uint32_t world_items = 0;   // items[0, world_items) is the world BVH; instance child ranges follow

    DScene host_view() const;   // DScene whose pointers address these vectors
    // exact re-treeing, for a scene traversed from global memory: [ref_items | sentinel | items] in one array (see DScene::walk_start)
    std::vector<DItem> combined_items(uint32_t &walk_start) const;
};

struct LinearizeOptions {
    // What may be rebuilt (vk_linearize.cpp; diagnostic switch VK_RETREE):
    //   0  nothing: the tree as handed over, everywhere
    //   1  every draw-free subtree, object by object (VK_SCENE_FAST_ACCEL: results may differ where a hit lies a rounding error
    //      outside its box)
    //   2  EXACT re-treeing: the world tree of a scene of spheres only, over the reference's leaf units, with the tree as handed
    //      over riding along for the samples that need it (vk_trace.h segment_unsafe): results are the reference's
    //  -1  as vk_scene_desc.flags says: VK_SCENE_FAST_ACCEL -> 1, VK_SCENE_REFERENCE_TREE -> 0, else 2
    int retree = -1;
    // test switches of exact re-treeing (environment VK_GATE_GROW=0, VK_T_PAD=x: read by tests/emu and by the DEBUG build of the library
    // only, never by the product): the unit boxes as handed over instead of grown ones, another relative padding of the gate
    // (0 = RT_PAD).  Neither is sound: a tree built with either is never reported as proven and takes allow_empirical.  They exist so
    // that the counter-examples of the gate lemma can be shown to bite.
    bool gate_grow = true;
    float t_pad = 0.0f;
    bool want_proof = true;     // VK_GATE_PROOF=0: the empirical form even where the proven one is cheap (comparisons; an unproven tree
                                // still takes VK_SCENE_EMPIRICAL_TREES in the description)
    bool near_form = true;      // VK_NEAR_FORM=0 (emulator / debug library): no near form where the unit form is not eligible
    bool unit_form = true;      // VK_UNIT_FORM=0 (emulator / debug library): the near form even where the unit form is eligible (comparisons)
    bool grid_form = true;      // VK_GRID_FORM=0 (emulator / debug library): no grid form (comparisons)
    bool near_first = true;     // VK_NEAR_FIRST=0 (emulator / debug library): the unit form where both are eligible (comparisons)
    bool allow_empirical = false;   // tests/emu and the debug library, VK_EMPIRICAL_TREES=1: as vk_scene_desc.flags & VK_SCENE_EMPIRICAL_TREES
};

// ---- exact re-treeing: the arithmetic behind the soundness of the rebuilt tree's gates ("Gate lemma").
// Sphere::hit in f32 (hittable.rs:65-95) reports hit points that need not lie on the sphere: with u = 2^-24 and rho = |o - c|, the point
// o + t d of an accepted root t lies within rt_eta(rho, R) of the sphere's surface (forward error analysis: K <= 30; measured <= 7;
// RT_KAPPA = 32 u).  A unit's gate box is the reference's box grown by rt_unit_growth(): then, for every ray origin inside the trusted
// ball `dom`, the gate passes whenever the reference could have accepted one of the unit's spheres — because the hit point lies inside
// the grown box (near origins) or because it precedes the box entry by less than the gate's relative padding RT_PAD (far origins).
// tests/test_gate_lemma.py hammers both statements with the kernel's own arithmetic.
struct RtDomain { double c0[3]; double r0; };
constexpr double RT_KAPPA = 1.0 / 524288.0;       // 2^-19 = 32 * 2^-24
constexpr double RT_PAD = 1.0 / 4.0;              // LinearScene::t_pad of exact re-treeing with grown gates (the proven form)
constexpr double RT_PAD_EMPIRICAL = 1.0 / 16.0;   // ... with the units' boxes as handed over (the empirical form)
// the near form: a small sphere's own box grows by RT_NEAR_GROWTH of the median small radius (which fixes the radius within which its
// gate is sound: eta(near_radius) = 0.8 RT_NEAR_GROWTH R, near_radius ~ 144 R); a sphere whose gate can be sound for EVERY origin of the
// trusted ball at RT_NEAR_BIG_GROWTH of its radius is "big" and never restricts a segment; the ball reaches RT_NEAR_BALL extents
constexpr double RT_PAD_NEAR = 1.0 / 64.0;
constexpr double RT_NEAR_GROWTH = 0.05;
constexpr double RT_NEAR_BIG_GROWTH = 0.01;
constexpr double RT_NEAR_BALL = 4.0;
constexpr double RT_MAX_AREA_GROWTH = 0.05;       // the grown gates are used when they cost at most this much leaf surface area
constexpr double RT_LONG_GROWTH = 0.25;           // a unit whose gate would grow by more than this (in units of its smaller radius) is "long"
inline double rt_eta(double rho, double R) { return RT_KAPPA * (rho + R) * (rho + R) / R; }
// growth (>= 0) of the box [umn, umx] gating the n spheres (centers, radii), or < 0 if no growth makes the gate sound for origins
// as far away as the ball allows (the caller shrinks the ball)
double rt_unit_growth(const float umn[3], const float umx[3], int n, const float (*centers)[3], const float *radii, const RtDomain &dom,
    double t_pad = RT_PAD, bool box_grows = true);

// returns VK_OK or an error code with `err` set
int linearize(const vk_scene_desc *desc, LinearScene &out, std::string &err, const LinearizeOptions &opt = LinearizeOptions());

}  // namespace vkd
#endif
