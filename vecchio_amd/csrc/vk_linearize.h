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
    //      over riding along for the samples that need it (vk_trace.h winner_is_early): results are the reference's
    //  -1  as vk_scene_desc.flags says: VK_SCENE_FAST_ACCEL -> 1, VK_SCENE_REFERENCE_TREE -> 0, else 2
    int retree = -1;
};

// returns VK_OK or an error code with `err` set
int linearize(const vk_scene_desc *desc, LinearScene &out, std::string &err, const LinearizeOptions &opt = LinearizeOptions());

}  // namespace vkd
#endif
