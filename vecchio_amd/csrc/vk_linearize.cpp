// vk_linearize.cpp — see vk_linearize.h
#include "vk_linearize.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <unordered_map>

namespace vkd {

DScene LinearScene::host_view() const {
    DScene s;
    memset(&s, 0, sizeof(s));
    s.items = items.data(); s.n_items = (uint32_t)items.size(); s.n_world_items = world_items;
    s.spheres = spheres.data(); s.sphere_mat = sphere_mat.data(); s.n_spheres = (uint32_t)spheres.size();
    s.moving = moving.data(); s.rects = rects.data();
    s.boxes = boxes.data(); s.n_boxes = (uint32_t)boxes.size();
    s.lists = lists.data(); s.list_refs = list_refs.data();
    s.media = media.data(); s.instances = instances.data();
    s.materials = materials.data(); s.textures = textures.data();
    s.sphere_material = sphere_material.empty() ? nullptr : sphere_material.data();
    s.images = images.data(); s.image_bytes = image_bytes.data();
    s.perlins = perlins.data();
    s.lights = lights.data(); s.n_lights = (uint32_t)lights.size();
    s.features = features;
    s.tie_rank = tie_rank.empty() ? nullptr : tie_rank.data();
    s.tie_base_rect = tie_base_rect; s.tie_base_box = tie_base_box; s.tie_base_list = tie_base_list;
    s.n_noise_spheres = n_noise_spheres;
    for (int k = 0; k < 4; k++) { s.noise_sphere[k] = noise_sphere[k]; s.noise_tex[k] = noise_tex[k]; s.noise_perlin[k] = noise_perlin[k]; }
    s.ref_items = ref_items.empty() ? nullptr : ref_items.data(); s.n_ref_items = (uint32_t)ref_items.size();
    s.t_pad = ref_items.empty() ? 0.0f : t_pad;
    s.gate_scale = 1.0f / (1.0f + s.t_pad);
    s.tmin_gate = s.t_pad > 0.0f ? 0.001f * s.gate_scale * 0.999999f : 0.001f;     // (T_MIN of vk_trace.h)
    for (int k = 0; k < 3; k++) s.trust_c0[k] = trust_c0[k];
    s.trust_r0sq = trust_r0 * trust_r0;
    s.reach = ref_items.empty() ? 0.0f : reach; s.primary_ref = 0u;
    for (int k = 0; k < 3; k++) { s.small_clo[k] = small_lo[k]; s.small_chi[k] = small_hi[k]; }
    s.clear_k = clear_k; s.clear_r2 = clear_r2; s.clear_slack = clear_slack;
    // (only where Lemma 1 is PROVEN: in the empirical form the stricter own-box test is part of what keeps it right in practice)
    s.unit_item = (ref_items.empty() || unit_item.empty() || !proven) ? nullptr : unit_item.data(); s.unit_tree = ref_items.data();
    // (the grid form, where the world is eligible: whoever walks it — use_grid() — lifts the tree forms' conditions)
    if (!ref_items.empty() && grid.nu != 0u) { s.grid = grid; s.grid_cells = grid_cells.data(); s.grid_refs = grid_refs.data(); }
    s.fast_div = 1u;
    for (const DSphere &sp : spheres)
        if (!(fabsf(sp.cx) < 1073741824.0f && fabsf(sp.cy) < 1073741824.0f && fabsf(sp.cz) < 1073741824.0f && fabsf(sp.r) < 1073741824.0f)) s.fast_div = 0u;
    return s;
}

std::vector<DItem> LinearScene::combined_items(uint32_t &walk_start) const {
    const uint32_t n_ref = (uint32_t)ref_items.size(), n_new = (uint32_t)items.size(), total = n_ref + 1u + n_new;
    std::vector<DItem> out;
    out.reserve(total);
    for (DItem it : ref_items) {                      // the tree as handed over: its exits lead past everything
        if ((it.w0 >> 28) == 0u && it.w0 >= n_ref) it.w0 = total;
        out.push_back(it);
    }
    DItem stop; memset(&stop, 0, sizeof(stop));       // after its last leaf: an inner item no ray passes (an x slab at +infinity)
    stop.mnx = INFINITY; stop.mxx = INFINITY; stop.mny = -3.0e38f; stop.mxy = 3.0e38f; stop.mnz = -3.0e38f; stop.mxz = 3.0e38f;
    stop.w0 = total; stop.w1 = 0;
    out.push_back(stop);
    walk_start = n_ref + 1u;
    for (DItem it : items) {                          // the rebuilt tree, moved up
        if ((it.w0 >> 28) == 0u) it.w0 += walk_start;
        out.push_back(it);
    }
    return out;
}

double rt_unit_growth(const float umn[3], const float umx[3], int n, const float (*centers)[3], const float *radii, const RtDomain &dom,
    double t_pad, bool box_grows) {
    const double U24 = 1.0 / 16777216.0;
    const double pad = t_pad * (1.0 - 1.0 / 1024.0);       // (what the f32 gate arithmetic leaves of it, generously)
    // beyond rho1 = far (d* + 1.07 (R + g)) the padding covers a hit point outside the grown box: pad (rho1 - R - eta) >= d* + R + eta
    // given g >= eta(rho1), because pad * far = 1 + pad and 1.07 (1 + pad) - pad >= 1 + pad / 2
    const double far = 1.0 / pad + 1.0;
    double maxabs = 0.0;
    for (int a = 0; a < 3; a++) maxabs = std::max(maxabs, std::max(std::fabs((double)umn[a]), std::fabs((double)umx[a])));
    double g = 0.0;
    // farthest point of the box grown by g from a sphere's centre
    auto dstar = [&](const float *c, double gg) {
        double s2 = 0.0;
        if (!box_grows) gg = 8.0 * U24 * maxabs;       // the box stays (but for its rounding slack): the growth is the spheres' own boxes'
        for (int a = 0; a < 3; a++) {
            const double d = std::max(std::fabs((double)umn[a] - gg - c[a]), std::fabs((double)umx[a] + gg - c[a]));
            s2 += d * d;
        }
        return std::sqrt(s2);
    };
    auto rho_dom = [&](const float *c) {
        double s2 = 0.0;
        for (int a = 0; a < 3; a++) s2 += ((double)c[a] - dom.c0[a]) * ((double)c[a] - dom.c0[a]);
        return std::sqrt(s2) + dom.r0;
    };
    // Near origins (rho <= rho1): the hit point lies within eta(rho) of the sphere, hence inside the box grown by g >= 1.25 eta(rho1)
    // (the quarter is the margin the slab test's own rounding needs).  rho1 depends on g through the grown box: smallest fixed point.
    for (int iter = 0; iter < 64; iter++) {
        double gn = 0.0;
        for (int k = 0; k < n; k++) {
            const double R = radii[k];
            const double rho1 = far * (dstar(centers[k], g) + 1.07 * (R + g));
            gn = std::max(gn, 1.25 * rt_eta(std::min(rho1, rho_dom(centers[k])), R));
        }
        if (gn <= g) break;
        g = gn * (1.0 + 1e-9);
        if (iter == 63) {         // no fixed point below the ball's cap: the cap itself (every origin of the ball is a near one)
            g = 0.0;
            for (int k = 0; k < n; k++) g = std::max(g, 1.25 * rt_eta(rho_dom(centers[k]), radii[k]));
        }
    }
    // Far origins (rho1 < rho <= rho_dom): the hit point may lie outside the grown box, but then precedes the ray's entry into it by at
    // most d* + R + eta(rho), which the padding covers when pad (rho - R - eta) >= d* + R + eta.  The difference is concave in rho and
    // non-negative at rho1 (by the choice of rho1), so it is enough to check the far end.
    for (int k = 0; k < n; k++) {
        const double R = radii[k], rd = rho_dom(centers[k]);
        const double ds = dstar(centers[k], g), rho1 = far * (ds + 1.07 * (R + g));
        if (rd <= rho1) continue;
        const double eta = rt_eta(rd, R);
        if (pad * (rd - R - eta) < ds + R + eta) return -1.0;
    }
    // the f32 roundings of the box bounds and of center -+ radius (Sphere::bounding_box): a few ulps of the largest coordinate
    return g + 8.0 * U24 * maxabs;
}

namespace {

struct Builder {
    const vk_scene_desc *d;
    LinearScene &L;
    std::string &err;
    int status = VK_OK;
    std::map<uint32_t, uint32_t> list_memo, medium_memo, box_memo;
    struct Pending { uint32_t bvh_index; uint32_t flip; int32_t inst; };
    std::deque<Pending> pending;   // BVH children of instances, emitted after the current range

    Builder(const vk_scene_desc *d_, LinearScene &l, std::string &e) : d(d_), L(l), err(e) {}

    bool fail(int code, const std::string &m) { if (status == VK_OK) { status = code; err = m; } return false; }

    bool check_ref(vk_ref r) {
        uint32_t k = VK_REF_KIND(r), i = VK_REF_INDEX(r);
        switch (k) {
            case VK_KIND_BVH: return i < d->n_bvh || fail(VK_ERR_BAD_ARG, "bvh index out of range");
            case VK_KIND_SPHERE: return i < d->n_spheres || fail(VK_ERR_BAD_ARG, "sphere index out of range");
            case VK_KIND_MOVING_SPHERE: return i < d->n_moving_spheres || fail(VK_ERR_BAD_ARG, "moving sphere index out of range");
            case VK_KIND_RECT: return i < d->n_rects || fail(VK_ERR_BAD_ARG, "rect index out of range");
            case VK_KIND_LIST: return i < d->n_lists || fail(VK_ERR_BAD_ARG, "list index out of range");
            case VK_KIND_MEDIUM: return i < d->n_media || fail(VK_ERR_BAD_ARG, "medium index out of range");
            case VK_KIND_TRANSLATE: return i < d->n_translates || fail(VK_ERR_BAD_ARG, "translate index out of range");
            case VK_KIND_ROTATE: return i < d->n_rotates || fail(VK_ERR_BAD_ARG, "rotate index out of range");
            default: return fail(VK_ERR_BAD_ARG, "unknown hittable kind in reference");
        }
    }

    static bool is_simple(uint32_t kind) { return kind == VK_KIND_SPHERE || kind == VK_KIND_MOVING_SPHERE || kind == VK_KIND_RECT; }

    uint32_t simple_dref(vk_ref r, uint32_t flip) {
        uint32_t k = VK_REF_KIND(r), i = VK_REF_INDEX(r);
        uint32_t f = ((r & VK_REF_FLIP) ? DREF_FLIP : 0u) ^ flip;
        uint32_t dk = k == VK_KIND_SPHERE ? DK_SPHERE : (k == VK_KIND_MOVING_SPHERE ? DK_MOVING : DK_RECT);
        if (k == VK_KIND_MOVING_SPHERE) L.features |= VKF_MOVING;
        if (k == VK_KIND_RECT) L.features |= VKF_RECT;
        return VKD_MAKE(dk, i) | f;
    }

    // list of primitives (Boxy::sides); flip of the list reference is applied by the caller
    bool convert_list(uint32_t idx, uint32_t &out) {
        auto it = list_memo.find(idx);
        if (it != list_memo.end()) { out = it->second; return true; }
        const vk_list &l = d->lists[idx];
        if ((uint64_t)l.first + l.count > d->n_list_items) return fail(VK_ERR_BAD_ARG, "list items out of range");
        DList dl; dl.first = (uint32_t)L.list_refs.size(); dl.count = l.count;
        for (uint32_t j = 0; j < l.count; j++) {
            vk_ref r = d->list_items[l.first + j];
            if (!check_ref(r)) return false;
            if (!is_simple(VK_REF_KIND(r)))
                return fail(VK_ERR_UNSUPPORTED, "device path: list items must be Sphere/MovingSphere/Rect (optionally FlipFace-wrapped)");
            L.list_refs.push_back(simple_dref(r, 0));
        }
        L.lists.push_back(dl);
        L.features |= VKF_LIST;
        out = (uint32_t)L.lists.size() - 1;
        list_memo[idx] = out;
        return true;
    }

    // Is list `idx` exactly Boxy::new(p0, p1, mat) (hittable.rs:321-359)?  Then store it as a DBox.
    bool as_box(uint32_t idx, uint32_t &out) {
        auto it = box_memo.find(idx);
        if (it != box_memo.end()) { out = it->second; return out != 0xFFFFFFFFu; }
        box_memo[idx] = 0xFFFFFFFFu;
        const vk_list &l = d->lists[idx];
        if (l.count != 6 || (uint64_t)l.first + 6 > d->n_list_items) return false;
        const vk_rect *q[6];
        static const uint8_t ax[6][3] = {{0, 1, 2}, {0, 1, 2}, {0, 2, 1}, {0, 2, 1}, {1, 2, 0}, {1, 2, 0}};
        for (int f = 0; f < 6; f++) {
            vk_ref r = d->list_items[l.first + f];
            if (VK_REF_KIND(r) != VK_KIND_RECT || VK_REF_INDEX(r) >= d->n_rects) return false;
            if (((r & VK_REF_FLIP) != 0) != ((f & 1) != 0)) return false;
            q[f] = &d->rects[VK_REF_INDEX(r)];
            if (q[f]->axis0 != ax[f][0] || q[f]->axis1 != ax[f][1] || q[f]->axis2 != ax[f][2]) return false;
            if (q[f]->material != q[0]->material) return false;
        }
        float p0x = q[0]->c0, p1x = q[0]->c1, p0y = q[0]->d0, p1y = q[0]->d1, p1z = q[0]->k, p0z = q[1]->k;
        auto same = [](float a, float b) { return memcmp(&a, &b, 4) == 0; };
        bool ok = same(q[1]->c0, p0x) && same(q[1]->c1, p1x) && same(q[1]->d0, p0y) && same(q[1]->d1, p1y) &&
                  same(q[2]->c0, p0x) && same(q[2]->c1, p1x) && same(q[2]->d0, p0z) && same(q[2]->d1, p1z) && same(q[2]->k, p1y) &&
                  same(q[3]->c0, p0x) && same(q[3]->c1, p1x) && same(q[3]->d0, p0z) && same(q[3]->d1, p1z) && same(q[3]->k, p0y) &&
                  same(q[4]->c0, p0y) && same(q[4]->c1, p1y) && same(q[4]->d0, p0z) && same(q[4]->d1, p1z) && same(q[4]->k, p1x) &&
                  same(q[5]->c0, p0y) && same(q[5]->c1, p1y) && same(q[5]->d0, p0z) && same(q[5]->d1, p1z) && same(q[5]->k, p0x);
        if (!ok || q[0]->material >= d->n_materials) return false;
        DBox b; memset(&b, 0, sizeof(b));
        b.p0[0] = p0x; b.p0[1] = p0y; b.p0[2] = p0z; b.p1x = p1x; b.p1y = p1y; b.p1z = p1z; b.mat = q[0]->material;
        L.boxes.push_back(b);
        L.features |= VKF_BOX;
        out = (uint32_t)L.boxes.size() - 1;
        box_memo[idx] = out;
        return true;
    }

    bool convert_medium(uint32_t idx, uint32_t &out) {
        auto it = medium_memo.find(idx);
        if (it != medium_memo.end()) { out = it->second; return true; }
        const vk_medium &m = d->media[idx];
        if (!check_ref(m.boundary)) return false;
        if (m.material >= d->n_materials) return fail(VK_ERR_BAD_ARG, "material index out of range");
        uint32_t bk = VK_REF_KIND(m.boundary);
        DMedium dm; memset(&dm, 0, sizeof(dm));
        if (is_simple(bk)) dm.boundary = simple_dref(m.boundary, 0);
        else if (bk == VK_KIND_LIST) {
            uint32_t li;
            if (!convert_list(VK_REF_INDEX(m.boundary), li)) return false;
            dm.boundary = VKD_MAKE(DK_LIST, li) | ((m.boundary & VK_REF_FLIP) ? DREF_FLIP : 0u);
        } else return fail(VK_ERR_UNSUPPORTED, "device path: ConstantMedium boundary must be a Sphere, MovingSphere, Rect or Boxy");
        dm.neg_inv_density = m.neg_inv_density; dm.mat = m.material;
        L.media.push_back(dm);
        L.features |= VKF_MEDIUM;
        out = (uint32_t)L.media.size() - 1;
        medium_memo[idx] = out;
        return true;
    }

    // Translate / Rotate* chain -> one DInstance (tree-unique: a new record per occurrence)
    bool convert_instance(vk_ref r, uint32_t flip, int32_t parent_inst, uint32_t &out_dref, int nest) {
        if (nest > 64) return fail(VK_ERR_BAD_ARG, "transform chain too deep / cyclic");
        DInstance I; memset(&I, 0, sizeof(I));
        I.parent = parent_inst;
        I.flip = ((r & VK_REF_FLIP) ? DREF_FLIP : 0u) ^ flip;
        uint32_t pdepth = 0;
        if (parent_inst >= 0) {
            const DInstance &P = L.instances[parent_inst];
            pdepth = P.depth + 1;
            if (pdepth >= (uint32_t)MAX_DEPTH_INST) return fail(VK_ERR_UNSUPPORTED, "device path: instance nesting deeper than 4");
            for (uint32_t l = 0; l <= P.depth; l++) I.chain[l] = P.chain[l];
        }
        I.depth = pdepth;
        vk_ref cur = r;
        while (I.n_ops < (uint32_t)MAX_OPS) {
            uint32_t k = VK_REF_KIND(cur), i = VK_REF_INDEX(cur);
            if (k == VK_KIND_TRANSLATE) {
                const vk_translate &t = d->translates[i];
                DOp op; op.kind = OP_TRANSLATE; op.a = t.offset[0]; op.b = t.offset[1]; op.c = t.offset[2];
                I.ops[I.n_ops++] = op;
                cur = t.child;
            } else if (k == VK_KIND_ROTATE) {
                const vk_rotate &t = d->rotates[i];
                if (t.axis > 2) return fail(VK_ERR_BAD_ARG, "rotate axis out of range");
                DOp op; op.kind = t.axis == 0 ? OP_ROTATE_X : (t.axis == 1 ? OP_ROTATE_Y : OP_ROTATE_Z); op.a = t.sin_theta;
                op.b = t.cos_theta; op.c = 0.0f;
                I.ops[I.n_ops++] = op;
                cur = t.child;
            } else break;
            if (!check_ref(cur)) return false;
            // a FlipFace between two wrappers only negates `front`, which the outer wrapper's
            // set_face_normal overwrites (hittable.rs:519,618): it has no effect and is dropped
        }
        int32_t self = (int32_t)L.instances.size();
        I.chain[I.depth] = self;
        L.instances.push_back(I);
        L.features |= VKF_INSTANCE;
        uint32_t ck = VK_REF_KIND(cur);
        // the child's own FlipFace parity is likewise overwritten by this wrapper: dropped
        if (ck == VK_KIND_BVH) {
            pending.push_back(Pending{VK_REF_INDEX(cur), 0u, self});
        } else {
            uint32_t cref;
            if (!convert_object(cur & ~VK_REF_FLIP, 0, self, cref, nest + 1)) return false;
            L.instances[self].child_ref = cref;
        }
        out_dref = VKD_MAKE(DK_INSTANCE, self);
        return true;
    }

    // any non-BVH hittable -> dref
    bool convert_object(vk_ref r, uint32_t flip, int32_t inst, uint32_t &out, int nest = 0) {
        if (!check_ref(r)) return false;
        uint32_t k = VK_REF_KIND(r), i = VK_REF_INDEX(r);
        uint32_t f = ((r & VK_REF_FLIP) ? DREF_FLIP : 0u) ^ flip;
        if (is_simple(k)) { out = simple_dref(r, flip); return true; }
        if (k == VK_KIND_LIST) {
            uint32_t bi;
            if (as_box(i, bi)) { out = VKD_MAKE(DK_BOX, bi) | f; return true; }
            uint32_t li; if (!convert_list(i, li)) return false; out = VKD_MAKE(DK_LIST, li) | f; return true;
        }
        if (k == VK_KIND_MEDIUM) { uint32_t mi; if (!convert_medium(i, mi)) return false; out = VKD_MAKE(DK_MEDIUM, mi) | f; return true; }
        if (k == VK_KIND_TRANSLATE || k == VK_KIND_ROTATE) return convert_instance(r, flip, inst, out, nest);
        return fail(VK_ERR_UNSUPPORTED, "device path: a BVHNode may only appear as the world, as a BVH child or under Translate/Rotate");
    }

    void set_home(uint32_t dref, uint32_t home_next, uint32_t home_pend) {
        if (VKD_KIND(dref) == DK_INSTANCE) {
            DInstance &I = L.instances[VKD_INDEX(dref)];
            I.home_next = home_next; I.home_pend = home_pend;
        }
    }

    // an object that is a direct BVH child beside a BVH sibling has no box test of its own in the
    // reference: a leaf whose box every ray hits (decided on the fast path, no special case in the kernel)
    static DItem always_hit_leaf() {
        DItem it; memset(&it, 0, sizeof(it));
        it.mnx = it.mny = it.mnz = -3.0e38f; it.mxx = it.mxy = it.mxz = 3.0e38f;
        return it;
    }

    // A `len == 1` node (accel.rs:108-111: both children the same Arc) calls its object twice, the second time with tmax = the first
    // call's hit.  When is the second call unobservable?
    //  * ONE object in which nothing draws (Sphere, Rect, list, Boxy; also under a Translate / Rotate chain, which passes tmin and tmax
    //    through, hittable.rs:507-524,578-624): the second call returns None (Sphere::hit and the list scan are strict in tmax,
    //    hittable.rs:75,386) or the SAME object at the same t (Rect::hit accepts t == tmax, hittable.rs:232): the node's result is the
    //    first call's either way.
    //  * a BVHNode (directly or under such a chain) over SPHERES only: every box the second call passes it passed in the first call
    //    (AxisBB::hit is monotone in tmax, and the second call's tmax is the first call's final one), so every sphere it tests was tested
    //    then with a larger tmax, and a root below the final t would have been accepted: None.
    //    (The final scene's 1 000-sphere cluster sits in such a node: 213 -> 173 steps per sample, C3 +17 %.)
    //  * a BVHNode over Rects, lists or Boxys is NOT covered: two coplanar Rects tie at t0, the first call ends with the later one, and
    //    the second call — whose boxes are tested against tmax = t0 — can lose exactly that one to a box whose entry rounds to t0 and
    //    return the other: same t, another material.  Such a node is entered twice, as the reference does (round 4 skipped it: ADVICE
    //    r4; tests/test_dup_instance.py has the coplanar case).
    std::vector<int8_t> spheres_only_memo;      // per vk_bvh_node: 1 / 0, -1 unknown
    bool spheres_only_subtree(uint32_t root, bool &yes) {
        if (spheres_only_memo.empty()) spheres_only_memo.assign(d->n_bvh, -1);
        std::vector<uint32_t> st{root}, seen;
        yes = true;
        size_t guard = 0;
        while (!st.empty() && yes) {
            if (++guard > (size_t)8 * (d->n_bvh + 16) + 1024) return fail(VK_ERR_BAD_ARG, "BVH graph is cyclic");
            const uint32_t ni = st.back(); st.pop_back();
            if (spheres_only_memo[ni] == 1) continue;
            if (spheres_only_memo[ni] == 0) { yes = false; break; }
            seen.push_back(ni);
            const vk_bvh_node &n = d->bvh[ni];
            for (vk_ref c : {n.left, n.right}) {
                if (!check_ref(c)) return false;
                if (VK_REF_KIND(c) == VK_KIND_BVH) st.push_back(VK_REF_INDEX(c));
                else if (VK_REF_KIND(c) != VK_KIND_SPHERE) yes = false;
            }
        }
        if (yes) for (uint32_t ni : seen) spheres_only_memo[ni] = 1;
        else spheres_only_memo[root] = 0;
        return true;
    }
    bool draw_free_instance(uint32_t idx, bool &yes) {
        yes = false;
        const DInstance &I = L.instances[idx];
        if (I.child_ref != 0u) {
            if (VKD_KIND(I.child_ref) == DK_INSTANCE) return draw_free_instance(VKD_INDEX(I.child_ref), yes);
            yes = draw_free(I.child_ref);
            return true;
        }
        for (const Pending &p : pending)
            if (p.inst == (int32_t)idx) return spheres_only_subtree(p.bvh_index, yes);
        return true;
    }
    static bool draw_free(uint32_t dref) { uint32_t k = VKD_KIND(dref);
        return k == DK_SPHERE || k == DK_MOVING || k == DK_RECT || k == DK_LIST || k == DK_BOX; }

    // ------------------------------------------------------------------ re-treeing of draw-free subtrees
    // BVHNode::hit (accel.rs:58-83) returns the closest hit of its subtree, and for a subtree whose objects are all
    // Sphere / Rect / Boxy / lists of those (no ConstantMedium: nothing draws during traversal; no Translate/Rotate) that
    // result does not depend on the tree over them — only on the objects, the (tmin, tmax) it was called with and, when two
    // objects are hit at EXACTLY the same t, on which one the reference reaches last (tie rules, see tie_rank below).
    // BVHNode::new's trees (random axis, median split, accel.rs:98-136) cost ~2x the box tests of a surface-area-heuristic
    // tree over the same objects, so such subtrees are rebuilt here with a binned SAH builder and emitted in the same
    // threaded pre-order format; the kernel, the item records and the visit order OUTSIDE the subtree are unchanged.
    bool retree = true;                     // LinearizeOptions::retree
    static constexpr uint32_t RETREE_MIN = 16;     // objects: below this the reference's tree is kept as it is
    std::vector<int32_t> simple_count;      // per vk_bvh_node: number of object slots if the subtree is draw-free, -1 if not, -2 unknown
    uint32_t n_blocks = 0;
    bool world_rebuilt = false;             // the block is the whole world tree

    static bool retree_kind(uint32_t k) { return k == VK_KIND_SPHERE || k == VK_KIND_RECT || k == VK_KIND_LIST; }

    // simple_count for every node of the BVH rooted at `root` (iterative post-order; a cyclic graph is caught by emit_bvh's guard)
    bool classify(uint32_t root) {
        if (simple_count.empty()) simple_count.assign(d->n_bvh, -2);
        struct Fr { uint32_t node; int stage; };
        std::vector<Fr> st;
        st.push_back(Fr{root, 0});
        size_t guard = 0;
        while (!st.empty()) {
            if (++guard > (size_t)8 * (d->n_bvh + 16) + 1024) return fail(VK_ERR_BAD_ARG, "BVH graph is cyclic");
            Fr fr = st.back();
            const vk_bvh_node &n = d->bvh[fr.node];
            if (fr.stage == 0) {
                if (!check_ref(n.left) || !check_ref(n.right)) return false;
                st.back().stage = 1;
                for (vk_ref c : {n.left, n.right})
                    if (VK_REF_KIND(c) == VK_KIND_BVH && simple_count[VK_REF_INDEX(c)] == -2) st.push_back(Fr{VK_REF_INDEX(c), 0});
                continue;
            }
            int64_t total = 0;
            for (vk_ref c : {n.left, n.right}) {
                uint32_t k = VK_REF_KIND(c);
                int32_t v = k == VK_KIND_BVH ? simple_count[VK_REF_INDEX(c)] : (retree_kind(k) ? 1 : -1);
                if (v < 0 || total < 0) total = -1; else total += v;
            }
            simple_count[fr.node] = total > 0x3FFFFFFF ? -1 : (int32_t)total;
            st.pop_back();
        }
        return true;
    }

    // (dref2, rank2: the unit's second object; parent: the BVH node whose box gates the unit — exact re-treeing, rt_collect)
    struct RtObj {
        float mn[3], mx[3], c[3];
        uint32_t dref; uint32_t dref2 = 0; uint32_t parent = 0xFFFFFFFFu; uint32_t rank = 0, rank2 = 0;
    };
    bool retree_units = false;              // LinearizeOptions::retree == 2: exact re-treeing asked for (see rt_collect)
    // ... collected object by object, every sphere behind its OWN box (the near form, rt_grow_near), instead of unit by unit
    bool own_gates = false;
    bool unit_form() const { return retree_units && !own_gates; }
    static float rt_half_area(const float *mn, const float *mx) {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
    static void rt_grow(float *mn, float *mx, const float *omn, const float *omx) {
        for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], omn[a]); mx[a] = fmaxf(mx[a], omx[a]); }
    }
    // bounding_box() of a simple object, as the reference computes it (hittable.rs:97-102 Sphere, 258-269 Rect,
    // 355-357 Boxy = its corners, 396-418 Vec = surrounding boxes)
    bool rt_bounds(vk_ref r, float *mn, float *mx) {
        uint32_t k = VK_REF_KIND(r), i = VK_REF_INDEX(r);
        if (k == VK_KIND_SPHERE) {
            const vk_sphere &sp = d->spheres[i];
            // center -/+ radius exactly as Sphere::bounding_box does (hittable.rs:97-102): a negative radius (the hollow glass
            // sphere of scene.rs:123-127) gives an INVERTED box, which shrinks its ancestors' boxes in the reference's tree, so
            // hits on such a sphere depend on that tree: rt_collect sees the inverted box and keeps the tree handed over
            for (int a = 0; a < 3; a++) { mn[a] = sp.center[a] - sp.radius; mx[a] = sp.center[a] + sp.radius; }
            return true;
        }
        if (k == VK_KIND_RECT) {
            const vk_rect &q = d->rects[i];
            if (q.axis0 > 2 || q.axis1 > 2 || q.axis2 > 2) return fail(VK_ERR_BAD_ARG, "rect axis out of range");
            mn[q.axis0] = q.c0; mx[q.axis0] = q.c1; mn[q.axis1] = q.d0; mx[q.axis1] = q.d1;
            mn[q.axis2] = q.k - 0.0001f; mx[q.axis2] = q.k + 0.0001f;
            return true;
        }
        const vk_list &l = d->lists[i];
        if (l.count == 0 || (uint64_t)l.first + l.count > d->n_list_items) return false;
        bool inverted = false;
        for (uint32_t j = 0; j < l.count; j++) {
            vk_ref it = d->list_items[l.first + j];
            if (!check_ref(it)) return false;
            uint32_t ik = VK_REF_KIND(it);
            if (ik != VK_KIND_SPHERE && ik != VK_KIND_RECT) return false;
            float a[3], b[3];
            if (!rt_bounds(it, a, b)) return false;
            for (int x = 0; x < 3; x++) inverted |= !(a[x] <= b[x]);
            if (j == 0) { memcpy(mn, a, 12); memcpy(mx, b, 12); } else rt_grow(mn, mx, a, b);
        }
        // a hollow sphere among the items: reported as an inverted box, like the sphere itself
        if (inverted) { mn[0] = 1.0f; mx[0] = 0.0f; }
        return true;
    }

    // objects of the subtree in the REFERENCE's visiting order (pre-order, left first).
    // UNIT mode (retree_units): what is collected is not the single object but the reference's GATE for it: BVHNode::hit tests an
    // object child iff the node's own box passes (accel.rs:58-72), and — boxes of a BVHNode::new tree being nested — iff the boxes of
    // all its ancestors pass as well (a box that contains another passes whenever the inner one does: fl(b - o) / d is monotone in b,
    // and tmax only shrinks).  So a unit = {the node's box as handed over, its one or two object children in the reference's order},
    // and ANY tree of nested boxes over the units tests an object under the same condition as the reference: its unit's box against
    // the closest hit so far.  Checked here: every box of the subtree lies inside its parent's and is regular (finite, min <= max);
    // otherwise the reference's tree is kept.
    bool rt_collect(uint32_t root, uint32_t flip0, int32_t inst, std::vector<RtObj> &out, bool &ok) {
        struct Fr { vk_ref ref; uint32_t flip; uint32_t parent; };
        std::vector<Fr> st;
        st.push_back(Fr{VK_MAKE_REF(VK_KIND_BVH, root), flip0, 0xFFFFFFFFu});
        ok = true;
        std::unordered_map<uint32_t, uint32_t> in_block;
        uint32_t rank = 0;
        while (!st.empty()) {
            Fr fr = st.back(); st.pop_back();
            if (VK_REF_KIND(fr.ref) != VK_KIND_BVH) in_block[fr.ref & ~VK_REF_FLIP]++;
            if (VK_REF_KIND(fr.ref) == VK_KIND_BVH) {
                const uint32_t ni = VK_REF_INDEX(fr.ref);
                const vk_bvh_node &n = d->bvh[ni];
                if (retree_units) {
                    for (int a = 0; a < 3; a++) {
                        if (!(n.bb_min[a] <= n.bb_max[a]) || !std::isfinite(n.bb_min[a]) || !std::isfinite(n.bb_max[a])) {
                            ok = false; return true;
                        }
                        if (fr.parent != 0xFFFFFFFFu) {
                            const vk_bvh_node &pn = d->bvh[fr.parent];
                            if (!(pn.bb_min[a] <= n.bb_min[a] && n.bb_max[a] <= pn.bb_max[a])) { ok = false; return true; }
                        }
                    }
                }
                // len == 1: the same object twice, the second test is a no-op
                bool dup = n.left == n.right && VK_REF_KIND(n.left) != VK_KIND_BVH;
                if (!dup) st.push_back(Fr{n.right, fr.flip ^ ((VK_REF_KIND(n.right) == VK_KIND_BVH && (n.right & VK_REF_FLIP))
                    ? DREF_FLIP : 0u), ni});
                st.push_back(Fr{n.left, fr.flip ^ ((VK_REF_KIND(n.left) == VK_KIND_BVH && (n.left & VK_REF_FLIP)) ? DREF_FLIP : 0u), ni});
                continue;
            }
            RtObj o;
            if (retree_units && VK_REF_KIND(fr.ref) != VK_KIND_SPHERE) { ok = false; return true; }      // exact re-treeing: spheres only
            // ... of positive radius: a hollow sphere (negative radius, scene.rs:123-127) has an inverted box, which its unit's box need
            // not contain, so its hits can precede the unit's entry by the sphere's whole size — beyond any padding of the gate
            if (retree_units && !(d->spheres[VK_REF_INDEX(fr.ref)].radius > 0.0f)) { ok = false; return true; }
            if (!rt_bounds(fr.ref, o.mn, o.mx)) { ok = false; return status == VK_OK; }
            if (retree_units) {
                // the early-winner test (vk_trace.h segment_unsafe) takes the sphere's own box for a subset of its unit's: true of any
                // tree BVHNode::new builds, checked because the tree is the caller's
                const vk_bvh_node &pn = d->bvh[fr.parent];
                for (int a = 0; a < 3; a++)
                    if (!(pn.bb_min[a] <= o.mn[a] && o.mx[a] <= pn.bb_max[a]) && o.mn[a] <= o.mx[a]) { ok = false; return true; }
                // unit form: the object is gated by its unit's box; near form: by its own (grown in rt_grow_near)
                if (unit_form()) { memcpy(o.mn, pn.bb_min, 12); memcpy(o.mx, pn.bb_max, 12); }
            }
            for (int a = 0; a < 3; a++) {
                // NaN / inverted box: keep the reference's tree
                if (!(o.mn[a] <= o.mx[a]) || !std::isfinite(o.mn[a]) || !std::isfinite(o.mx[a])) { ok = false; return true; }
                o.c[a] = 0.5f * o.mn[a] + 0.5f * o.mx[a];
            }
            uint32_t dref;
            if (!convert_object(fr.ref, fr.flip, inst, dref)) return false;
            // the second object child of the same node (they are visited back to back): same unit
            if (unit_form() && !out.empty() && out.back().parent == fr.parent && out.back().dref2 == 0u) {
                out.back().dref2 = dref; out.back().rank2 = rank++;
                continue;
            }
            o.dref = dref; o.parent = fr.parent; o.rank = rank++;
            out.push_back(o);
        }
        for (const auto &kv : in_block)      // a shared object with an occurrence outside this block: see node_refs
            if (node_refs[kv.first] != kv.second) { ok = false; break; }
        return true;
    }

    // binned surface-area-heuristic build over objs[begin, end), emitted in threaded pre-order; leaves hold two objects
    void rt_emit(std::vector<RtObj> &objs, size_t begin, size_t end, int depth) {
        struct Job { size_t begin, end; int depth; uint32_t parent_item; };      // parent_item: INNER item whose skip link ends here
        // explicit stack: emit node, then left, then right; skip links are patched when a subtree is complete
        struct Fr { size_t begin, end; int depth; uint32_t item; int stage; size_t mid; };
        std::vector<Fr> st;
        st.push_back(Fr{begin, end, depth, 0, 0, 0});
        while (!st.empty()) {
            Fr &fr = st.back();
            size_t len = fr.end - fr.begin;
            if (fr.stage == 0) {
                DItem it; memset(&it, 0, sizeof(it));
                float mn[3], mx[3];
                memcpy(mn, objs[fr.begin].mn, 12); memcpy(mx, objs[fr.begin].mx, 12);
                for (size_t i = fr.begin + 1; i < fr.end; i++) rt_grow(mn, mx, objs[i].mn, objs[i].mx);
                it.mnx = mn[0]; it.mny = mn[1]; it.mnz = mn[2]; it.mxx = mx[0]; it.mxy = mx[1]; it.mxz = mx[2];
                if (unit_form() && len == 1) {          // one unit per leaf: the reference's box, its objects in the reference's order
                    it.w0 = objs[fr.begin].dref; it.w1 = objs[fr.begin].dref2;
                    L.items.push_back(it);
                    L.n_prims += it.w1 ? 2u : 1u;
                    st.pop_back();
                    continue;
                }
                if (!unit_form() && len <= 2) {
                    size_t a = fr.begin, b = fr.begin + 1;
                    // the larger object first: its hit's t culls more of what follows in this fixed-order walk
                    if (len == 2 && rt_half_area(objs[b].mn, objs[b].mx) > rt_half_area(objs[a].mn, objs[a].mx)) std::swap(a, b);
                    it.w0 = objs[a].dref; it.w1 = len == 2 ? objs[b].dref : 0u;
                    L.items.push_back(it);
                    L.n_prims += (uint32_t)len;
                    st.pop_back();
                    continue;
                }
                fr.item = (uint32_t)L.items.size();
                L.items.push_back(it);
                // ---- choose the split: 32 centroid bins per axis (16: 5 % more steps on C2; 64 and a full sweep of small nodes: no
                // fewer), cost = A(left) n(left) + A(right) n(right)
                float cmin[3], cmax[3];
                memcpy(cmin, objs[fr.begin].c, 12); memcpy(cmax, objs[fr.begin].c, 12);
                for (size_t i = fr.begin + 1; i < fr.end; i++) rt_grow(cmin, cmax, objs[i].c, objs[i].c);
                const int NB = 32;
                int best_axis = -1, best_split = 0; double best_cost = 1e300;
                if (fr.depth < 64) {
                    for (int ax = 0; ax < 3; ax++) {
                        if (!(cmax[ax] > cmin[ax])) continue;
                        float scale = (float)NB / (cmax[ax] - cmin[ax]);
                        float bmn[NB][3], bmx[NB][3]; size_t cnt[NB];
                        for (int b = 0; b < NB; b++) cnt[b] = 0;
                        for (size_t i = fr.begin; i < fr.end; i++) {
                            int b = (int)((objs[i].c[ax] - cmin[ax]) * scale);
                            b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                            if (cnt[b]++ == 0) { memcpy(bmn[b], objs[i].mn, 12); memcpy(bmx[b], objs[i].mx, 12);
                                } else rt_grow(bmn[b], bmx[b], objs[i].mn, objs[i].mx);
                        }
                        double la[NB]; size_t ln[NB];
                        float amn[3], amx[3]; size_t n = 0;
                        for (int b = 0; b < NB; b++) {
                            if (cnt[b]) { if (n == 0) { memcpy(amn, bmn[b], 12); memcpy(amx, bmx[b], 12);
                                } else rt_grow(amn, amx, bmn[b], bmx[b]); n += cnt[b]; }
                            la[b] = n ? (double)rt_half_area(amn, amx) : 0.0; ln[b] = n;
                        }
                        n = 0;
                        for (int b = NB - 1; b >= 1; b--) {
                            if (cnt[b]) { if (n == 0) { memcpy(amn, bmn[b], 12); memcpy(amx, bmx[b], 12);
                                } else rt_grow(amn, amx, bmn[b], bmx[b]); n += cnt[b]; }
                            if (n == 0 || ln[b - 1] == 0) continue;
                            double cost = la[b - 1] * (double)ln[b - 1] + (double)rt_half_area(amn, amx) * (double)n;
                            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = b; }
                        }
                    }
                }
                size_t mid = fr.begin + len / 2;
                if (best_axis >= 0) {
                    float lo = cmin[best_axis], scale = (float)NB / (cmax[best_axis] - cmin[best_axis]);
                    int ax = best_axis, sp = best_split;
                    auto itp = std::stable_partition(objs.begin() + fr.begin, objs.begin() + fr.end, [&](const RtObj &q) {
                        int b = (int)((q.c[ax] - lo) * scale);
                        b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                        return b < sp;
                    });
                    size_t m = (size_t)(itp - objs.begin());
                    if (m > fr.begin && m < fr.end) mid = m;
                }
                // the child with the larger box first (see above): swap the two slices if the right one is larger
                float lmn[3], lmx[3], rmn[3], rmx[3];
                memcpy(lmn, objs[fr.begin].mn, 12); memcpy(lmx, objs[fr.begin].mx, 12);
                for (size_t i = fr.begin + 1; i < mid; i++) rt_grow(lmn, lmx, objs[i].mn, objs[i].mx);
                memcpy(rmn, objs[mid].mn, 12); memcpy(rmx, objs[mid].mx, 12);
                for (size_t i = mid + 1; i < fr.end; i++) rt_grow(rmn, rmx, objs[i].mn, objs[i].mx);
                if (rt_half_area(rmn, rmx) > rt_half_area(lmn, lmx)) {
                    std::rotate(objs.begin() + fr.begin, objs.begin() + mid, objs.begin() + fr.end);
                    mid = fr.begin + (fr.end - mid);
                }
                fr.mid = mid; fr.stage = 1;
                Fr child{fr.begin, mid, fr.depth + 1, 0, 0, 0};
                st.push_back(child);           // (fr is invalid after this)
                continue;
            }
            if (fr.stage == 1) {
                fr.stage = 2;
                Fr child{fr.mid, fr.end, fr.depth + 1, 0, 0, 0};
                st.push_back(child);
                continue;
            }
            L.items[fr.item].w0 = (uint32_t)L.items.size();   // skip link
            st.pop_back();
        }
    }

    // Exact re-treeing: the trusted origin ball and the growth of every unit's gate box (vk_linearize.h rt_unit_growth).  The ball is
    // centred on the component-wise median of the sphere centres and reaches eight times as far as the spheres of ordinary size do
    // (a ground sphere thousands of times larger than the rest does not count towards the extent: rays start on its near side) and,
    // beyond that, as far as the far side of every sphere.  It is ONE radius, never shrunk: a unit whose far-origin check fails for it
    // counts as long (a smaller ball does not help where it matters — on the 1 M-sphere stress scene a ball of 1.7e4, the largest that
    // every compact unit passes, still leaves 71 000 of 524 000 units long by their growth alone).
    // The growth goes with the square of a unit's size, and BVHNode::new's units can be long (random axis, median split; a scene whose
    // spheres all share one coordinate wastes every split on that axis: one unit in a hundred of the 1 M-sphere stress scene is
    // longer than 100 sphere diameters).  Grown, such units would overlap everything around them.  A unit is LONG when its growth
    // would exceed RT_LONG_GROWTH of its smaller radius or its far-origin check fails, and a world with a long unit is not rebuilt in
    // the proven form.  (Round 4 tried to keep long units in a pruned copy of the tree as handed over, walked first in the reference's
    // order — "the reference has visited everything this walk has, so its closest hit is no larger, so the gate passes whenever the
    // reference's does".  Writing the proof out showed the hole: this walk can accept an EARLY candidate of a long unit that the
    // reference skips because it already holds a hit from a compact unit, and from then on its closest hit is SMALLER than the
    // reference's; a second early candidate in a later long unit is then missed although the reference takes it.  Two early candidates
    // on one ray: never observed, but not a theorem.  docs/gate_lemma.md section 4.)
    // `proven` = no long unit, and the units' boxes grow by less than RT_MAX_AREA_GROWTH in surface area on average.
    // Otherwise the scene is walked on the tree as handed over, unless the caller asked for VK_SCENE_EMPIRICAL_TREES: then every unit
    // is rebuilt with its box as handed over, the padding is RT_PAD_EMPIRICAL and the exactness of the rebuilt walk is what the test
    // suites have measured, not what the gate lemma proves (include/vecchio_amd.h; tests/test_gate_lemma.py constructs a ray on which
    // it fails).
    bool gate_grow = true, want_proof = true, proof_only = true;
    bool proven = false;
    double gate_pad = RT_PAD;
    bool rt_grow_units(std::vector<RtObj> &objs) {
        if (objs.empty()) return false;
        std::vector<float> cx, cy, cz, rr;
        for (const RtObj &o : objs)
            for (uint32_t dr : {o.dref, o.dref2}) {
                if (!dr) continue;
                const DSphere &sp = L.spheres[VKD_INDEX(dr)];
                cx.push_back(sp.cx); cy.push_back(sp.cy); cz.push_back(sp.cz); rr.push_back(sp.r);
            }
        // (the error analysis behind rt_eta assumes no overflow or underflow inside Sphere::hit: coordinates and radii below 2^30, radii
        // above 2^-40)
        for (size_t i = 0; i < rr.size(); i++)
            if (!(std::fabs(cx[i]) < 1073741824.0f && std::fabs(cy[i]) < 1073741824.0f && std::fabs(cz[i]) < 1073741824.0f &&
                  rr[i] < 1073741824.0f && rr[i] > 9.0949470177292824e-13f)) return false;
        auto median = [](std::vector<float> v) { std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end()); return (double)v[v.size() / 2]; };
        RtDomain dom;
        dom.c0[0] = median(cx); dom.c0[1] = median(cy); dom.c0[2] = median(cz);
        const double r_med = median(rr);
        double ext = 0.0, reach = 0.0;
        for (size_t i = 0; i < rr.size(); i++) {
            const double dx = cx[i] - dom.c0[0], dy = cy[i] - dom.c0[1], dz = cz[i] - dom.c0[2];
            const double far_side = std::sqrt(dx * dx + dy * dy + dz * dz) + (double)rr[i];
            reach = std::max(reach, far_side);
            if ((double)rr[i] <= 64.0 * r_med) ext = std::max(ext, far_side);
        }
        if (!(ext > 0.0) || !std::isfinite(reach)) return false;
        // The ball: eight times as far as the spheres of ordinary size reach and, beyond that, as far as the far side of every sphere:
        // paths do get inside a ground sphere (a grazing scatter from a hit point computed a hair below the surface) and then bounce
        // there to the depth limit — 0.4 % of the InOneWeekend scene's samples, 5 % of its segments, all starting up to two ground radii
        // away.  (A unit whose far-origin check fails for so large a ball counts as long.)
        dom.r0 = std::max(8.0 * ext, 1.05 * reach);
        const double long_thr = RT_LONG_GROWTH;
        std::vector<double> grow(objs.size());
        std::vector<char> is_long(objs.size(), 0);
        for (size_t i = 0; i < objs.size(); i++) {
            float c[2][3], r[2]; int n = 0;
            for (uint32_t dr : {objs[i].dref, objs[i].dref2}) {
                if (!dr) continue;
                const DSphere &sp = L.spheres[VKD_INDEX(dr)];
                c[n][0] = sp.cx; c[n][1] = sp.cy; c[n][2] = sp.cz; r[n] = sp.r; n++;
            }
            grow[i] = rt_unit_growth(objs[i].mn, objs[i].mx, n, c, r, dom, gate_pad);
            is_long[i] = grow[i] < 0.0 || grow[i] > long_thr * std::min(r[0], r[n - 1]);
        }
        double area0 = 0.0, area1 = 0.0, gmax = 0.0;
        size_t n_long = 0;
        for (size_t i = 0; i < objs.size(); i++) {
            if (is_long[i]) { n_long++; continue; }
            float mn[3], mx[3];
            const float g = std::nextafter((float)grow[i], INFINITY);
            for (int a = 0; a < 3; a++) { mn[a] = std::nextafter(objs[i].mn[a] - g, -INFINITY); mx[a] = std::nextafter(objs[i].mx[a] + g, INFINITY);
                if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) return false; }
            // (per unit, relative: a ground sphere's box would outweigh all others in a sum of areas)
            area0 += 1.0; area1 += (double)rt_half_area(mn, mx) / std::max((double)rt_half_area(objs[i].mn, objs[i].mx), 1e-300);
            gmax = std::max(gmax, (double)g);
        }
        const bool cheap = n_long == 0 && area1 <= (1.0 + RT_MAX_AREA_GROWTH) * area0;
        if (getenv("VK_RETREE_DEBUG"))
            fprintf(stderr, "vecchio_amd: exact re-treeing: %zu units, %zu of them long; trusted ball "
                "centre (%g %g %g) radius %g (extent %g); growth of the others: largest %g, leaf area x %.4f at padding %g -> %s\n",
                objs.size(), n_long, dom.c0[0], dom.c0[1], dom.c0[2], dom.r0, ext, gmax, area1 / std::max(area0, 1.0), gate_pad,
                cheap ? "grown gates (proven)" : "too dear");
        if (!cheap) return false;
        if (gate_grow)
            for (size_t i = 0; i < objs.size(); i++) {
                const float g = std::nextafter((float)grow[i], INFINITY);
                for (int a = 0; a < 3; a++) { objs[i].mn[a] = std::nextafter(objs[i].mn[a] - g, -INFINITY);
                    objs[i].mx[a] = std::nextafter(objs[i].mx[a] + g, INFINITY); }
            }
        for (int a = 0; a < 3; a++) L.trust_c0[a] = (float)dom.c0[a];
        L.trust_r0 = (float)(dom.r0 * (1.0 - 1e-6));
        return true;
    }

    // ---- The NEAR form of exact re-treeing (round 5; docs/gate_lemma.md section 7).  Where the unit form is not eligible (long units),
    // every sphere is gated by its OWN box — which section 6 of that note refutes for origins FAR from the sphere, and which the near
    // rule alone proves for origins NEAR it: for rho = |o - c| <= rho_near the point of an accepted root lies within eta(rho_near) of the
    // sphere, hence inside the own box grown by g >= 1.25 eta(rho_near), and the gate passes for every T above the candidate.  No
    // far-origin rule, no padding to speak of (RT_PAD_NEAR only pads the test's own rounding).  The gates of spheres farther than rho_near
    // from a ray's origin are NOT sound — so the walk's result is only taken when no such sphere can matter:
    //   REACH LEMMA.  The point H of a candidate lies within D(rho) = sqrt(R^2 + 32 u (rho + R)^2) <= R + eta(rho) of the centre, and
    //   |H - o| = cand |d|, so rho <= cand |d| + D(rho); rho - D(rho) is increasing in rho (D' <= sqrt(32 u) < 1).  Hence a sphere with
    //   rho > rho_near has cand |d| > rho_near - D(rho_near) >= reach := rho_near - max (R + eta(rho_near)): if the rebuilt walk ends with
    //   T_f |d| <= reach, every sphere holding a candidate below T_f is within rho_near of the origin, i.e. behind a sound gate, and
    //   Lemma 1's induction gives T_f <= T*; the safe-winner test (Lemma 2) does the rest as in the unit form.
    // A segment whose hit lies beyond `reach` — or that misses — is walked again on the tree as handed over (vk_trace.h segment_unsafe;
    // both trees in items[]: scenes traversed from global memory).  Spheres whose gate can be made sound for EVERY origin of the trusted
    // ball at a growth of RT_NEAR_BIG_GROWTH of their radius (a ground sphere: eta(rho) ~ 4 u rho^2 / R is small against a large R) are
    // "big": always sound, they do not enter `reach`.  The ball reaches RT_NEAR_BALL extents of the ordinary spheres; segments that
    // start outside it are the handed-over tree's from the start (begin_segment).
    float near_reach = 0.0f, near_radius = 0.0f;
    bool near_form = false, allow_near = true, allow_unit = true, near_first = true;
    bool near_spans = false;       // reach spans the small spheres' whole box (or there are none): hardly a segment is walked twice
    bool rt_grow_near(std::vector<RtObj> &objs) {
        near_spans = false;
        if (objs.empty()) return false;
        const size_t n = objs.size();
        const double U24 = 1.0 / 16777216.0;
        std::vector<float> cx(n), cy(n), cz(n), rr(n);
        for (size_t i = 0; i < n; i++) {
            const DSphere &sp = L.spheres[VKD_INDEX(objs[i].dref)];
            cx[i] = sp.cx; cy[i] = sp.cy; cz[i] = sp.cz; rr[i] = sp.r;
            if (!(std::fabs(cx[i]) < 1073741824.0f && std::fabs(cy[i]) < 1073741824.0f && std::fabs(cz[i]) < 1073741824.0f &&
                  rr[i] < 1073741824.0f && rr[i] > 9.0949470177292824e-13f)) return false;      // (rt_eta's range, as rt_grow_units)
        }
        auto median = [](std::vector<float> v) { std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end()); return (double)v[v.size() / 2]; };
        RtDomain dom;
        dom.c0[0] = median(cx); dom.c0[1] = median(cy); dom.c0[2] = median(cz);
        const double r_med = median(rr);
        std::vector<double> dist(n);
        double ext = 0.0, far_side = 0.0;
        for (size_t i = 0; i < n; i++) {
            const double dx = cx[i] - dom.c0[0], dy = cy[i] - dom.c0[1], dz = cz[i] - dom.c0[2];
            dist[i] = std::sqrt(dx * dx + dy * dy + dz * dz);
            far_side = std::max(far_side, dist[i] + (double)rr[i]);
            if ((double)rr[i] <= 64.0 * r_med) ext = std::max(ext, dist[i] + (double)rr[i]);
        }
        if (!(ext > 0.0) || !std::isfinite(far_side)) return false;
        // The ball only bounds what the BIG spheres' gates must cover (a small sphere's soundness is a matter of rho_near): as far as the far
        // side of every sphere — paths inside a ground sphere start there — if the world is eligible with the spheres that are big at THAT
        // size (a sphere that is big only for a smaller ball is a small one then), else RT_NEAR_BALL extents
        std::vector<char> big(n, 0);
        std::vector<double> grow(n, 0.0);
        std::vector<float> small_r;
        double rho_near = INFINITY, reach = INFINITY;
        auto attempt = [&](double r0, bool say) -> bool {
            dom.r0 = r0;
            // big spheres: sound for every origin of the ball (the near rule at the ball's far end) at a small relative growth
            small_r.clear();
            for (size_t i = 0; i < n; i++) {
                const double g_all = 1.25 * rt_eta(dist[i] + dom.r0, rr[i]);
                big[i] = g_all <= RT_NEAR_BIG_GROWTH * (double)rr[i];
                grow[i] = big[i] ? g_all : 0.0;
                if (!big[i]) small_r.push_back(rr[i]);
            }
            rho_near = INFINITY; reach = INFINITY;
            near_spans = small_r.empty();
            if (small_r.empty()) return true;
            // eta(rho_near) = 0.8 RT_NEAR_GROWTH R for the median small sphere
            const double rs = median(small_r);
            rho_near = std::sqrt(0.8 * RT_NEAR_GROWTH * rs * rs / RT_KAPPA) - rs;
            {   // ... a little farther if that lets `reach` span the small spheres' whole box: a ray that starts among them and travels
                // beyond reach has then left the box, which the clearance test sees — at up to twice the growth
                double lo2[3] = {INFINITY, INFINITY, INFINITY}, hi2[3] = {-INFINITY, -INFINITY, -INFINITY}, rmx = 0.0;
                for (size_t i = 0; i < n; i++) {
                    if (big[i]) continue;
                    for (int a = 0; a < 3; a++) { lo2[a] = std::min(lo2[a], (double)objs[i].mn[a]); hi2[a] = std::max(hi2[a], (double)objs[i].mx[a]); }
                    rmx = std::max(rmx, (double)rr[i]);
                }
                const double diag = std::sqrt((hi2[0] - lo2[0]) * (hi2[0] - lo2[0]) + (hi2[1] - lo2[1]) * (hi2[1] - lo2[1]) + (hi2[2] - lo2[2]) * (hi2[2] - lo2[2]));
                const double want = (diag + 2.0 * rmx) * 1.05, cap = std::sqrt(0.8 * 2.0 * RT_NEAR_GROWTH * rs * rs / RT_KAPPA) - rs;
                if (want > rho_near && want <= cap) rho_near = want;
                near_spans = rho_near >= want;
            }
            double worst = 0.0, grow_sum = 0.0;
            size_t n_small = 0;
            for (size_t i = 0; i < n; i++) {
                if (big[i]) continue;
                const double eta = rt_eta(rho_near, rr[i]);
                grow[i] = 1.25 * eta;
                // (eta goes with 1 / R: a sphere far smaller than the median grows by more than its own radius — which is still little
                // against its neighbours' boxes as long as it stays below the MEDIAN radius, and as long as such spheres are few: the mean
                // growth is held to 15 % of the median radius; beyond that the world is not eligible)
                grow_sum += grow[i]; n_small++;
                if (grow[i] > rs) {
                    if (say) fprintf(stderr, "vecchio_amd: near form: a sphere of radius %g among spheres of median radius %g "
                        "would grow by %g: not eligible\n", (double)rr[i], rs, grow[i]);
                    return false;
                }
                worst = std::max(worst, (double)rr[i] + eta);
            }
            reach = rho_near - worst;
            if (grow_sum > 0.15 * rs * (double)n_small) {
                if (say) fprintf(stderr, "vecchio_amd: near form: mean growth %g of a median radius of %g: not eligible\n",
                    grow_sum / (double)n_small, rs);
                return false;
            }
            if (!(reach > 8.0 * rs)) {                                      // (not even the neighbours are within reach)
                if (say) fprintf(stderr, "vecchio_amd: near form: reach %g for a median radius of %g: not eligible\n", reach, rs);
                return false;
            }
            return true;
        };
        {
            const bool say = getenv("VK_RETREE_DEBUG") != nullptr;
            const double wide = std::max(RT_NEAR_BALL * ext, 1.05 * far_side), narrow = RT_NEAR_BALL * ext;
            if (!attempt(wide, false) && !(narrow < wide && attempt(narrow, say))) {
                if (say && !(narrow < wide)) (void)attempt(wide, true);
                return false;
            }
        }
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double r_max_small = 0.0;
        L.n_big = 0;
        for (size_t i = 0; i < n; i++) {
            double maxabs = 0.0;
            for (int a = 0; a < 3; a++) maxabs = std::max(maxabs, std::max(std::fabs((double)objs[i].mn[a]), std::fabs((double)objs[i].mx[a])));
            const float g = std::nextafter((float)(grow[i] + 8.0 * U24 * maxabs), INFINITY);
            if (big[i]) {
                if (L.n_big != 0xFFFFFFFFu) {
                    if (L.n_big < 8u) { float *b = L.big[L.n_big++]; b[0] = cx[i]; b[1] = cy[i]; b[2] = cz[i]; b[3] = rr[i]; }
                    else L.n_big = 0xFFFFFFFFu;
                }
            } else {
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], objs[i].mn[a]); hi[a] = fmaxf(hi[a], objs[i].mx[a]); }
                const float cc[3] = {cx[i], cy[i], cz[i]};
                for (int a = 0; a < 3; a++) { clo[a] = fminf(clo[a], cc[a]); chi[a] = fmaxf(chi[a], cc[a]); }
                r_max_small = std::max(r_max_small, (double)rr[i]);
            }
            if (gate_grow)
                for (int a = 0; a < 3; a++) {
                    objs[i].mn[a] = std::nextafter(objs[i].mn[a] - g, -INFINITY); objs[i].mx[a] = std::nextafter(objs[i].mx[a] + g, INFINITY);
                    if (!std::isfinite(objs[i].mn[a]) || !std::isfinite(objs[i].mx[a])) return false;
                }
            for (int a = 0; a < 3; a++) objs[i].c[a] = 0.5f * objs[i].mn[a] + 0.5f * objs[i].mx[a];
        }
        for (int a = 0; a < 3; a++) { L.small_lo[a] = lo[a]; L.small_hi[a] = hi[a]; L.trust_c0[a] = (float)dom.c0[a]; }
        {   // CLEARANCE (DScene::clear_k ..., vk_trace.h clear_of_small_spheres; the box [small_lo, small_hi] is the one around the small
            // spheres' SURFACES).  A far sphere's hit point H = P(s) lies within D <= R + b (rho + R) of its centre, b = sqrt(32 u) =
            // 1.381e-3, i.e. within delta = b (rho + R) of the sphere, hence of that box; so s <= D_far + delta (D_far: the origin's distance
            // from the box's far corner) and rho <= s + R + delta give delta <= b (D_far + 2 R) / (1 - 2 b).  2 % and the boxes' rounding on
            // top (the slab test's own rounding, 3 u D_far, is a hundredth of the 2 %).
            const double b = std::sqrt(RT_KAPPA);
            double maxabs = 0.0;
            for (int a = 0; a < 3; a++) maxabs = std::max(maxabs, std::max(std::fabs((double)lo[a]), std::fabs((double)hi[a])));
            L.clear_k = (float)(1.02 * b / (1.0 - 2.0 * b));
            L.clear_r2 = (float)(2.0 * r_max_small * (1.0 + 1e-6));
            L.clear_slack = (float)(64.0 * U24 * maxabs);
            (void)clo; (void)chi;
        }
        L.trust_r0 = (float)(dom.r0 * (1.0 - 1e-6));
        near_reach = std::isfinite(reach) ? (float)(reach * (1.0 - 1e-5)) : 3.0e38f;
        near_radius = std::isfinite(rho_near) ? (float)rho_near : 3.0e38f;
        if (getenv("VK_RETREE_DEBUG"))
            fprintf(stderr, "vecchio_amd: exact re-treeing, near form: %zu spheres (%zu small, own boxes grown by 1.25 eta(%g)), reach %g; trusted "
                "ball centre (%g %g %g) radius %g\n", n, small_r.size(), rho_near, (double)near_reach, dom.c0[0], dom.c0[1], dom.c0[2], dom.r0);
        return true;
    }

    // ---- The GRID form of exact re-treeing (round 5; DGrid in vk_device_scene.h, vk_trace.h grid_step, docs/gate_lemma.md section 8).
    // `units`: every sphere of the world (dref, dref2).  Eligible: spheres of positive radius only (the caller saw to that), all but at
    // most 8 of them "small" (radius <= 2.5 median radii) and lying in a LAYER across y — the box around their surfaces at least four
    // times as wide in x and in z as in y — and at least 64 of them.  Cell size: one small sphere per cell on average, never below 2.5
    // median radii (a sphere's box then overlaps four cells at most), never so small that a side needs more than 1000 cells.
    bool allow_grid = true;
    bool rt_build_grid(const std::vector<RtObj> &units) {
        L.grid = DGrid{}; L.grid_cells.clear(); L.grid_refs.clear();
        std::vector<uint32_t> all;
        for (const RtObj &u : units) { all.push_back(u.dref); if (u.dref2) all.push_back(u.dref2); }
        for (uint32_t r : all) if (VKD_KIND(r) != DK_SPHERE) return false;
        if (all.size() < 72u) return false;
        std::vector<float> rr;
        for (uint32_t r : all) {
            const DSphere &sp = L.spheres[VKD_INDEX(r)];
            if (!(sp.r > 9.0949470177292824e-13f && sp.r < 1073741824.0f && std::fabs(sp.cx) < 1073741824.0f && std::fabs(sp.cy) < 1073741824.0f &&
                  std::fabs(sp.cz) < 1073741824.0f)) return false;
            rr.push_back(sp.r);
        }
        std::vector<float> tmp = rr;
        std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
        const double r_med = tmp[tmp.size() / 2];
        std::vector<uint32_t> always, field;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, r_max = 0.0;
        for (size_t i = 0; i < all.size(); i++) {
            const DSphere &sp = L.spheres[VKD_INDEX(all[i])];
            if ((double)sp.r > 2.5 * r_med) { always.push_back(all[i]); continue; }
            field.push_back(all[i]);
            const double c[3] = {sp.cx, sp.cy, sp.cz};
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c[a] - (double)sp.r); hi[a] = std::max(hi[a], c[a] + (double)sp.r); }
            r_max = std::max(r_max, (double)sp.r);
        }
        if (always.size() > 8u || field.size() < 64u) return false;
        // the largest first; those within 64 median radii (not a ground sphere, whose box every ray is in) share one gate box
        std::sort(always.begin(), always.end(), [&](uint32_t a, uint32_t b) { return L.spheres[VKD_INDEX(a)].r > L.spheres[VKD_INDEX(b)].r; });
        uint32_t n_gated = 0;
        float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t r : always) {
            const DSphere &sp = L.spheres[VKD_INDEX(r)];
            if (!((double)sp.r <= 64.0 * r_med)) continue;
            n_gated++;
            const float c[3] = {sp.cx, sp.cy, sp.cz};
            for (int a = 0; a < 3; a++) { alo[a] = fminf(alo[a], std::nextafter(c[a] - sp.r, -INFINITY)); ahi[a] = fmaxf(ahi[a], std::nextafter(c[a] + sp.r, INFINITY)); }
        }
        const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        if (!(ex >= 4.0 * ey && ez >= 4.0 * ey && ex > 0.0 && ez > 0.0)) return false;
        double h = std::sqrt(ex * ez / (double)field.size());
        if (const char *e = getenv("VK_GRID_CELL_SCALE")) { const double f = atof(e); if (f >= 0.25 && f <= 4.0) h *= f; }      // (diagnostics)
        h = std::max(h, 2.5 * r_med);
        h = std::max(h, std::max(ex, ez) / 990.0);
        double maxabs = 0.0;
        for (int a = 0; a < 3; a++) maxabs = std::max(maxabs, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
        // the cells' own margin: a sphere is registered wherever its box comes within m_reg of a cell, the walk visits every cell within
        // dl + m_walk of the ray: together they cover the f32 rounding of both (2^-20 of the coordinates, 2^-10 of a cell)
        const double m_reg = h / 1024.0 + maxabs / 1048576.0;
        const double g0x = lo[0] - 2.0 * m_reg, g0z = lo[2] - 2.0 * m_reg;
        const uint32_t nu = (uint32_t)std::ceil((ex + 4.0 * m_reg) / h), nv = (uint32_t)std::ceil((ez + 4.0 * m_reg) / h);
        if (nu < 1u || nv < 1u || nu > 1000u || nv > 1000u) return false;
        const float hf = (float)h, g0xf = (float)g0x, g0zf = (float)g0z;
        std::vector<uint32_t> count((size_t)nu * nv + 1u, 0u);
        auto range = [&](double a, double b, float g0, uint32_t n, uint32_t &i0, uint32_t &i1) {
            const double f0 = std::floor((a - m_reg - (double)g0) / (double)hf), f1 = std::floor((b + m_reg - (double)g0) / (double)hf);
            i0 = (uint32_t)std::min<double>(std::max(f0, 0.0), (double)(n - 1u)); i1 = (uint32_t)std::min<double>(std::max(f1, 0.0), (double)(n - 1u));
        };
        for (int pass = 0; pass < 2; pass++) {
            for (uint32_t r : field) {
                const DSphere &sp = L.spheres[VKD_INDEX(r)];
                uint32_t x0, x1, z0, z1;
                range((double)sp.cx - sp.r, (double)sp.cx + sp.r, g0xf, nu, x0, x1);
                range((double)sp.cz - sp.r, (double)sp.cz + sp.r, g0zf, nv, z0, z1);
                for (uint32_t ix = x0; ix <= x1; ix++)
                    for (uint32_t iz = z0; iz <= z1; iz++) {
                        const size_t c = (size_t)ix * nv + iz;
                        if (pass == 0) count[c + 1]++;
                        else L.grid_refs[count[c]++] = r;
                    }
            }
            if (pass == 0) {
                count[0] = (uint32_t)always.size();
                for (size_t c = 1; c < count.size(); c++) count[c] += count[c - 1];
                if (count.back() > (1u << 27)) return false;
                L.grid_cells = count;
                L.grid_refs.assign(count.back() + 1u, 0u);       // (+1: the walk reads refs in pairs)
                for (size_t i = 0; i < always.size(); i++) L.grid_refs[i] = always[i];
            }
        }
        DGrid &G = L.grid;
        G.nu = nu; G.nv = nv; G.ou = g0xf; G.ov = g0zf; G.cell = hf; G.inv_cell = 1.0f / hf;
        for (int a = 0; a < 3; a++) { G.lo[a] = std::nextafter((float)lo[a], -INFINITY); G.hi[a] = std::nextafter((float)hi[a], INFINITY); }
        const double b = std::sqrt(RT_KAPPA);
        G.k = (float)(1.05 * b / (1.0 - 2.0 * b));
        G.r2 = (float)(2.0 * r_max * (1.0 + 1e-6));
        G.slack = (float)(2.0 * m_reg);
        G.n_always = (uint32_t)always.size();
        G.n_gated = n_gated;
        for (int a = 0; a < 3; a++) { G.alo[a] = n_gated ? alo[a] : 0.0f; G.ahi[a] = n_gated ? ahi[a] : 0.0f; }
        if (getenv("VK_RETREE_DEBUG"))
            fprintf(stderr, "vecchio_amd: exact re-treeing, grid form: %zu spheres in %u x %u cells of %g (%zu references), %zu tested always\n",
                field.size(), nu, nv, h, L.grid_refs.size() - 1u - always.size(), always.size());
        return true;
    }

    // dense object id of a dref for the tie table: [spheres][rects][boxes][lists]
    uint32_t tie_id(uint32_t dref) const {
        uint32_t k = VKD_KIND(dref), i = VKD_INDEX(dref);
        if (k == DK_SPHERE) return i;
        if (k == DK_RECT) return d->n_spheres + i;
        if (k == DK_BOX) return d->n_spheres + d->n_rects + i;
        return d->n_spheres + d->n_rects + d->n_lists + i;     // DK_LIST (boxes and lists are both converted vk_lists: <= n_lists each)
    }

    // How often every simple object is the child of a BVH node anywhere in the description (a `len == 1` node's two equal
    // children count once).  An object has ONE tie rank, valid inside ONE rebuilt block: a shared object (one Arc under two
    // nodes) whose occurrences do not all lie in the block being collected keeps that block on the reference's tree.
    std::unordered_map<uint32_t, uint32_t> node_refs;
    void count_node_refs() {
        for (uint32_t i = 0; i < d->n_bvh; i++) {
            const vk_bvh_node &n = d->bvh[i];
            const bool dup = n.left == n.right;
            if (VK_REF_KIND(n.left) != VK_KIND_BVH) node_refs[n.left & ~VK_REF_FLIP]++;
            if (VK_REF_KIND(n.right) != VK_KIND_BVH && !dup) node_refs[n.right & ~VK_REF_FLIP]++;
        }
    }

    // tries to replace the subtree rooted at BVH node `root` by a rebuilt one; `done` says whether it did
    bool try_retree(uint32_t root, uint32_t flip, int32_t inst, bool &done) {
        done = false;
        if (!retree) return true;
        if (node_refs.empty()) count_node_refs();
        if (simple_count.empty() || simple_count[root] == -2) { if (!classify(root)) return false; }
        if (simple_count[root] < (int32_t)RETREE_MIN || n_blocks >= 4095u) return true;
        // exact re-treeing is for the world tree as a whole (the second render pass walks the whole tree as handed over)
        if (retree_units && !(inst < 0 && VK_REF_KIND(d->world) == VK_KIND_BVH && root == VK_REF_INDEX(d->world))) {
            retree = false; return true;
        }
        const size_t items0 = L.items.size(), boxes0 = L.boxes.size(), lists0 = L.lists.size(), refs0 = L.list_refs.size();
        const uint32_t prims0 = L.n_prims, feat0 = L.features;
        std::vector<RtObj> objs;
        objs.reserve((size_t)simple_count[root]);
        bool ok = true;
        own_gates = false;
        if (!rt_collect(root, flip, inst, objs, ok)) return false;
        std::vector<RtObj> all_units;           // (every unit has its tie ranks, the long ones too)
        if (ok && retree_units) {
            all_units = objs;
            // Two proven forms.  The NEAR form first where its reach spans the small spheres' whole box (hardly a segment is walked twice
            // then, and own boxes are tighter gates than the reference's units: the InOneWeekend scene +3 %), else the UNIT form where
            // it is cheap, else the near form whatever its reach (long units: the 10^6-sphere stress scene).  (A world of spheres
            // converts nothing but sphere references: collecting twice has no side effect to undo.)
            // (the test switches gate_grow = false / another padding leave a tree the lemma does not cover: it is never reported as proven
            // and, like any unproven tree, takes VK_SCENE_EMPIRICAL_TREES)
            const bool provable = want_proof && gate_grow && gate_pad == RT_PAD;
            std::vector<RtObj> own;
            bool near_ok = false;
            if (provable && allow_near) {
                own.reserve((size_t)simple_count[root]);
                own_gates = true;
                bool ok2 = true;
                if (!rt_collect(root, flip, inst, own, ok2)) return false;
                near_ok = ok2 && rt_grow_near(own);
                own_gates = false;
            }
            const bool spans = near_spans;
            proven = false;
            if (!(near_ok && spans && near_first)) proven = provable && allow_unit && rt_grow_units(objs);
            if (!proven && near_ok) {
                // (rt_grow_units writes the trusted ball only where it succeeds: the near form's is still in place)
                objs.swap(own); all_units.clear(); proven = true; near_form = true; own_gates = true;
            }
            if (!proven && proof_only) ok = false;      // (no VK_SCENE_EMPIRICAL_TREES: the tree as handed over rather than an unproven one)
        }
        const std::vector<RtObj> &sized = all_units.empty() ? objs : all_units;
        if (!ok || objs.size() < RETREE_MIN || sized.size() >= (1u << 20) ||
            (!sized.empty() && (sized.back().rank | sized.back().rank2) >= (1u << 20))) {
            // keep the reference's tree for this subtree: undo what collecting converted (memo entries past the old sizes)
            L.items.resize(items0); L.boxes.resize(boxes0); L.lists.resize(lists0); L.list_refs.resize(refs0); L.n_prims = prims0;
            L.features = feat0;
            for (auto it = box_memo.begin(); it != box_memo.end();) { if (it->second != 0xFFFFFFFFu &&
                it->second >= boxes0) it = box_memo.erase(it); else ++it; }
            for (auto it = list_memo.begin(); it != list_memo.end();) { if (it->second >= lists0) it = list_memo.erase(it); else ++it; }
            return true;
        }
        // tie ranks: position in the reference's visiting order, tagged with the block (ties are only re-ordered inside a block)
        n_blocks++;
        if (L.tie_rank.empty()) L.tie_rank.assign((size_t)d->n_spheres + d->n_rects + 2u * (size_t)d->n_lists, 0u);
        // An object that occurs more than once in the block (a shared Arc): among exact ties the reference ends up with the LAST
        // Rect it reaches, else with the FIRST object it reached (see tie_replaces in vk_trace.h), so a Rect is ranked by its last
        // occurrence and everything else by its first.
        const std::vector<RtObj> &ranked = all_units.empty() ? objs : all_units;
        for (size_t i = 0; i < ranked.size(); i++) {
            for (int k = 0; k < 2; k++) {
                const uint32_t dr = k ? ranked[i].dref2 : ranked[i].dref;
                if (k && !dr) continue;
                uint32_t id = tie_id(dr);
                bool first = (L.tie_rank[id] >> 20) != n_blocks;
                if (first || VKD_KIND(dr) == DK_RECT) L.tie_rank[id] = (n_blocks << 20) | (k ? ranked[i].rank2 : ranked[i].rank);
            }
        }
        // (a world without lights: it can only be rendered with the scatter integrator, whose sphere-only kernel has the grid walk)
        if (retree_units && proven && allow_grid && inst < 0 && items0 == 0 && d->n_lights == 0u) (void)rt_build_grid(ranked);
        rt_emit(objs, 0, objs.size(), 0);
        done = true;
        if (inst < 0 && VK_REF_KIND(d->world) == VK_KIND_BVH && root == VK_REF_INDEX(d->world)) world_rebuilt = true;
        return true;
    }

    // pre-order emission of one BVH (accel.rs:58-83 order: box, left, right); iterative to
    // survive 1M-primitive trees and degenerate depth
    bool emit_bvh(uint32_t root, uint32_t flip0, int32_t inst) {
        struct Frame { uint32_t node; uint32_t flip; uint32_t item; int stage; };
        std::vector<Frame> st;
        st.push_back(Frame{root, flip0, 0, 0});
        size_t guard = 0;
        while (!st.empty()) {
            if (++guard > (size_t)8 * (d->n_bvh + 16) + 1024) return fail(VK_ERR_BAD_ARG, "BVH graph is cyclic");
            Frame &fr = st.back();
            const vk_bvh_node &n = d->bvh[fr.node];
            if (fr.stage == 0) {
                if (!check_ref(n.left) || !check_ref(n.right)) return false;
                {
                    bool done = false;
                    uint32_t node = fr.node, flip = fr.flip;
                    if (!try_retree(node, flip, inst, done)) return false;
                    if (done) { st.pop_back(); continue; }
                }
                DItem it; memset(&it, 0, sizeof(it));
                it.mnx = n.bb_min[0]; it.mny = n.bb_min[1]; it.mnz = n.bb_min[2];
                it.mxx = n.bb_max[0]; it.mxy = n.bb_max[1]; it.mxz = n.bb_max[2];
                uint32_t lk = VK_REF_KIND(n.left), rk = VK_REF_KIND(n.right);
                uint32_t pos = (uint32_t)L.items.size();
                if (lk != VK_KIND_BVH && rk != VK_KIND_BVH) {
                    L.items.push_back(it);
                    uint32_t a, b = 0;
                    if (!convert_object(n.left, fr.flip, inst, a)) return false;
                    bool dup = (n.left == n.right);
                    bool skip_second = dup && draw_free(a);
                    if (dup && !skip_second && VKD_KIND(a) == DK_INSTANCE) { if (!draw_free_instance(VKD_INDEX(a), skip_second)) return false; }
                    if (!skip_second) { if (!convert_object(n.right, fr.flip, inst, b)) return false; }
                    L.items[pos].w0 = a; L.items[pos].w1 = b;
                    set_home(a, pos + 1, b);
                    set_home(b, pos + 1, 0);
                    L.n_prims += b ? 2 : 1;
                    st.pop_back();
                    continue;
                }
                L.items.push_back(it);
                fr.item = pos; fr.stage = 1;
                continue;
            }
            if (fr.stage == 1 || fr.stage == 2) {
                vk_ref child = fr.stage == 1 ? n.left : n.right;
                const bool second = fr.stage == 2;
                fr.stage += 1;
                uint32_t cf = fr.flip ^ ((child & VK_REF_FLIP) ? DREF_FLIP : 0u);
                if (second && n.left == n.right && VK_REF_KIND(child) == VK_KIND_BVH) {
                    // a len-1 node over a BVHNode of spheres only: the second call returns None (see draw_free_instance)
                    bool skip = false;
                    if (!spheres_only_subtree(VK_REF_INDEX(child), skip)) return false;
                    if (skip) continue;
                }
                if (VK_REF_KIND(child) == VK_KIND_BVH) {
                    Frame nf{VK_REF_INDEX(child), cf, 0, 0};
                    st.push_back(nf);   // (fr is invalid after this)
                } else {
                    uint32_t pos = (uint32_t)L.items.size();
                    L.items.push_back(always_hit_leaf());
                    uint32_t a;
                    if (!convert_object(child, fr.flip, inst, a)) return false;
                    L.items[pos].w0 = a;
                    set_home(a, pos + 1, 0);
                    L.n_prims += 1;
                }
                continue;
            }
            L.items[fr.item].w0 = (uint32_t)L.items.size();   // skip link: first item after this subtree
            st.pop_back();
        }
        if (L.items.size() >= (size_t)DREF_INDEX) return fail(VK_ERR_UNSUPPORTED, "too many BVH items");
        return true;
    }

    bool drain_pending() {
        while (!pending.empty()) {
            Pending p = pending.front();
            pending.pop_front();
            uint32_t begin = (uint32_t)L.items.size();
            if (!emit_bvh(p.bvh_index, p.flip, p.inst)) return false;
            L.instances[p.inst].child_begin = begin;
            L.instances[p.inst].child_end = (uint32_t)L.items.size();
        }
        return true;
    }

    bool materials_and_textures() {
        for (uint32_t i = 0; i < d->n_textures; i++) {
            const vk_texture &t = d->textures[i];
            DTexture o; memset(&o, 0, sizeof(o));
            o.kind = t.kind; o.r = t.color[0]; o.g = t.color[1]; o.b = t.color[2]; o.a = t.a; o.b_ = t.b; o.scale = t.scale;
            if (t.kind == VK_TEX_CHECKER) { if (t.a >= d->n_textures || t.b >= d->n_textures) return fail(VK_ERR_BAD_ARG,
                "checker child out of range"); }
            else if (t.kind == VK_TEX_IMAGE) { if (t.a >= d->n_images) return fail(VK_ERR_BAD_ARG, "image index out of range"); }
            else if (t.kind == VK_TEX_NOISE) { if (t.a >= d->n_perlins) return fail(VK_ERR_BAD_ARG, "perlin index out of range"); }
            else if (t.kind != VK_TEX_SOLID) return fail(VK_ERR_BAD_ARG, "unknown texture kind");
            if (t.kind != VK_TEX_SOLID) L.features |= VKF_TEXTURES;
            if (t.kind == VK_TEX_NOISE) L.features |= VKF_NOISE;
            L.textures.push_back(o);
        }
        for (uint32_t i = 0; i < d->n_images; i++) {
            const vk_image &im = d->images[i];
            if (!im.rgb || !im.width || !im.height) return fail(VK_ERR_BAD_ARG, "empty image");
            DImage o; o.width = im.width; o.height = im.height; o.offset = L.image_bytes.size();
            size_t n = (size_t)im.width * im.height * 3;
            L.image_bytes.insert(L.image_bytes.end(), im.rgb, im.rgb + n);
            while (L.image_bytes.size() % 16) L.image_bytes.push_back(0);
            L.images.push_back(o);
        }
        for (uint32_t i = 0; i < d->n_perlins; i++) {
            const vk_perlin &p = d->perlins[i];
            DPerlin o;
            for (int k = 0; k < 256; k++) {
                o.ranvec[k][0] = p.ranvec[k][0]; o.ranvec[k][1] = p.ranvec[k][1]; o.ranvec[k][2] = p.ranvec[k][2];
                o.perm_x[k] = (uint8_t)(p.perm_x[k] & 255u); o.perm_y[k] = (uint8_t)(p.perm_y[k] & 255u);
                o.perm_z[k] = (uint8_t)(p.perm_z[k] & 255u);
            }
            L.perlins.push_back(o);
        }
        for (uint32_t i = 0; i < d->n_materials; i++) {
            const vk_material &m = d->materials[i];
            DMaterial o; memset(&o, 0, sizeof(o));
            o.kind = m.kind; o.tex = m.texture; o.param = m.param; o.tex_kind = VK_TEX_SOLID;
            bool needs_tex = m.kind == VK_MAT_LAMBERTIAN || m.kind == VK_MAT_METAL || m.kind == VK_MAT_DIFFUSE_LIGHT ||
                m.kind == VK_MAT_ISOTROPIC;
            if (needs_tex) {
                if (m.texture >= d->n_textures) return fail(VK_ERR_BAD_ARG, "texture index out of range");
                const vk_texture &t = d->textures[m.texture];
                o.tex_kind = t.kind; o.r = t.color[0]; o.g = t.color[1]; o.b = t.color[2];
            } else if (m.kind == VK_MAT_SPEC_DIFFUSE) {
                if (m.a >= d->n_materials || m.b >= d->n_materials) return fail(VK_ERR_BAD_ARG, "spec_diffuse child out of range");
                if (m.a >= 65536 || m.b >= 65536) return fail(VK_ERR_UNSUPPORTED,
                    "device path: SpecDiffuse children must have material index < 65536");
                o.ab = m.a | (m.b << 16);
                L.features |= VKF_SPEC_DIFFUSE | VKF_TEXTURES;
            } else if (m.kind != VK_MAT_DIELECTRIC) return fail(VK_ERR_BAD_ARG, "unknown material kind");
            L.materials.push_back(o);
        }
        return true;
    }

    bool run() {
        if (!d) return fail(VK_ERR_BAD_ARG, "null scene description");
        if (d->abi_version != VK_ABI_VERSION) return fail(VK_ERR_BAD_ARG, "abi version mismatch");
        L.tie_base_rect = d->n_spheres; L.tie_base_box = d->n_spheres + d->n_rects;
        L.tie_base_list = d->n_spheres + d->n_rects + d->n_lists;
        if (!materials_and_textures()) return false;
        for (uint32_t i = 0; i < d->n_spheres; i++) {
            const vk_sphere &s = d->spheres[i];
            if (s.material >= d->n_materials) return fail(VK_ERR_BAD_ARG, "material index out of range");
            L.spheres.push_back(DSphere{s.center[0], s.center[1], s.center[2], s.radius});
            L.sphere_mat.push_back(s.material);
            const DMaterial &sm = L.materials[s.material];
            // (tex_kind is only set for materials that read a texture)
            if (sm.tex_kind == VK_TEX_NOISE && L.n_noise_spheres != 0xFFFFFFFFu) {
                if (L.n_noise_spheres < 4u) {
                    L.noise_sphere[L.n_noise_spheres] = i; L.noise_tex[L.n_noise_spheres] = sm.tex;
                    L.noise_perlin[L.n_noise_spheres] = L.textures[sm.tex].a;
                    L.n_noise_spheres++;
                } else L.n_noise_spheres = 0xFFFFFFFFu;
            }
        }
        for (uint32_t i = 0; i < d->n_moving_spheres; i++) {
            const vk_moving_sphere &s = d->moving_spheres[i];
            if (s.material >= d->n_materials) return fail(VK_ERR_BAD_ARG, "material index out of range");
            DMoving m; memset(&m, 0, sizeof(m));
            for (int k = 0; k < 3; k++) { m.c0[k] = s.center0[k]; m.c1[k] = s.center1[k]; }
            m.t0 = s.time0; m.t1 = s.time1; m.r = s.radius; m.mat = s.material;
            L.moving.push_back(m);
        }
        for (uint32_t i = 0; i < d->n_rects; i++) {
            const vk_rect &s = d->rects[i];
            if (s.material >= d->n_materials) return fail(VK_ERR_BAD_ARG, "material index out of range");
            if (s.axis0 > 2 || s.axis1 > 2 || s.axis2 > 2) return fail(VK_ERR_BAD_ARG, "rect axis out of range");
            DRect q; memset(&q, 0, sizeof(q));
            q.c0 = s.c0; q.c1 = s.c1; q.d0 = s.d0; q.d1 = s.d1; q.k = s.k;
            q.axes = (uint32_t)s.axis0 | ((uint32_t)s.axis1 << 2) | ((uint32_t)s.axis2 << 4); q.mat = s.material;
            L.rects.push_back(q);
        }
        if (!check_ref(d->world)) return false;
        if (VK_REF_KIND(d->world) == VK_KIND_BVH) {
            if (!emit_bvh(VK_REF_INDEX(d->world), (d->world & VK_REF_FLIP) ? DREF_FLIP : 0u, -1)) return false;
        } else {
            L.items.push_back(always_hit_leaf());
            uint32_t a;
            if (!convert_object(d->world, 0, -1, a)) return false;
            L.items[0].w0 = a;
            set_home(a, 1, 0);
            L.n_prims = 1;
        }
        uint32_t world_end = (uint32_t)L.items.size();
        if (!drain_pending()) return false;
        // the world range must be [0, world_end): the kernel starts with end = n_items only
        // when nothing follows; otherwise it uses world_end stored in items' owner (see api)
        world_items = world_end;
        for (uint32_t i = 0; i < d->n_lights; i++) {
            vk_ref r = d->lights[i];
            if (!check_ref(r)) return false;
            uint32_t k = VK_REF_KIND(r);
            uint32_t out = 0;   // kinds without pdf_value/random impls behave as the trait defaults
            if (k == VK_KIND_RECT && !(r & VK_REF_FLIP)) {
                const vk_rect &q = d->rects[VK_REF_INDEX(r)];
                if (!(q.c0 < q.c1) || !(q.d0 < q.d1)) return fail(VK_ERR_BAD_ARG,
                    "light Rect with an empty extent (gen_range(c0,c1) panics, hittable.rs:287-288)");
            }
            if (k == VK_KIND_SPHERE || k == VK_KIND_RECT) out = simple_dref(r, 0);
            else if (k == VK_KIND_LIST) { uint32_t li; if (!convert_list(VK_REF_INDEX(r), li)) return false;
                out = VKD_MAKE(DK_LIST, li) | ((r & VK_REF_FLIP) ? DREF_FLIP : 0u); }
            L.lights.push_back(out);
        }
        return status == VK_OK;
    }
    uint32_t world_items = 0;
};

}  // namespace

int linearize(const vk_scene_desc *desc, LinearScene &out, std::string &err, const LinearizeOptions &opt) {
    Builder b(desc, out, err);
    int mode = opt.retree;
    if (mode < 0) mode = !desc ? 0 : ((desc->flags & VK_SCENE_FAST_ACCEL) ? 1 : ((desc->flags & VK_SCENE_REFERENCE_TREE) ? 0 : 2));
    b.retree = mode != 0;
    b.retree_units = mode == 2;
    // opt.t_pad is a test switch; anything that is not a positive number below 1 would leave a rebuilt tree without a working gate: the
    // tree as handed over is used instead
    if (opt.t_pad != 0.0f && !(opt.t_pad > 0.0f && opt.t_pad < 1.0f)) { b.retree = false; b.retree_units = false; }
    b.gate_grow = opt.gate_grow;
    b.want_proof = opt.want_proof;
    b.proof_only = !(desc && (desc->flags & VK_SCENE_EMPIRICAL_TREES) != 0u) && !opt.allow_empirical;
    b.allow_near = opt.near_form; b.allow_unit = opt.unit_form; b.near_first = opt.near_first; b.allow_grid = opt.grid_form;
    if (opt.t_pad > 0.0f && opt.t_pad < 1.0f) b.gate_pad = opt.t_pad;
    if (!b.run()) return b.status == VK_OK ? VK_ERR_BAD_ARG : b.status;
    out.world_items = b.world_items;
    if (out.features == 0u) {            // what the sphere-only kernel variants shade from
        out.sphere_material.resize(out.spheres.size());
        for (size_t i = 0; i < out.spheres.size(); i++) out.sphere_material[i] = out.materials[out.sphere_mat[i]];
    }
    if (b.retree_units && b.n_blocks != 0 && !(b.n_blocks == 1 && b.world_rebuilt && out.features == 0u && out.instances.empty() &&
        out.spheres.size() < (1u << 26))) {
        // (cannot happen for a world of spheres only; never leave a unit-mode tree without its second pass)
        out = LinearScene();
        LinearizeOptions o0; o0.retree = 0;
        return linearize(desc, out, err, o0);
    }
    if (b.retree_units && b.n_blocks == 1) {
        // exact mode: the tree as handed over rides along (same conversion order, hence the same sphere indices)
        LinearScene ref; std::string e2;
        LinearizeOptions o2; o2.retree = 0;
        int st = linearize(desc, ref, e2, o2);
        if (st != VK_OK) { err = e2; return st; }
        const bool same = ref.spheres.size() == out.spheres.size() && ref.sphere_mat == out.sphere_mat &&
            memcmp(ref.spheres.data(), out.spheres.data(), ref.spheres.size() * sizeof(DSphere)) == 0;
        if (!same) { err = "exact re-tree: the two linearisations number the spheres differently"; return VK_ERR_UNSUPPORTED; }
        out.ref_items = ref.items;
        // every sphere's leaf in the tree as handed over (the node whose box gates it there): the safe-winner test's second chance
        // (vk_trace.h segment_unsafe).  A shared sphere: the first leaf that holds it.
        out.unit_item.assign(out.spheres.size(), 0xFFFFFFFFu);
        {
            // (an object that is a bare child of a node — which BVHNode::new never builds, descriptions may — sits in a leaf item whose
            // box lets everything through: what gates it in the reference is its PARENT node's box, the innermost open inner item)
            std::vector<uint32_t> open;
            for (size_t i = 0; i < out.ref_items.size(); i++) {
                const DItem &it = out.ref_items[i];
                while (!open.empty() && out.ref_items[open.back()].w0 <= (uint32_t)i) open.pop_back();
                if ((it.w0 >> 28) == 0u) { open.push_back((uint32_t)i); continue; }
                const bool bare = !(it.mnx > -1.0e38f);
                const uint32_t gate = bare ? (open.empty() ? 0xFFFFFFFFu : open.back()) : (uint32_t)i;
                for (uint32_t w : {it.w0, it.w1})
                    if (w && VKD_KIND(w) == DK_SPHERE && out.unit_item[VKD_INDEX(w)] == 0xFFFFFFFFu) out.unit_item[VKD_INDEX(w)] = gate;
            }
        }
        // The gate's relative padding.  Proven form (rt_grow_units succeeded): RT_PAD, which covers the hit points of FAR origins — they
        // may precede the ray's entry into the unit's grown box by up to the box's size; near origins are covered by the growth.
        // Empirical form: the units' boxes as handed over and RT_PAD_EMPIRICAL (1/256 lost one sample in 1.9 G on the stress scene, 1/16
        // none in 12 G: DESIGN.md section 5), every origin "trusted".
        out.proven = b.proven;
        out.near_form = b.near_form; out.reach = b.near_form ? b.near_reach : 0.0f; out.near_radius = b.near_radius;
        out.near_spans = b.near_form && b.near_spans;
        out.t_pad = (float)(b.near_form ? RT_PAD_NEAR : (b.proven ? RT_PAD : RT_PAD_EMPIRICAL));
        if (opt.t_pad > 0.0f && opt.t_pad < 1.0f) out.t_pad = opt.t_pad;
        if (!b.proven) { out.trust_c0[0] = out.trust_c0[1] = out.trust_c0[2] = 0.0f; out.trust_r0 = INFINITY; }
    }
    return VK_OK;
}

}  // namespace vkd
