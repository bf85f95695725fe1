// host.cpp — flatten() impls, BVHNode::new, Camera::new (see vecchio_host.hpp)
#include "vecchio_host.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include <zlib.h>

namespace vecchio {

static thread_local vk::BuildRng g_build_rng = vk::rng_for_stream(1, 0);
vk::BuildRng &thread_rng() { return g_build_rng; }
void seed_thread_rng(uint64_t seed) { g_build_rng = vk::rng_for_stream(seed, 0); }

// ------------------------------------------------------------------ FlatBuilder
uint32_t FlatBuilder::material(const MaterialP &m) {
    auto it = seen_material.find(m.get());
    if (it != seen_material.end()) return it->second;
    uint32_t idx = m->flatten(*this);
    seen_material[m.get()] = idx;
    return idx;
}
uint32_t FlatBuilder::texture(const TextureP &t) {
    auto it = seen_texture.find(t.get());
    if (it != seen_texture.end()) return it->second;
    uint32_t idx = t->flatten(*this);
    seen_texture[t.get()] = idx;
    return idx;
}
vk_ref FlatBuilder::hittable(const HittableP &h) {
    auto it = seen_hittable.find(h.get());
    if (it != seen_hittable.end()) return it->second;
    vk_ref r = h->flatten(*this);
    seen_hittable[h.get()] = r;
    return r;
}
vk_scene_desc FlatBuilder::desc() {
    for (size_t i = 0; i < images.size(); i++) images[i].rgb = image_storage[i].data();
    vk_scene_desc d;
    memset(&d, 0, sizeof(d));
    d.abi_version = VK_ABI_VERSION;
    d.n_bvh = (uint32_t)bvh.size(); d.bvh = bvh.data();
    d.n_spheres = (uint32_t)spheres.size(); d.spheres = spheres.data();
    d.n_moving_spheres = (uint32_t)moving_spheres.size(); d.moving_spheres = moving_spheres.data();
    d.n_rects = (uint32_t)rects.size(); d.rects = rects.data();
    d.n_lists = (uint32_t)lists.size(); d.lists = lists.data();
    d.n_list_items = (uint32_t)list_items.size(); d.list_items = list_items.data();
    d.n_media = (uint32_t)media.size(); d.media = media.data();
    d.n_translates = (uint32_t)translates.size(); d.translates = translates.data();
    d.n_rotates = (uint32_t)rotates.size(); d.rotates = rotates.data();
    d.n_materials = (uint32_t)materials.size(); d.materials = materials.data();
    d.n_textures = (uint32_t)textures.size(); d.textures = textures.data();
    d.n_images = (uint32_t)images.size(); d.images = images.data();
    d.n_perlins = (uint32_t)perlins.size(); d.perlins = perlins.data();
    d.world = world;
    d.n_lights = (uint32_t)lights.size(); d.lights = lights.data();
    return d;
}

// ------------------------------------------------------------------ textures
uint32_t SolidColor::flatten(FlatBuilder &b) const {
    vk_texture t; memset(&t, 0, sizeof(t));
    t.kind = VK_TEX_SOLID; t.color[0] = color_value.x; t.color[1] = color_value.y; t.color[2] = color_value.z;
    b.textures.push_back(t);
    return (uint32_t)b.textures.size() - 1;
}
uint32_t Checker::flatten(FlatBuilder &b) const {
    vk_texture t; memset(&t, 0, sizeof(t));
    t.kind = VK_TEX_CHECKER; t.a = b.texture(odd); t.b = b.texture(even);
    b.textures.push_back(t);
    return (uint32_t)b.textures.size() - 1;
}
uint32_t ImageTexture::flatten(FlatBuilder &b) const {
    b.image_storage.push_back(buf);
    vk_image im; im.width = width; im.height = height; im.rgb = nullptr;  // pointer fixed up after all pushes
    b.images.push_back(im);
    vk_texture t; memset(&t, 0, sizeof(t));
    t.kind = VK_TEX_IMAGE; t.a = (uint32_t)b.images.size() - 1;
    b.textures.push_back(t);
    return (uint32_t)b.textures.size() - 1;
}
// Where ImageTexture::open finds the decoded images: the directory that plays the role of the reference's `assets/`.
static std::string g_assets_dir;
void set_assets_dir(const std::string &dir) { g_assets_dir = dir; }
std::string assets_dir() {
    if (!g_assets_dir.empty()) return g_assets_dir;
    const char *e = getenv("VECCHIO_ASSETS");
    return e && *e ? std::string(e) : std::string("assets");  // relative to the working directory, as in the reference
}

// A binary 8-bit PPM (P6), plain or gzip'd (zlib's gz* functions read both): the bytes `png::Reader::next_frame` leaves in
// ImageTexture::buf (material.rs:269-279) — row 0 = top, 3 bytes per pixel.
std::shared_ptr<ImageTexture> ImageTexture::from_ppm(const std::string &path) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    auto token = [&]() {  // whitespace-separated header token, '#' comments skipped
        std::string t;
        int c;
        while ((c = gzgetc(f)) != -1) {
            if (c == '#') { while ((c = gzgetc(f)) != -1 && c != '\n') {} continue; }
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { if (!t.empty()) break; continue; }
            t.push_back((char)c);
        }
        return t;
    };
    std::string magic = token();
    unsigned w = (unsigned)atoi(token().c_str()), h = (unsigned)atoi(token().c_str()), maxv = (unsigned)atoi(token().c_str());
    if (magic != "P6" || maxv != 255 || !w || !h || w > 65535 || h > 65535) {
        gzclose(f);
        throw std::runtime_error("not a binary 8-bit PPM: " + path);
    }
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    size_t got = 0;
    while (got < rgb.size()) {
        int n = gzread(f, rgb.data() + got, (unsigned)std::min<size_t>(rgb.size() - got, 1u << 30));
        if (n <= 0) break;
        got += (size_t)n;
    }
    gzclose(f);
    if (got != rgb.size()) throw std::runtime_error("short PPM: " + path);
    return std::make_shared<ImageTexture>(w, h, std::move(rgb));
}
// ImageTexture::new("assets/<name>.png") (material.rs:269-279).  The PNG decode is third-party (`png 0.16.6`, lossless);
// here the decoded bytes come from <assets_dir>/<name>.ppm.gz (tests/golden/make_assets.py made them from the reference's
// PNGs), or from <name>.ppm.  A missing file is an error, as it is in the reference (`File::open(path).unwrap()`).
std::shared_ptr<ImageTexture> ImageTexture::open(const std::string &path) {
    std::string name = path;
    size_t slash = name.find_last_of('/');
    if (slash != std::string::npos) name = name.substr(slash + 1);
    if (name.size() > 4 && name.compare(name.size() - 4, 4, ".png") == 0) name.resize(name.size() - 4);
    std::string base = assets_dir() + "/" + name;
    for (const char *ext : {".ppm.gz", ".ppm"}) {
        std::string p = base + ext;
        if (FILE *t = fopen(p.c_str(), "rb")) { fclose(t); return from_ppm(p); }
    }
    throw std::runtime_error("cannot open " + base + ".ppm.gz (decoded copy of " + path + "; set VECCHIO_ASSETS or --assets)");
}

Perlin::Perlin() {  // material.rs:357-377
    for (int i = 0; i < 256; i++) random_data[i] = Vec3::random_range(-1.0f, 1.0f).unit_vector();
    for (uint32_t i = 0; i < 256; i++) { perm_x[i] = i; perm_y[i] = i; perm_z[i] = i; }
    // SliceRandom::shuffle (rand 0.7.3): for i in (1..len).rev() { swap(i, gen_index(i+1)) }
    auto shuffle = [](uint32_t *p) {
        for (uint32_t i = 255; i >= 1; i--) std::swap(p[i], p[gen_index(i + 1)]);
    };
    shuffle(perm_x); shuffle(perm_y); shuffle(perm_z);
}
uint32_t NoiseTexture::flatten(FlatBuilder &b) const {
    vk_perlin p;
    for (int i = 0; i < 256; i++) {
        p.ranvec[i][0] = noise.random_data[i].x; p.ranvec[i][1] = noise.random_data[i].y; p.ranvec[i][2] = noise.random_data[i].z;
        p.perm_x[i] = noise.perm_x[i]; p.perm_y[i] = noise.perm_y[i]; p.perm_z[i] = noise.perm_z[i];
    }
    b.perlins.push_back(p);
    vk_texture t; memset(&t, 0, sizeof(t));
    t.kind = VK_TEX_NOISE; t.a = (uint32_t)b.perlins.size() - 1; t.scale = scale;
    b.textures.push_back(t);
    return (uint32_t)b.textures.size() - 1;
}

// ------------------------------------------------------------------ materials
static uint32_t push_mat(FlatBuilder &b, uint32_t kind, uint32_t tex, float param, uint32_t a = 0, uint32_t bb = 0) {
    vk_material m; m.kind = kind; m.texture = tex; m.param = param; m.a = a; m.b = bb;
    b.materials.push_back(m);
    return (uint32_t)b.materials.size() - 1;
}
uint32_t Lambertian::flatten(FlatBuilder &b) const { return push_mat(b, VK_MAT_LAMBERTIAN, b.texture(albedo), 0.0f); }
uint32_t Metal::flatten(FlatBuilder &b) const { return push_mat(b, VK_MAT_METAL, b.texture(albedo), fuzz); }
uint32_t Dielectric::flatten(FlatBuilder &b) const { return push_mat(b, VK_MAT_DIELECTRIC, 0, ref_idx); }
uint32_t DiffuseLight::flatten(FlatBuilder &b) const { return push_mat(b, VK_MAT_DIFFUSE_LIGHT, b.texture(emit), 0.0f); }
uint32_t Isotropic::flatten(FlatBuilder &b) const { return push_mat(b, VK_MAT_ISOTROPIC, b.texture(albedo), 0.0f); }
uint32_t SpecDiffuse::flatten(FlatBuilder &b) const {
    uint32_t s = b.material(specular), d = b.material(diffuse);
    return push_mat(b, VK_MAT_SPEC_DIFFUSE, 0, pct, s, d);
}

// ------------------------------------------------------------------ hittables
vk_ref Sphere::flatten(FlatBuilder &b) const {
    vk_sphere s; s.center[0] = center.x; s.center[1] = center.y; s.center[2] = center.z; s.radius = radius;
    s.material = b.material(material);
    b.spheres.push_back(s);
    return VK_MAKE_REF(VK_KIND_SPHERE, b.spheres.size() - 1);
}
vk_ref MovingSphere::flatten(FlatBuilder &b) const {
    vk_moving_sphere s;
    s.center0[0] = center0.x; s.center0[1] = center0.y; s.center0[2] = center0.z;
    s.center1[0] = center1.x; s.center1[1] = center1.y; s.center1[2] = center1.z;
    s.time0 = time0; s.time1 = time1; s.radius = radius; s.material = b.material(material);
    b.moving_spheres.push_back(s);
    return VK_MAKE_REF(VK_KIND_MOVING_SPHERE, b.moving_spheres.size() - 1);
}
vk_ref Rect::flatten(FlatBuilder &b) const {
    vk_rect r; r.c0 = c0; r.c1 = c1; r.d0 = d0; r.d1 = d1; r.k = k;
    r.axis0 = (uint8_t)axis0; r.axis1 = (uint8_t)axis1; r.axis2 = (uint8_t)axis2; r._pad = 0; r.material = b.material(mat);
    b.rects.push_back(r);
    return VK_MAKE_REF(VK_KIND_RECT, b.rects.size() - 1);
}
vk_ref FlipFace::flatten(FlatBuilder &b) const { return b.hittable(ptr) ^ VK_REF_FLIP; }  // only negates `front`

Boxy::Boxy(Vec3 p0, Vec3 p1, MaterialP mat) : box_min(p0), box_max(p1) {  // hittable.rs:321-359
    assert(p0.x < p1.x); assert(p0.y < p1.y); assert(p0.z < p1.z);
    sides.push_back(Rect::XYRect(p0.x, p1.x, p0.y, p1.y, p1.z, mat));
    sides.push_back(std::make_shared<FlipFace>(Rect::XYRect(p0.x, p1.x, p0.y, p1.y, p0.z, mat)));
    sides.push_back(Rect::XZRect(p0.x, p1.x, p0.z, p1.z, p1.y, mat));
    sides.push_back(std::make_shared<FlipFace>(Rect::XZRect(p0.x, p1.x, p0.z, p1.z, p0.y, mat)));
    sides.push_back(Rect::YZRect(p0.y, p1.y, p0.z, p1.z, p1.x, mat));
    sides.push_back(std::make_shared<FlipFace>(Rect::YZRect(p0.y, p1.y, p0.z, p1.z, p0.x, mat)));
}
static vk_ref flatten_list(FlatBuilder &b, const std::vector<HittableP> &items) {
    std::vector<vk_ref> refs;
    for (auto &h : items) refs.push_back(b.hittable(h));
    vk_list l; l.first = (uint32_t)b.list_items.size(); l.count = (uint32_t)refs.size();
    b.list_items.insert(b.list_items.end(), refs.begin(), refs.end());
    b.lists.push_back(l);
    return VK_MAKE_REF(VK_KIND_LIST, b.lists.size() - 1);
}
vk_ref Boxy::flatten(FlatBuilder &b) const { return flatten_list(b, sides); }  // Boxy::hit forwards to sides (hittable.rs:362-365)
std::optional<AxisBB> HittableList::bounding_box(float t0, float t1) const {  // hittable.rs:396-418
    if (items.empty()) return std::nullopt;
    std::optional<AxisBB> out;
    bool first = true;
    for (auto &o : items) {
        auto bb = o->bounding_box(t0, t1);
        if (!bb) return std::nullopt;
        out = first ? *bb : AxisBB::surrounding_box(*bb, *out);
        first = false;
    }
    return out;
}
vk_ref HittableList::flatten(FlatBuilder &b) const { return flatten_list(b, items); }
vk_ref ConstantMedium::flatten(FlatBuilder &b) const {
    vk_medium m; m.boundary = b.hittable(boundary); m.neg_inv_density = neg_inv_density; m.material = b.material(phase_function);
    b.media.push_back(m);
    return VK_MAKE_REF(VK_KIND_MEDIUM, b.media.size() - 1);
}
vk_ref Translate::flatten(FlatBuilder &b) const {
    vk_translate t; t.child = b.hittable(ptr); t.offset[0] = offset.x; t.offset[1] = offset.y; t.offset[2] = offset.z;
    b.translates.push_back(t);
    return VK_MAKE_REF(VK_KIND_TRANSLATE, b.translates.size() - 1);
}
Rotate::Rotate(HittableP p, int axis_, float angle) : ptr(p), axis(axis_) {  // hittable.rs:542-575, 639-672, 728-761
    sin_theta = sinf(to_radians(angle));
    cos_theta = cosf(to_radians(angle));
    AxisBB bbox = *p->bounding_box(0.0f, 1.0f);
    Vec3 mn = Vec3::new_const(INFINITY), mx = Vec3::new_const(-INFINITY);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                float x = i == 1 ? bbox.max.x : bbox.min.x;
                float y = j == 1 ? bbox.max.y : bbox.min.y;
                float z = k == 1 ? bbox.max.z : bbox.min.z;
                Vec3 tester;
                if (axis == 1) tester = Vec3(cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z);
                else if (axis == 0) tester = Vec3(x, cos_theta * y - sin_theta * z, sin_theta * y + cos_theta * z);
                else tester = Vec3(cos_theta * x - sin_theta * y, sin_theta * x + cos_theta * y, z);
                for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], tester[c]); mx[c] = fmaxf(mx[c], tester[c]); }
            }
    bb = AxisBB{mn, mx};
}
vk_ref Rotate::flatten(FlatBuilder &b) const {
    vk_rotate r; r.child = b.hittable(ptr); r.axis = (uint32_t)axis; r.sin_theta = sin_theta; r.cos_theta = cos_theta;
    b.rotates.push_back(r);
    return VK_MAKE_REF(VK_KIND_ROTATE, b.rotates.size() - 1);
}

// accel.rs:98-136
std::shared_ptr<BVHNode> BVHNode::build(std::vector<HittableP> &objects, size_t begin, size_t end) {
    uint32_t axis = gen_index(3);  // rng.gen_range(0, 3)
    size_t len = end - begin;
    if (len == 0) throw std::runtime_error("BVHNode::new on an empty slice (index out of bounds in the reference)");
    auto node = std::make_shared<BVHNode>();
    auto bbox = [](const HittableP &h) {
        auto bb = h->bounding_box(0.0f, 0.0f);
        if (!bb) throw std::runtime_error("bounding_box() is None (unwrap panics, accel.rs:93)");
        return *bb;
    };
    if (len == 1) {
        node->left = objects[begin]; node->right = objects[begin];
        node->bb = AxisBB::surrounding_box(bbox(objects[begin]), bbox(objects[begin]));
    } else if (len == 2) {
        AxisBB a_bb = bbox(objects[begin]), b_bb = bbox(objects[begin + 1]);
        size_t i1 = begin, i2 = begin + 1;
        if (a_bb.min[(int)axis] < b_bb.min[(int)axis]) { i1 = begin + 1; i2 = begin; }  // sic: larger min goes LEFT
        node->left = objects[i1]; node->right = objects[i2];
        node->bb = AxisBB::surrounding_box(bbox(objects[i1]), bbox(objects[i2]));
    } else {
        // slice::sort_by is a stable merge sort; partial_cmp().unwrap() panics on NaN
        std::vector<std::pair<float, HittableP>> keyed;
        keyed.reserve(len);
        for (size_t i = begin; i < end; i++) {
            float k = bbox(objects[i]).min[(int)axis];
            if (k != k) throw std::runtime_error("NaN in bounding box (partial_cmp unwrap panics, accel.rs:125)");
            keyed.emplace_back(k, objects[i]);
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        for (size_t i = 0; i < len; i++) objects[begin + i] = keyed[i].second;
        keyed.clear(); keyed.shrink_to_fit();
        size_t mid = begin + len / 2;
        auto l = build(objects, begin, mid);
        auto r = build(objects, mid, end);
        node->left = l; node->right = r;
        node->bb = AxisBB::surrounding_box(l->bb, r->bb);
    }
    return node;
}
static thread_local BvhBuilder g_builder = BvhBuilder::REFERENCE;
void set_bvh_builder(BvhBuilder b) { g_builder = b; }
BvhBuilder bvh_builder() { return g_builder; }

std::shared_ptr<BVHNode> BVHNode::build(std::vector<HittableP> &objects) {
    if (g_builder == BvhBuilder::SAH) {
        std::vector<HittableP> scratch(objects);
        (void)build(scratch, 0, scratch.size());          // consumes the reference's axis draws; tree discarded
        return build_sah(objects);
    }
    return build(objects, 0, objects.size());
}

// ---- binned SAH builder (not in the reference: SURVEY §8f-2).  Same node shape as BVHNode::new produces
// (two children; one object -> both children are that object; two objects -> a leaf), so the flat format,
// the oracle and the kernel need nothing new.  Traversal order is fixed (left, then right with the left
// hit's t as tmax, accel.rs:64-70), so the child with the larger surface area goes LEFT: large objects
// (a ground sphere) are hit first and their t culls the boxes behind them.
namespace {
struct SahPrim { AxisBB bb; Vec3 c; HittableP obj; };
inline float half_area(const AxisBB &b) {
    float dx = b.max.x - b.min.x, dy = b.max.y - b.min.y, dz = b.max.z - b.min.z;
    return dx * dy + dy * dz + dz * dx;
}
inline float axis_of(const Vec3 &v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

std::shared_ptr<BVHNode> sah_rec(std::vector<SahPrim> &P, size_t begin, size_t end, int depth) {
    size_t len = end - begin;
    auto node = std::make_shared<BVHNode>();
    if (len == 1) {
        node->left = P[begin].obj; node->right = P[begin].obj;
        node->bb = AxisBB::surrounding_box(P[begin].bb, P[begin].bb);
        return node;
    }
    if (len == 2) {
        size_t a = begin, b = begin + 1;
        if (half_area(P[b].bb) > half_area(P[a].bb)) std::swap(a, b);
        node->left = P[a].obj; node->right = P[b].obj;
        node->bb = AxisBB::surrounding_box(P[a].bb, P[b].bb);
        return node;
    }
    // centroid bounds
    Vec3 cmin = P[begin].c, cmax = P[begin].c;
    for (size_t i = begin + 1; i < end; i++) {
        cmin = Vec3(fminf(cmin.x, P[i].c.x), fminf(cmin.y, P[i].c.y), fminf(cmin.z, P[i].c.z));
        cmax = Vec3(fmaxf(cmax.x, P[i].c.x), fmaxf(cmax.y, P[i].c.y), fmaxf(cmax.z, P[i].c.z));
    }
    const int NB = 32;
    int best_axis = -1, best_split = 0; double best_cost = 1e300;
    if (depth < 96) {
        for (int ax = 0; ax < 3; ax++) {
            float lo = axis_of(cmin, ax), hi = axis_of(cmax, ax);
            if (!(hi > lo)) continue;
            float scale = (float)NB / (hi - lo);
            AxisBB bb[NB]; size_t cnt[NB]; bool used[NB];
            for (int b = 0; b < NB; b++) { cnt[b] = 0; used[b] = false; }
            for (size_t i = begin; i < end; i++) {
                int b = (int)((axis_of(P[i].c, ax) - lo) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                bb[b] = used[b] ? AxisBB::surrounding_box(bb[b], P[i].bb) : P[i].bb;
                used[b] = true; cnt[b]++;
            }
            // sweep: cost(split after bin s) = A(left)*n(left) + A(right)*n(right)
            double la[NB]; size_t ln[NB];
            AxisBB acc{}; bool have = false; size_t n = 0;
            for (int b = 0; b < NB; b++) {
                if (used[b]) { acc = have ? AxisBB::surrounding_box(acc, bb[b]) : bb[b]; have = true; n += cnt[b]; }
                la[b] = have ? (double)half_area(acc) : 0.0; ln[b] = n;
            }
            have = false; n = 0;
            for (int b = NB - 1; b >= 1; b--) {
                if (used[b]) { acc = have ? AxisBB::surrounding_box(acc, bb[b]) : bb[b]; have = true; n += cnt[b]; }
                if (n == 0 || ln[b - 1] == 0) continue;
                double cost = la[b - 1] * (double)ln[b - 1] + (double)half_area(acc) * (double)n;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = b; }
            }
        }
    }
    size_t mid;
    if (best_axis >= 0) {
        float lo = axis_of(cmin, best_axis), hi = axis_of(cmax, best_axis);
        float scale = (float)NB / (hi - lo);
        int ax = best_axis, sp = best_split;
        auto it = std::stable_partition(P.begin() + begin, P.begin() + end, [&](const SahPrim &q) {
            int b = (int)((axis_of(q.c, ax) - lo) * scale);
            b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
            return b < sp;
        });
        mid = (size_t)(it - P.begin());
    } else {
        mid = begin + len / 2;      // coincident centroids (or a runaway depth): split the slice in the middle
    }
    if (mid == begin || mid == end) mid = begin + len / 2;
    auto l = sah_rec(P, begin, mid, depth + 1);
    auto r = sah_rec(P, mid, end, depth + 1);
    if (half_area(r->bb) > half_area(l->bb)) std::swap(l, r);
    node->left = l; node->right = r;
    node->bb = AxisBB::surrounding_box(l->bb, r->bb);
    return node;
}
}  // namespace

std::shared_ptr<BVHNode> BVHNode::build_sah(std::vector<HittableP> &objects) {
    if (objects.empty()) throw std::runtime_error("BVHNode::new on an empty slice (index out of bounds in the reference)");
    std::vector<SahPrim> P;
    P.reserve(objects.size());
    for (auto &h : objects) {
        auto bb = h->bounding_box(0.0f, 0.0f);             // the boxes the reference builder uses (accel.rs:93)
        if (!bb) throw std::runtime_error("bounding_box() is None (unwrap panics, accel.rs:93)");
        if (bb->min.x != bb->min.x || bb->max.x != bb->max.x) throw std::runtime_error("NaN in bounding box");
        Vec3 c((bb->min.x + bb->max.x) * 0.5f, (bb->min.y + bb->max.y) * 0.5f, (bb->min.z + bb->max.z) * 0.5f);
        P.push_back(SahPrim{*bb, c, h});
    }
    return sah_rec(P, 0, P.size(), 0);
}

vk_ref BVHNode::flatten(FlatBuilder &b) const {
    vk_bvh_node n;
    n.bb_min[0] = bb.min.x; n.bb_min[1] = bb.min.y; n.bb_min[2] = bb.min.z;
    n.bb_max[0] = bb.max.x; n.bb_max[1] = bb.max.y; n.bb_max[2] = bb.max.z;
    size_t idx = b.bvh.size();
    b.bvh.push_back(n);
    vk_ref l = b.hittable(left);
    vk_ref r = b.hittable(right);
    b.bvh[idx].left = l; b.bvh[idx].right = r;
    return VK_MAKE_REF(VK_KIND_BVH, idx);
}

// main.rs:71-109
vk_camera camera_new(Vec3 lookfrom, Vec3 lookat, Vec3 vup, float vfov, float aspect_ratio, float aperture,
                     float focus_dist, float time0, float time1) {
    float theta = to_radians(vfov);
    float h = tanf(theta / 2.0f);
    float viewport_height = h * 2.0f;
    float viewport_width = aspect_ratio * viewport_height;
    Vec3 w = (lookfrom - lookat).unit_vector();
    Vec3 u = vup.cross(w).unit_vector();
    Vec3 v = w.cross(u);
    Vec3 origin = lookfrom;
    Vec3 horizontal = u * viewport_width * focus_dist;
    Vec3 vertical = v * viewport_height * focus_dist;
    Vec3 llc = origin - horizontal / 2.0f - vertical / 2.0f - w * focus_dist;
    vk_camera c;
    auto put = [](float *d, Vec3 s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    put(c.origin, origin); put(c.lower_left_corner, llc); put(c.horizontal, horizontal); put(c.vertical, vertical);
    put(c.u, u); put(c.v, v); put(c.w, w);
    c.lens_radius = aperture / 2.0f; c.time0 = time0; c.time1 = time1;
    return c;
}

std::function<bool(vk_camera &)> FixedCamera(vk_camera cam) {  // scene.rs:24-46
    auto called = std::make_shared<bool>(false);
    return [cam, called](vk_camera &out) {
        if (*called) return false;
        *called = true; out = cam; return true;
    };
}
std::function<bool(vk_camera &)> RotatingCamera(Vec3 lookat, Vec3 vup, float vfov, float aspect_ratio, float aperture,
                                                float focus_dist, float time0, float time1, float height, float angle,
                                                float radius, float incr, float limit) {  // scene.rs:48-91
    auto ang = std::make_shared<float>(angle);
    return [=](vk_camera &out) {
        if (*ang > limit) return false;
        float look_x = radius * cosf(to_radians(*ang));
        float look_z = radius * sinf(to_radians(*ang));
        out = camera_new(Vec3(look_x, height, look_z), lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time0, time1);
        *ang += incr;
        return true;
    };
}

}  // namespace vecchio
