// vecchio_host.hpp — host side ABOVE the C ABI, mirroring the reference's own interface.
//
// The reference's host code is Rust (scene.rs builders, BVHNode::new, Camera::new, main());
// there is no Rust toolchain in this image, so this is the C++ stand-in a Rust maintainer
// would read side by side with vecchio_amd/rust_shim/*.rs: same type names, same
// constructor arguments, same bounding_box() results, plus the ONE method the shim adds to
// each trait — flatten(&self, &mut FlatBuilder) -> vk_ref — because the traits offer no
// introspection (hittable.rs:33-42, material.rs:20-41,228-230).
//
// Nothing here intersects a ray: hit()/scatter()/value() live on the device
// (vecchio_amd/csrc) and, for checking, in oracle/.
#ifndef VECCHIO_HOST_HPP
#define VECCHIO_HOST_HPP

#include <cassert>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "../../include/vecchio_amd.h"
#include "../csrc/vk_math.h"

namespace vecchio {

// ------------------------------------------------------------------ seeded thread_rng()
// scene.rs / accel.rs / material.rs draw from rand::thread_rng() while building; the
// stand-in is one seeded stream per scene build.
vk::BuildRng &thread_rng();
void seed_thread_rng(uint64_t seed);
inline float gen_f32() { return vk::gen_f32(thread_rng()); }
inline float gen_range(float lo, float hi) { return vk::gen_range(thread_rng(), lo, hi); }
inline uint32_t gen_index(uint32_t n) { return vk::gen_index(thread_rng(), n); }

// ------------------------------------------------------------------ vec3.rs
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3 new_const(float v) { return Vec3(v, v, v); }
    float dot(Vec3 v) const { return x * v.x + y * v.y + z * v.z; }
    Vec3 cross(Vec3 v) const { return Vec3(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x); }
    float length2() const { return x * x + y * y + z * z; }
    float length() const { return sqrtf(length2()); }
    Vec3 unit_vector() const { float n = sqrtf(length2()); return Vec3(x / n, y / n, z / n); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    static Vec3 random() { float a = gen_f32(), b = gen_f32(), c = gen_f32(); return Vec3(a, b, c); }              // vec3.rs:68-72
    static Vec3 random_range(float mn, float mx) {                                                                  // vec3.rs:74-82
        float a = gen_range(mn, mx), b = gen_range(mn, mx), c = gen_range(mn, mx);
        return Vec3(a, b, c);
    }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline float to_radians(float deg) { return deg * (3.14159265358979323846f / 180.0f); }  // f32::to_radians

// ------------------------------------------------------------------ accel.rs:10-50
struct AxisBB {
    Vec3 min, max;
    static AxisBB surrounding_box(AxisBB a, AxisBB b) {  // accel.rs:37-49
        return AxisBB{Vec3(fminf(a.min.x, b.min.x), fminf(a.min.y, b.min.y), fminf(a.min.z, b.min.z)),
                      Vec3(fmaxf(a.max.x, b.max.x), fmaxf(a.max.y, b.max.y), fmaxf(a.max.z, b.max.z))};
    }
};

// ------------------------------------------------------------------ FlatBuilder: what flatten() pushes into
class Texture;
class Material;
class Hittable;
typedef std::shared_ptr<const Texture> TextureP;
typedef std::shared_ptr<const Material> MaterialP;
typedef std::shared_ptr<const Hittable> HittableP;

struct FlatBuilder {
    std::vector<vk_bvh_node> bvh;
    std::vector<vk_sphere> spheres;
    std::vector<vk_moving_sphere> moving_spheres;
    std::vector<vk_rect> rects;
    std::vector<vk_list> lists;
    std::vector<vk_ref> list_items;
    std::vector<vk_medium> media;
    std::vector<vk_translate> translates;
    std::vector<vk_rotate> rotates;
    std::vector<vk_material> materials;
    std::vector<vk_texture> textures;
    std::vector<vk_image> images;
    std::vector<std::vector<uint8_t>> image_storage;
    std::vector<vk_perlin> perlins;
    std::vector<vk_ref> lights;
    vk_ref world = 0;
    // Arc identity -> record, so shared Arcs (a light in world AND in lights, one material
    // on 400 boxes) flatten to one record, exactly as Arc::clone shares one object.
    std::map<const void *, uint32_t> seen_hittable, seen_material, seen_texture;

    uint32_t material(const MaterialP &m);
    uint32_t texture(const TextureP &t);
    vk_ref hittable(const HittableP &h);
    vk_scene_desc desc();  // pointers stay valid while this builder is alive and unmodified
};

// ------------------------------------------------------------------ material.rs:228-434 textures
class Texture {
  public:
    virtual ~Texture() {}
    virtual uint32_t flatten(FlatBuilder &b) const = 0;
};
class SolidColor : public Texture {  // material.rs:233-236
  public:
    Vec3 color_value;
    explicit SolidColor(Vec3 c) : color_value(c) {}
    uint32_t flatten(FlatBuilder &b) const override;
};
class Checker : public Texture {  // material.rs:244-248
  public:
    TextureP odd, even;
    Checker(TextureP o, TextureP e) : odd(o), even(e) {}
    uint32_t flatten(FlatBuilder &b) const override;
};
// the directory ImageTexture::open reads (default: $VECCHIO_ASSETS, else "assets" as in the reference)
void set_assets_dir(const std::string &dir);
std::string assets_dir();
class ImageTexture : public Texture {  // material.rs:261-280 (png decode replaced: raw RGB8 or binary PPM)
  public:
    std::vector<uint8_t> buf;
    uint32_t width = 0, height = 0;
    ImageTexture(uint32_t w, uint32_t h, std::vector<uint8_t> rgb) : buf(std::move(rgb)), width(w), height(h) {}
    static std::shared_ptr<ImageTexture> open(const std::string &path);       // ImageTexture::new("assets/x.png"): <assets_dir>/x.ppm.gz
    static std::shared_ptr<ImageTexture> from_ppm(const std::string &path);   // P6, plain or gzip'd
    uint32_t flatten(FlatBuilder &b) const override;
};
class Perlin {  // material.rs:306-377
  public:
    Vec3 random_data[256];
    uint32_t perm_x[256], perm_y[256], perm_z[256];
    Perlin();  // material.rs:357-377 (draws from thread_rng())
};
class NoiseTexture : public Texture {  // material.rs:416-428
  public:
    Perlin noise;
    float scale;
    explicit NoiseTexture(float s) : scale(s) {}
    uint32_t flatten(FlatBuilder &b) const override;
};

// ------------------------------------------------------------------ material.rs materials
class Material {
  public:
    virtual ~Material() {}
    virtual uint32_t flatten(FlatBuilder &b) const = 0;
};
class Lambertian : public Material {  // material.rs:45-48
  public:
    TextureP albedo;
    explicit Lambertian(TextureP a) : albedo(a) {}
    uint32_t flatten(FlatBuilder &b) const override;
};
class Metal : public Material {  // material.rs:111-115
  public:
    TextureP albedo; float fuzz;
    Metal(TextureP a, float f) : albedo(a), fuzz(f) {}
    uint32_t flatten(FlatBuilder &b) const override;
};
class Dielectric : public Material {  // material.rs:144-147
  public:
    float ref_idx;
    explicit Dielectric(float r) : ref_idx(r) {}
    uint32_t flatten(FlatBuilder &b) const override;
};
class DiffuseLight : public Material {  // material.rs:209-212
  public:
    TextureP emit;
    explicit DiffuseLight(TextureP e) : emit(e) {}
    uint32_t flatten(FlatBuilder &b) const override;
};
class Isotropic : public Material {  // material.rs:436-439
  public:
    TextureP albedo;
    explicit Isotropic(TextureP a) : albedo(a) {}
    uint32_t flatten(FlatBuilder &b) const override;
};
class SpecDiffuse : public Material {  // material.rs:467-472
  public:
    MaterialP specular, diffuse; float pct;
    SpecDiffuse(MaterialP s, MaterialP d, float p) : specular(s), diffuse(d), pct(p) {}
    uint32_t flatten(FlatBuilder &b) const override;
};

// ------------------------------------------------------------------ hittable.rs
class Hittable {  // trait Hittable, hittable.rs:33-42 (+ flatten)
  public:
    virtual ~Hittable() {}
    virtual std::optional<AxisBB> bounding_box(float t0, float t1) const = 0;
    virtual vk_ref flatten(FlatBuilder &b) const = 0;
};
class Sphere : public Hittable {  // hittable.rs:46-51,97-102
  public:
    Vec3 center; float radius; MaterialP material;
    Sphere(Vec3 c, float r, MaterialP m) : center(c), radius(r), material(m) {}
    std::optional<AxisBB> bounding_box(float, float) const override {
        return AxisBB{center - Vec3::new_const(radius), center + Vec3::new_const(radius)};
    }
    vk_ref flatten(FlatBuilder &b) const override;
};
class MovingSphere : public Hittable {  // hittable.rs:136-151,186-196
  public:
    Vec3 center0, center1; float time0, time1, radius; MaterialP material;
    MovingSphere(Vec3 c0, Vec3 c1, float t0, float t1, float r, MaterialP m)
        : center0(c0), center1(c1), time0(t0), time1(t1), radius(r), material(m) {}
    Vec3 center(float time) const { return center0 + (center1 - center0) * ((time - time0) / (time1 - time0)); }
    std::optional<AxisBB> bounding_box(float, float) const override {
        AxisBB bb1{center(time0) - Vec3::new_const(radius), center(time0) + Vec3::new_const(radius)};
        AxisBB bb2{center(time1) - Vec3::new_const(radius), center(time1) + Vec3::new_const(radius)};
        return AxisBB::surrounding_box(bb1, bb2);
    }
    vk_ref flatten(FlatBuilder &b) const override;
};
class Rect : public Hittable {  // hittable.rs:199-227,258-269
  public:
    float c0, c1, d0, d1, k; int axis0, axis1, axis2; MaterialP mat;
    Rect(float c0_, float c1_, float d0_, float d1_, float k_, int a0, int a1, int a2, MaterialP m)
        : c0(c0_), c1(c1_), d0(d0_), d1(d1_), k(k_), axis0(a0), axis1(a1), axis2(a2), mat(m) {}
    static std::shared_ptr<Rect> XYRect(float x0, float x1, float y0, float y1, float k,
        MaterialP m) { return std::make_shared<Rect>(x0, x1, y0, y1, k, 0, 1, 2, m); }
    static std::shared_ptr<Rect> XZRect(float x0, float x1, float z0, float z1, float k,
        MaterialP m) { return std::make_shared<Rect>(x0, x1, z0, z1, k, 0, 2, 1, m); }
    static std::shared_ptr<Rect> YZRect(float y0, float y1, float z0, float z1, float k,
        MaterialP m) { return std::make_shared<Rect>(y0, y1, z0, z1, k, 1, 2, 0, m); }
    std::optional<AxisBB> bounding_box(float, float) const override {
        Vec3 v1, v2;
        v1[axis0] = c0; v1[axis1] = d0; v1[axis2] = k - 0.0001f;
        v2[axis0] = c1; v2[axis1] = d1; v2[axis2] = k + 0.0001f;
        return AxisBB{v1, v2};
    }
    vk_ref flatten(FlatBuilder &b) const override;
};
class FlipFace : public Hittable {  // hittable.rs:294-312
  public:
    HittableP ptr;
    explicit FlipFace(HittableP p) : ptr(p) {}
    std::optional<AxisBB> bounding_box(float t0, float t1) const override { return ptr->bounding_box(t0, t1); }
    vk_ref flatten(FlatBuilder &b) const override;
};
class Boxy : public Hittable {  // hittable.rs:314-378
  public:
    Vec3 box_min, box_max; std::vector<HittableP> sides;
    Boxy(Vec3 p0, Vec3 p1, MaterialP mat);
    std::optional<AxisBB> bounding_box(float, float) const override { return AxisBB{box_min, box_max}; }
    vk_ref flatten(FlatBuilder &b) const override;
};
class HittableList : public Hittable {  // impl Hittable for Vec<Arc<HittableSS>>, hittable.rs:380-418
  public:
    std::vector<HittableP> items;
    std::optional<AxisBB> bounding_box(float t0, float t1) const override;
    vk_ref flatten(FlatBuilder &b) const override;
};
class ConstantMedium : public Hittable {  // hittable.rs:436-450,495-497
  public:
    HittableP boundary; MaterialP phase_function; float neg_inv_density;
    ConstantMedium(HittableP b, float density, TextureP albedo)
        : boundary(b), phase_function(std::make_shared<Isotropic>(albedo)), neg_inv_density(-1.0f / density) {}
    std::optional<AxisBB> bounding_box(float t0, float t1) const override { return boundary->bounding_box(t0, t1); }
    vk_ref flatten(FlatBuilder &b) const override;
};
class Translate : public Hittable {  // hittable.rs:500-504,525-531
  public:
    HittableP ptr; Vec3 offset;
    Translate(HittableP p, Vec3 o) : ptr(p), offset(o) {}
    std::optional<AxisBB> bounding_box(float t0, float t1) const override {
        auto bb = ptr->bounding_box(t0, t1);
        if (!bb) return std::nullopt;
        return AxisBB{bb->min + offset, bb->max + offset};
    }
    vk_ref flatten(FlatBuilder &b) const override;
};
class Rotate : public Hittable {  // RotateY hittable.rs:534-576, RotateX 631-673, RotateZ 720-762
  public:
    HittableP ptr; int axis; float sin_theta, cos_theta; std::optional<AxisBB> bb;
    Rotate(HittableP p, int axis_, float angle);
    std::optional<AxisBB> bounding_box(float, float) const override { return bb; }
    vk_ref flatten(FlatBuilder &b) const override;
};
inline std::shared_ptr<Rotate> RotateX(HittableP p, float angle) { return std::make_shared<Rotate>(p, 0, angle); }
inline std::shared_ptr<Rotate> RotateY(HittableP p, float angle) { return std::make_shared<Rotate>(p, 1, angle); }
inline std::shared_ptr<Rotate> RotateZ(HittableP p, float angle) { return std::make_shared<Rotate>(p, 2, angle); }

class BVHNode : public Hittable {  // accel.rs:52-56,85-136
  public:
    HittableP left, right; AxisBB bb;
    // BVHNode::new (accel.rs:98-136): random axis, stable sort by bb.min[axis], split len/2
    static std::shared_ptr<BVHNode> build(std::vector<HittableP> &objects, size_t begin, size_t end);
    static std::shared_ptr<BVHNode> build(std::vector<HittableP> &objects);
    // binned surface-area-heuristic builder over the same leaf objects (opt-in, see set_bvh_builder)
    static std::shared_ptr<BVHNode> build_sah(std::vector<HittableP> &objects);
    std::optional<AxisBB> bounding_box(float, float) const override { return bb; }
    vk_ref flatten(FlatBuilder &b) const override;
};

// Which builder BVHNode::build(objects) runs.  REFERENCE = accel.rs:98-136 (the default, and what every parity
// fixture uses).  SAH builds a better tree over the same objects; it first runs the reference builder and
// discards its tree so that the scene's random stream, hence its geometry, is identical in both modes.
enum class BvhBuilder { REFERENCE, SAH };
void set_bvh_builder(BvhBuilder b);
BvhBuilder bvh_builder();

// ------------------------------------------------------------------ main.rs:56-109 Camera::new
vk_camera camera_new(Vec3 lookfrom, Vec3 lookat, Vec3 vup, float vfov, float aspect_ratio, float aperture,
                     float focus_dist, float time0, float time1);

// ------------------------------------------------------------------ scene.rs:16-91
struct SceneConfig {
    std::vector<HittableP> world;
    std::vector<HittableP> lights;
    std::function<bool(vk_camera &)> cam_iter;  // yields the next Camera, false when exhausted
    float aspect_ratio = 1.0f;
    // defaults the reference hard-codes per git tag (main.rs:28-29,124; SURVEY §8a variants)
    uint32_t integrator = VK_INTEGRATOR_PDF;
    uint32_t background = VK_BACKGROUND_SOLID;
    Vec3 background_color;
};
std::function<bool(vk_camera &)> FixedCamera(vk_camera cam);                     // scene.rs:24-46
std::function<bool(vk_camera &)> RotatingCamera(Vec3 lookat, Vec3 vup, float vfov, float aspect_ratio, float aperture,
                                                float focus_dist, float time0, float time1, float height, float angle,
                                                float radius, float incr, float limit);  // scene.rs:48-91

SceneConfig balls_demo();            // scene.rs:93-165
SceneConfig random_spheres_demo();   // scene.rs:167-284 (HEAD: checker ground, earth, sky light, PDF integrator)
SceneConfig perlin_demo();           // scene.rs:286-338
SceneConfig bowser_demo();           // scene.rs:340-628 (PNG assets: decoded copies, ImageTexture::open)
SceneConfig cornell_box();           // scene.rs:630-730
SceneConfig final_scene();           // scene.rs:732-874
SceneConfig final_scene_nextweek();  // the same with the TheNextWeek tag's integrator (scatter + emitted, black background)
// InOneWeekend-tag variant of random_spheres_demo (sphere-only, sky background, aperture 0.1;
// BASELINE configs C1/C2) and its 1M-sphere extension (C5).  grid_half = 11 is the book scene.
SceneConfig random_spheres_iow(int grid_half);

}  // namespace vecchio
#endif
