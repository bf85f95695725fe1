// oracle.cpp — CPU oracle: recursive restatement of vecchio's per-pixel sample loop.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED at the per-sample level: the
// reference is unseeded Rust that cannot be built here; pins are statistical (Cornell
// sample image) plus closed-form unit tests.
//
// Every function follows the reference line by line — same operation order (IEEE f32,
// unfused: build with -ffp-contract=off), same draw order — and cites it:
//   Vec3            vec3.rs:1-208
//   Ray, Camera     main.rs:31-121
//   ray_color       main.rs:123-153 (PDF integrator); scatter integrator = the
//                   InOneWeekend/TheNextWeek form built on Material::scatter (not at HEAD)
//   pixel loop      main.rs:181-198
//   AxisBB, BVHNode accel.rs:10-88
//   Hittable impls  hittable.rs (Sphere 46-134, MovingSphere 136-197, Rect 199-292,
//                   FlipFace 294-312, Vec 380-434, ConstantMedium 436-498,
//                   Translate 500-532, RotateY 534-629, RotateX 631-718, RotateZ 720-807)
//   Materials       material.rs:13-226, 436-488; Textures material.rs:228-434
//   ONB, PDFs       util.rs:65-186; helpers util.rs:6-63
// rand::thread_rng() is replaced by a thread-local pointer to the current sample's
// counter-based stream (vk_math.h); draw ORDER and COUNT are the reference's.
#include "oracle.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../vecchio_amd/csrc/vk_math.h"

namespace {

thread_local std::string g_err;
thread_local vk::Rng *g_rng = nullptr;          // the "thread_rng()" of the current sample
thread_local oracle_counters g_cnt;             // per-thread visit counters
thread_local uint64_t g_nonfinite_segment = 0;  // 1 while the segment's ray has a NaN / infinite direction or origin (counters only)
thread_local bool g_panic = false;              // a reference panic!/unwrap was reached

inline vk::Rng &thread_rng() { return *g_rng; }
inline float gen_f32() { uint32_t c0 = g_rng->ctr; float v = vk::gen_f32(*g_rng); g_cnt.n_draws += g_rng->ctr - c0; return v; }
inline float gen_range(float lo, float hi) { uint32_t c0 = g_rng->ctr; float v = vk::gen_range(*g_rng, lo, hi);
    g_cnt.n_draws += g_rng->ctr - c0; return v; }
inline uint32_t gen_index(uint32_t n) { uint32_t c0 = g_rng->ctr; uint32_t v = vk::gen_index(*g_rng, n); g_cnt.n_draws += g_rng->ctr - c0;
    return v; }

const float PI = 3.14159265358979323846f;  // std::f32::consts::PI

// ---------------------------------------------------------------------------- vec3.rs
struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3 new_const(float v) { return Vec3(v, v, v); }                    // vec3.rs:11-13
    float dot(Vec3 v) const { return x * v.x + y * v.y + z * v.z; }              // vec3.rs:19-21
    Vec3 cross(Vec3 v) const {                                                   // vec3.rs:23-29
        return Vec3(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x);
    }
    float length2() const { return x * x + y * y + z * z; }                      // vec3.rs:31-33
    float length() const { return sqrtf(length2()); }                            // vec3.rs:35-37
    Vec3 unit_vector() const {                                                   // vec3.rs:39-42
        float norm = sqrtf(length2());
        return Vec3(x / norm, y / norm, z / norm);
    }
    static float clamp(float v, float mn, float mx) {                            // vec3.rs:44-52
        if (v < mn) return mn;
        else if (v > mx) return mx;
        else return v;
    }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }           // vec3.rs:181-202
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    static Vec3 random() {                                                       // vec3.rs:68-72
        float a = gen_f32(), b = gen_f32(), c = gen_f32();
        return Vec3(a, b, c);
    }
    static Vec3 random_range(float mn, float mx) {                               // vec3.rs:74-82
        float a = gen_range(mn, mx), b = gen_range(mn, mx), c = gen_range(mn, mx);
        return Vec3(a, b, c);
    }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }  // vec3.rs:85-94
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }  // vec3.rs:96-105
inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }        // vec3.rs:107-116
inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }        // vec3.rs:118-127
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }  // vec3.rs:129-138
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }                          // vec3.rs:140-149

// ---------------------------------------------------------------------------- util.rs
inline float fmin_(float a, float b) { return fminf(a, b); }  // util.rs:6-8   (f32::min ignores NaN)
inline float fmax_(float a, float b) { return fmaxf(a, b); }  // util.rs:10-12

inline Vec3 reflect(Vec3 v, Vec3 n) { return v - n * v.dot(n) * 2.0f; }  // util.rs:14-16

inline Vec3 refract(Vec3 uv, Vec3 n, float etai_over_etat) {  // util.rs:18-23
    float cos_theta = -uv.dot(n);
    Vec3 r_out_parallel = (uv + n * cos_theta) * etai_over_etat;
    Vec3 r_out_perp = n * -sqrtf(1.0f - r_out_parallel.length2());
    return r_out_parallel + r_out_perp;
}

inline float schlick(float cosine, float ref_idx) {  // util.rs:25-29
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * vk::pow5f_(1.0f - cosine);
}

inline Vec3 random_in_unit_sphere() {  // util.rs:31-39
    for (;;) {
        Vec3 p = Vec3::random_range(-1.0f, 1.0f);
        if (p.length2() >= 1.0f) continue;
        return p;
    }
}

inline Vec3 random_in_unit_disk() {  // util.rs:41-50
    for (;;) {
        float a = gen_range(-1.0f, 1.0f);
        float b = gen_range(-1.0f, 1.0f);
        Vec3 p(a, b, 0.0f);
        if (p.length2() >= 1.0f) continue;
        return p;
    }
}

inline Vec3 random_cosine_direction() {  // util.rs:52-63
    float r1 = gen_f32();
    float r2 = gen_f32();
    float z = sqrtf(1.0f - r2);
    float phi = 2.0f * r1 * PI;
    float x = vk::cosf_(phi) * sqrtf(r2);
    float y = vk::sinf_(phi) * sqrtf(r2);
    return Vec3(x, y, z);
}

struct ONB {  // util.rs:65-111
    Vec3 u, v, w;
    Vec3 local(Vec3 a) const { return u * a.x + v * a.y + w * a.z; }  // util.rs:95-97
    static ONB new_from_w(Vec3 n) {                                   // util.rs:99-110
        ONB o;
        o.w = n.unit_vector();
        Vec3 a = (fabsf(o.w.x) > 0.9f) ? Vec3(0.0f, 1.0f, 0.0f) : Vec3(1.0f, 0.0f, 0.0f);
        o.v = o.w.cross(a).unit_vector();
        o.u = o.w.cross(o.v);
        return o;
    }
};

// ---------------------------------------------------------------------------- main.rs:31-54
struct Ray {
    Vec3 origin, direction;
    float time;
    Ray() : time(0) {}
    Ray(Vec3 o, Vec3 d) : origin(o), direction(d), time(0.0f) {}               // Ray::new main.rs:39-41
    Ray(Vec3 o, Vec3 d, float t) : origin(o), direction(d), time(t) {}         // new_with_time main.rs:43-49
    Vec3 at(float t) const { return origin + direction * t; }                  // main.rs:51-53
};

struct Material;
struct HitRec {  // hittable.rs:11-31
    Vec3 p, normal;
    float t, u, v;
    bool front;
    const Material *material;
    uint32_t material_index;
    void set_face_normal(const Ray &r, Vec3 outward_normal) {  // hittable.rs:23-30
        front = r.direction.dot(outward_normal) < 0.0f;
        normal = front ? outward_normal : -outward_normal;
    }
};

// ---------------------------------------------------------------------------- accel.rs:10-50
struct AxisBB {
    Vec3 min, max;
    bool hit(const Ray &r, float tmin, float tmax) const {  // accel.rs:16-35
        g_cnt.n_aabb++;
        g_cnt.n_aabb_nonfinite += g_nonfinite_segment;
        float tmin_local = tmin;
        float tmax_local = tmax;
        for (int a = 0; a < 3; a++) {
            float t0 = fmin_((min[a] - r.origin[a]) / r.direction[a], (max[a] - r.origin[a]) / r.direction[a]);
            float t1 = fmax_((min[a] - r.origin[a]) / r.direction[a], (max[a] - r.origin[a]) / r.direction[a]);
            tmin_local = fmax_(t0, tmin_local);
            tmax_local = fmin_(t1, tmax_local);
            if (tmax_local <= tmin_local) return false;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------- hittable.rs:33-44
struct Hittable {
    virtual ~Hittable() {}
    virtual bool hit(const Ray &r, float tmin, float tmax, HitRec &rec) const = 0;
    virtual float pdf_value(Vec3, Vec3) const { return 0.0f; }                  // hittable.rs:36-38
    virtual Vec3 random(Vec3) const { return Vec3(1.0f, 0.0f, 0.0f); }         // hittable.rs:39-41
};
typedef std::shared_ptr<Hittable> HittableP;

// ---------------------------------------------------------------------------- textures, material.rs:228-434
struct Texture {
    virtual ~Texture() {}
    virtual Vec3 value(float u, float v, Vec3 p) const = 0;
};

struct SolidColor : Texture {  // material.rs:233-242
    Vec3 color_value;
    Vec3 value(float, float, Vec3) const override { return color_value; }
};

struct Checker : Texture {  // material.rs:244-259
    const Texture *odd, *even;
    Vec3 value(float u, float v, Vec3 p) const override {
        float sins = vk::sinf_(10.0f * p.x) * vk::sinf_(10.0f * p.y) * vk::sinf_(10.0f * p.z);
        if (sins < 0.0f) return odd->value(u, v, p);
        else return even->value(u, v, p);
    }
};

struct ImageTexture : Texture {  // material.rs:261-304
    const uint8_t *buf;
    size_t width, height;
    Vec3 value(float u, float v, Vec3) const override {
        g_cnt.n_texel++;
        u = Vec3::clamp(u, 0.0f, 1.0f);
        v = 1.0f - Vec3::clamp(v, 0.0f, 1.0f);
        size_t i = vk::sat_u32(u * (float)width);
        size_t j = vk::sat_u32(v * (float)height);
        if (i >= width) i = width - 1;
        if (j >= height) j = height - 1;
        size_t buf_start = j * width * 3 + i * 3;
        const uint8_t *pix = buf + buf_start;
        float color_scale = 1.0f / 255.0f;
        return Vec3(color_scale * (float)pix[0], color_scale * (float)pix[1], color_scale * (float)pix[2]);
    }
};

inline float perlin_interp(const Vec3 c[2][2][2], float u, float v, float w) {  // material.rs:331-352
    float accum = 0.0f;
    float uu = u * u * (3.0f - 2.0f * u);
    float vv = v * v * (3.0f - 2.0f * v);
    float ww = w * w * (3.0f - 2.0f * w);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                float fi = (float)i, fj = (float)j, fk = (float)k;
                Vec3 weight_v(u - fi, v - fj, w - fk);
                accum += (fi * uu + (1.0f - fi) * (1.0f - uu)) * (fj * vv + (1.0f - fj) * (1.0f - vv)) *
                         (fk * ww + (1.0f - fk) * (1.0f - ww)) * c[i][j][k].dot(weight_v);
            }
    return accum;
}

struct Perlin {  // material.rs:306-414
    Vec3 random_data[256];
    uint32_t perm_x[256], perm_y[256], perm_z[256];
    float noise(Vec3 p) const {  // material.rs:392-413
        g_cnt.n_perlin++;
        float u = p.x - floorf(p.x);
        float v = p.y - floorf(p.y);
        float w = p.z - floorf(p.z);
        Vec3 c[2][2][2];
        uint32_t i = vk::usize_low8(floorf(p.x));  // `as usize` saturates; only (i+d)&255 is used
        uint32_t j = vk::usize_low8(floorf(p.y));
        uint32_t k = vk::usize_low8(floorf(p.z));
        for (uint32_t di = 0; di < 2; di++)
            for (uint32_t dj = 0; dj < 2; dj++)
                for (uint32_t dk = 0; dk < 2; dk++)
                    c[di][dj][dk] = random_data[perm_x[(i + di) & 255] ^ perm_y[(j + dj) & 255] ^ perm_z[(k + dk) & 255]];
        return perlin_interp(c, u, v, w);
    }
    float turb(Vec3 p, int depth) const {  // material.rs:379-390
        float accum = 0.0f;
        Vec3 temp_p = p;
        float weight = 1.0f;
        for (int i = 0; i < depth; i++) {
            accum += weight * noise(temp_p);
            weight *= 0.5f;
            temp_p = temp_p * 2.0f;
        }
        return fabsf(accum);
    }
};

struct NoiseTexture : Texture {  // material.rs:416-434
    Perlin noise;
    float scale;
    Vec3 value(float, float, Vec3 p) const override {
        return Vec3::new_const(1.0f) * 0.5f * (1.0f + vk::sinf_(scale * p.z + 10.0f * noise.turb(p, 7)));
    }
};

// ---------------------------------------------------------------------------- PDFs, util.rs:113-186
struct PDF {
    virtual ~PDF() {}
    virtual float value(Vec3 direction) const = 0;
    virtual Vec3 generate() const = 0;
};

struct CosinePDF : PDF {  // util.rs:121-147
    ONB uvw;
    explicit CosinePDF(Vec3 w) : uvw(ONB::new_from_w(w)) {}
    float value(Vec3 direction) const override {
        float cos = direction.unit_vector().dot(uvw.w);
        if (cos <= 0.0f) return 0.0f;
        else return cos / PI;
    }
    Vec3 generate() const override { return uvw.local(random_cosine_direction()); }
};

struct HittablePDF : PDF {  // util.rs:149-162
    const Hittable *ptr;
    Vec3 o;
    HittablePDF(const Hittable *p, Vec3 o_) : ptr(p), o(o_) {}
    float value(Vec3 direction) const override { return ptr->pdf_value(o, direction); }
    Vec3 generate() const override { return ptr->random(o); }
};

struct MixturePDF : PDF {  // util.rs:164-186
    const PDF *ptr1; float f1; const PDF *ptr2; float f2;
    MixturePDF(const PDF *p1, float f1_, const PDF *p2, float f2_) : ptr1(p1), f1(f1_), ptr2(p2), f2(f2_) {}
    float value(Vec3 direction) const override { return f1 * ptr1->value(direction) + f2 * ptr2->value(direction); }
    Vec3 generate() const override {
        if (gen_f32() < f1) return ptr1->generate();
        else return ptr2->generate();
    }
};

// ---------------------------------------------------------------------------- materials, material.rs:13-226,436-488
struct ScatterRec {  // material.rs:13-18
    bool has_specular = false;
    Ray specular_ray;
    Vec3 attenuation;
    std::shared_ptr<PDF> pdf;
};

struct Material {
    virtual ~Material() {}
    // material.rs:21-28 default: via scatter_with_pdf, unwrap() of specular_ray
    virtual bool scatter(const Ray &r, const HitRec &rec, Vec3 &attenuation, Ray &scattered) const {
        ScatterRec srec;
        if (scatter_with_pdf(r, rec, srec)) {
            if (!srec.has_specular) { g_panic = true; return false; }  // Option::unwrap() on None
            attenuation = srec.attenuation;
            scattered = srec.specular_ray;
            return true;
        }
        return false;
    }
    virtual bool scatter_with_pdf(const Ray &, const HitRec &, ScatterRec &) const { return false; }  // material.rs:30-32
    virtual float scattering_pdf(const Ray &, const HitRec &, const Ray &) const { return 0.0f; }    // material.rs:34-36
    virtual Vec3 emitted(const HitRec &, float, float, Vec3) const { return Vec3::new_const(0.0f); } // material.rs:38-40
};

struct Lambertian : Material {  // material.rs:45-109
    const Texture *albedo;
    static Vec3 random() {  // material.rs:51-58
        float a = gen_range(0.0f, 2.0f * PI);
        float z = gen_range(-1.0f, 1.0f);
        float r = sqrtf(1.0f - z * z);
        return Vec3(r * vk::cosf_(a), r * vk::sinf_(a), z);
    }
    bool scatter(const Ray &r, const HitRec &rec, Vec3 &attenuation, Ray &scattered) const override {  // material.rs:85-90
        Vec3 scatter_direction = rec.normal + Lambertian::random();
        scattered = Ray(rec.p, scatter_direction, r.time);
        attenuation = albedo->value(rec.u, rec.v, rec.p);
        return true;
    }
    bool scatter_with_pdf(const Ray &, const HitRec &rec, ScatterRec &srec) const override {  // material.rs:92-98
        srec.has_specular = false;
        srec.attenuation = albedo->value(rec.u, rec.v, rec.p);
        srec.pdf = std::make_shared<CosinePDF>(rec.normal);
        return true;
    }
    float scattering_pdf(const Ray &, const HitRec &rec, const Ray &s) const override {  // material.rs:100-108
        float cos = rec.normal.dot(s.direction.unit_vector());
        if (cos < 0.0f) return 0.0f;
        else return cos / PI;
    }
};

struct Metal : Material {  // material.rs:111-142
    const Texture *albedo;
    float fuzz;
    bool scatter(const Ray &r, const HitRec &rec, Vec3 &attenuation, Ray &scattered) const override {  // material.rs:118-132
        Vec3 reflected = reflect(r.direction.unit_vector(), rec.normal);
        scattered = Ray(rec.p, reflected + random_in_unit_sphere() * fuzz, r.time);
        attenuation = albedo->value(rec.u, rec.v, rec.p);
        if (scattered.direction.dot(rec.normal) > 0.0f) return true;
        else return false;
    }
    bool scatter_with_pdf(const Ray &r, const HitRec &rec, ScatterRec &srec) const override {  // material.rs:134-141
        Vec3 reflected = reflect(r.direction.unit_vector(), rec.normal);
        srec.has_specular = true;
        srec.specular_ray = Ray(rec.p, reflected + random_in_unit_sphere() * fuzz);  // Ray::new: time 0
        srec.attenuation = albedo->value(rec.u, rec.v, rec.p);
        srec.pdf = std::make_shared<CosinePDF>(rec.normal);
        return true;
    }
};

struct Dielectric : Material {  // material.rs:144-207
    float ref_idx;
    bool scatter(const Ray &r, const HitRec &rec, Vec3 &attenuation, Ray &scattered) const override {  // material.rs:150-175
        attenuation = Vec3::new_const(1.0f);
        float etai_over_etat = rec.front ? 1.0f / ref_idx : ref_idx;
        Vec3 unit_direction = r.direction.unit_vector();
        float cos_theta = fminf((-unit_direction).dot(rec.normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        if (etai_over_etat * sin_theta > 1.0f) {
            Vec3 reflected = reflect(unit_direction, rec.normal);
            scattered = Ray(rec.p, reflected, r.time);
            return true;
        }
        float reflect_prob = schlick(cos_theta, etai_over_etat);
        if (gen_f32() < reflect_prob) {
            Vec3 reflected = reflect(unit_direction, rec.normal);
            scattered = Ray(rec.p, reflected, r.time);
            return true;
        }
        Vec3 refracted = refract(unit_direction, rec.normal, etai_over_etat);
        scattered = Ray(rec.p, refracted, r.time);
        return true;
    }
    bool scatter_with_pdf(const Ray &r, const HitRec &rec, ScatterRec &srec) const override {  // material.rs:177-206
        srec.has_specular = false;
        srec.attenuation = Vec3::new_const(1.0f);
        srec.pdf = std::make_shared<CosinePDF>(rec.normal);
        float etai_over_etat = rec.front ? 1.0f / ref_idx : ref_idx;
        Vec3 unit_direction = r.direction.unit_vector();
        float cos_theta = fminf((-unit_direction).dot(rec.normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        if (etai_over_etat * sin_theta > 1.0f) {
            Vec3 reflected = reflect(unit_direction, rec.normal);
            srec.has_specular = true;
            srec.specular_ray = Ray(rec.p, reflected, r.time);
            return true;
        }
        float reflect_prob = schlick(cos_theta, etai_over_etat);
        if (gen_f32() < reflect_prob) {
            Vec3 reflected = reflect(unit_direction, rec.normal);
            srec.has_specular = true;
            srec.specular_ray = Ray(rec.p, reflected, r.time);
            return true;
        }
        Vec3 refracted = refract(unit_direction, rec.normal, etai_over_etat);
        srec.has_specular = true;
        srec.specular_ray = Ray(rec.p, refracted, r.time);
        return true;
    }
};

struct DiffuseLight : Material {  // material.rs:209-226
    const Texture *emit;
    bool scatter(const Ray &, const HitRec &, Vec3 &, Ray &) const override { return false; }  // material.rs:215-217
    Vec3 emitted(const HitRec &rec, float u, float v, Vec3 p) const override {                 // material.rs:218-225
        if (rec.front) return emit->value(u, v, p);
        else return Vec3::new_const(0.0f);
    }
};

struct Isotropic : Material {  // material.rs:436-465
    const Texture *albedo;
    bool scatter(const Ray &r, const HitRec &rec, Vec3 &attenuation, Ray &scattered) const override {  // material.rs:442-446
        scattered = Ray(rec.p, random_in_unit_sphere(), r.time);
        attenuation = albedo->value(rec.u, rec.v, rec.p);
        return true;
    }
    bool scatter_with_pdf(const Ray &, const HitRec &rec, ScatterRec &srec) const override {  // material.rs:448-454
        srec.has_specular = false;
        srec.attenuation = albedo->value(rec.u, rec.v, rec.p);
        srec.pdf = std::make_shared<CosinePDF>(rec.normal);
        return true;
    }
    float scattering_pdf(const Ray &, const HitRec &rec, const Ray &s) const override {  // material.rs:456-464
        float cos = rec.normal.dot(s.direction.unit_vector());
        if (cos < 0.0f) return 0.0f;
        else return cos / PI;
    }
};

struct SpecDiffuse : Material {  // material.rs:467-488
    const Material *specular, *diffuse;
    float pct;
    bool scatter_with_pdf(const Ray &r, const HitRec &rec, ScatterRec &srec) const override {  // material.rs:475-483
        if (gen_f32() < pct) return specular->scatter_with_pdf(r, rec, srec);
        else return diffuse->scatter_with_pdf(r, rec, srec);
    }
    float scattering_pdf(const Ray &r, const HitRec &rec, const Ray &s) const override {  // material.rs:485-487
        return diffuse->scattering_pdf(r, rec, s);
    }
};

// ---------------------------------------------------------------------------- hittable.rs
inline void spherical(Vec3 p, float &u, float &v) {  // hittable.rs:54-61
    float pi = PI;
    float phi = vk::atan2f_(p.z, p.x);
    float theta = vk::asinf_(p.y);
    u = 1.0f - ((phi + pi) / (2.0f * pi));
    v = (theta + pi / 2.0f) / pi;
}

inline Vec3 random_to_sphere(float radius, float distance_squared) {  // hittable.rs:123-134 (quirk: (1-z*z), no sqrt)
    float r1 = gen_f32();
    float r2 = gen_f32();
    float z = 1.0f + r2 * (sqrtf(1.0f - radius * radius / distance_squared) - 1.0f);
    float phi = 2.0f * PI * r1;
    float x = vk::cosf_(phi) * (1.0f - z * z);
    float y = vk::sinf_(phi) * (1.0f - z * z);
    return Vec3(x, y, z);
}

struct Sphere : Hittable {  // hittable.rs:46-121
    Vec3 center; float radius; const Material *material; uint32_t mi;
    bool hit(const Ray &r, float tmin, float tmax, HitRec &ret) const override {  // hittable.rs:65-95
        g_cnt.n_sphere++;
        g_cnt.n_sphere_nonfinite += g_nonfinite_segment;
        Vec3 oc = r.origin - center;
        float a = r.direction.length2();
        float half_b = oc.dot(r.direction);
        float c = oc.length2() - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant > 0.0f) {
            float root = sqrtf(discriminant);
            float temps[2] = {(-half_b - root) / a, (-half_b + root) / a};
            for (int i = 0; i < 2; i++) {
                float temp = temps[i];
                if (tmin < temp && temp < tmax) {
                    ret.p = r.at(temp);
                    ret.normal = (r.at(temp) - center) / radius;
                    ret.t = temp; ret.u = 0.0f; ret.v = 0.0f; ret.front = false;
                    ret.material = material; ret.material_index = mi;
                    ret.set_face_normal(r, (ret.p - center) / radius);
                    spherical((ret.p - center) / radius, ret.u, ret.v);
                    return true;
                }
            }
        }
        return false;
    }
    float pdf_value(Vec3 o, Vec3 d) const override {  // hittable.rs:104-113
        HitRec tmp;
        if (hit(Ray(o, d), 0.001f, INFINITY, tmp)) {
            float cos_theta_max = sqrtf(1.0f - radius * radius / (center - o).length2());
            float solid_angle = 2.0f * PI * (1.0f - cos_theta_max);
            return 1.0f / solid_angle;
        } else return 0.0f;
    }
    Vec3 random(Vec3 o) const override {  // hittable.rs:115-120
        Vec3 direction = center - o;
        float distance_squared = direction.length2();
        ONB uvw = ONB::new_from_w(direction);
        return uvw.local(random_to_sphere(radius, distance_squared));
    }
};

struct MovingSphere : Hittable {  // hittable.rs:136-197
    Vec3 center0, center1; float time0, time1, radius; const Material *material; uint32_t mi;
    Vec3 center(float time) const {  // hittable.rs:147-150
        return center0 + (center1 - center0) * ((time - time0) / (time1 - time0));
    }
    bool hit(const Ray &r, float tmin, float tmax, HitRec &ret) const override {  // hittable.rs:154-184
        g_cnt.n_moving++;
        Vec3 oc = r.origin - center(r.time);
        float a = r.direction.length2();
        float half_b = oc.dot(r.direction);
        float c = oc.length2() - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant > 0.0f) {
            float root = sqrtf(discriminant);
            float temps[2] = {(-half_b - root) / a, (-half_b + root) / a};
            for (int i = 0; i < 2; i++) {
                float temp = temps[i];
                if (tmin < temp && temp < tmax) {
                    ret.p = r.at(temp);
                    ret.normal = (r.at(temp) - center(r.time)) / radius;
                    ret.t = temp; ret.u = 0.0f; ret.v = 0.0f; ret.front = false;
                    ret.material = material; ret.material_index = mi;
                    ret.set_face_normal(r, (ret.p - center(r.time)) / radius);
                    spherical((ret.p - center(r.time)) / radius, ret.u, ret.v);
                    return true;
                }
            }
        }
        return false;
    }
};

struct Rect : Hittable {  // hittable.rs:199-292
    float c0, c1, d0, d1, k; int axis0, axis1, axis2; const Material *mat; uint32_t mi;
    bool hit(const Ray &r, float tmin, float tmax, HitRec &ret) const override {  // hittable.rs:230-256
        g_cnt.n_rect++;
        float t = (k - r.origin[axis2]) / r.direction[axis2];
        if (t < tmin || t > tmax) return false;
        float a = r.origin[axis0] + t * r.direction[axis0];
        float b = r.origin[axis1] + t * r.direction[axis1];
        if (a < c0 || a > c1 || b < d0 || b > d1) return false;
        Vec3 outward_normal = Vec3::new_const(0.0f);
        outward_normal[axis2] = 1.0f;
        float u = (a - c0) / (c1 - c0);
        float v = (b - d0) / (d1 - d0);
        ret.p = r.at(t); ret.normal = Vec3::new_const(0.0f); ret.t = t; ret.u = u; ret.v = v; ret.front = false;
        ret.material = mat; ret.material_index = mi;
        ret.set_face_normal(r, outward_normal);
        return true;
    }
    float pdf_value(Vec3 origin, Vec3 v) const override {  // hittable.rs:271-282
        HitRec rec;
        if (hit(Ray(origin, v), 0.001f, INFINITY, rec)) {
            float area = (c1 - c0) * (d1 - d0);
            float distance_squared = rec.t * rec.t * v.length2();
            float cosine = fabsf(v.dot(rec.normal)) / v.length();
            return distance_squared / (cosine * area);
        } else return 0.0f;
    }
    Vec3 random(Vec3 origin) const override {  // hittable.rs:284-292
        Vec3 random_point = Vec3::new_const(0.0f);
        random_point[axis0] = gen_range(c0, c1);
        random_point[axis1] = gen_range(d0, d1);
        random_point[axis2] = k;
        return random_point - origin;
    }
};

struct FlipFace : Hittable {  // hittable.rs:294-312 (pdf_value/random NOT forwarded: trait defaults)
    HittableP ptr;
    bool hit(const Ray &r, float tmin, float tmax, HitRec &rec) const override {
        if (ptr->hit(r, tmin, tmax, rec)) { rec.front = !rec.front; return true; }
        return false;
    }
};

struct HittableList : Hittable {  // impl Hittable for Vec<Arc<HittableSS>>, hittable.rs:380-434 (Boxy forwards to it, 362-377)
    std::vector<HittableP> items;
    bool hit(const Ray &r, float tmin, float tmax, HitRec &out) const override {  // hittable.rs:381-394
        float closest_dist = tmax;
        bool found = false;
        for (const auto &w : items) {
            HitRec rec;
            if (w->hit(r, tmin, closest_dist, rec)) {
                if (rec.t < closest_dist) { closest_dist = rec.t; out = rec; found = true; }
            }
        }
        return found;
    }
    float pdf_value(Vec3 o, Vec3 v) const override {  // hittable.rs:420-427
        float weight = 1.0f / (float)items.size();
        float sum = 0.0f;
        for (const auto &obj : items) sum += weight * obj->pdf_value(o, v);
        return sum;
    }
    Vec3 random(Vec3 o) const override {  // hittable.rs:429-433 (choose -> gen_index)
        if (items.empty()) { g_panic = true; return Vec3(1.0f, 0.0f, 0.0f); }  // unwrap() on None
        uint32_t i = gen_index((uint32_t)items.size());
        return items[i]->random(o);
    }
};

struct ConstantMedium : Hittable {  // hittable.rs:436-498
    HittableP boundary; const Material *phase_function; uint32_t mi; float neg_inv_density;
    bool hit(const Ray &r, float tmin, float tmax, HitRec &out) const override {  // hittable.rs:453-493
        g_cnt.n_medium++;
        HitRec rec1, rec2;
        if (boundary->hit(r, -INFINITY, INFINITY, rec1)) {
            if (boundary->hit(r, rec1.t + 0.0001f, INFINITY, rec2)) {
                if (rec1.t < tmin) rec1.t = tmin;
                if (rec2.t > tmax) rec2.t = tmax;
                if (rec1.t >= rec2.t) return false;
                if (rec1.t < 0.0f) rec1.t = 0.0f;
                float ray_length = r.direction.length();
                float distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
                float hit_distance = neg_inv_density * vk::logf_(gen_f32());
                if (hit_distance > distance_inside_boundary) return false;
                float t = rec1.t + hit_distance / ray_length;
                out.p = r.at(t); out.normal = Vec3(1.0f, 0.0f, 0.0f); out.t = t;
                out.u = rec1.u; out.v = rec1.v; out.front = true;
                out.material = phase_function; out.material_index = mi;
                return true;
            }
        }
        return false;
    }
};

struct Translate : Hittable {  // hittable.rs:500-532
    HittableP ptr; Vec3 offset;
    bool hit(const Ray &r, float tmin, float tmax, HitRec &out) const override {  // hittable.rs:507-524
        g_cnt.n_xform++;
        Ray moved_r(r.origin - offset, r.direction, r.time);
        HitRec rec;
        if (ptr->hit(moved_r, tmin, tmax, rec)) {
            out = rec;
            out.p = rec.p + offset;
            out.set_face_normal(moved_r, rec.normal);
            return true;
        }
        return false;
    }
};

struct Rotate : Hittable {  // RotateY hittable.rs:578-624, RotateX 675-713, RotateZ 764-802
    HittableP ptr; int axis; float sin_theta, cos_theta;
    bool hit(const Ray &r, float tmin, float tmax, HitRec &out) const override {
        g_cnt.n_xform++;
        Vec3 origin = r.origin, direction = r.direction;
        if (axis == 1) {  // hittable.rs:591-595
            origin.x = cos_theta * r.origin.x - sin_theta * r.origin.z;
            origin.z = sin_theta * r.origin.x + cos_theta * r.origin.z;
            direction.x = cos_theta * r.direction.x - sin_theta * r.direction.z;
            direction.z = sin_theta * r.direction.x + cos_theta * r.direction.z;
        } else if (axis == 0) {  // hittable.rs:680-684
            origin.y = cos_theta * r.origin.y + sin_theta * r.origin.z;
            origin.z = -sin_theta * r.origin.y + cos_theta * r.origin.z;
            direction.y = cos_theta * r.direction.y + sin_theta * r.direction.z;
            direction.z = -sin_theta * r.direction.y + cos_theta * r.direction.z;
        } else {  // hittable.rs:769-773
            origin.x = cos_theta * r.origin.x + sin_theta * r.origin.y;
            origin.y = -sin_theta * r.origin.x + cos_theta * r.origin.y;
            direction.x = cos_theta * r.direction.x + sin_theta * r.direction.y;
            direction.y = -sin_theta * r.direction.x + cos_theta * r.direction.y;
        }
        Ray rotated_r(origin, direction, r.time);
        HitRec rec;
        if (ptr->hit(rotated_r, tmin, tmax, rec)) {
            Vec3 p = rec.p, normal = rec.normal;
            if (axis == 1) {  // hittable.rs:603-607
                p.x = cos_theta * rec.p.x + sin_theta * rec.p.z;
                p.z = -sin_theta * rec.p.x + cos_theta * rec.p.z;
                normal.x = cos_theta * rec.normal.x + sin_theta * rec.normal.z;
                normal.z = -sin_theta * rec.normal.x + cos_theta * rec.normal.z;
            } else if (axis == 0) {  // hittable.rs:692-696
                p.y = cos_theta * rec.p.y - sin_theta * rec.p.z;
                p.z = sin_theta * rec.p.y + cos_theta * rec.p.z;
                normal.y = cos_theta * rec.normal.y - sin_theta * rec.normal.z;
                normal.z = sin_theta * rec.normal.y + cos_theta * rec.normal.z;
            } else {  // hittable.rs:781-785
                p.x = cos_theta * rec.p.x - sin_theta * rec.p.y;
                p.y = sin_theta * rec.p.x + cos_theta * rec.p.y;
                normal.x = cos_theta * rec.normal.x - sin_theta * rec.normal.y;
                normal.y = sin_theta * rec.normal.x + cos_theta * rec.normal.y;
            }
            out = rec;
            out.p = p;
            out.set_face_normal(rotated_r, normal);  // quirk kept: object-space ray, world-space normal
            return true;
        }
        return false;
    }
};

struct BVHNode : Hittable {  // accel.rs:52-88
    HittableP left, right; AxisBB bb;
    bool hit(const Ray &r, float tmin, float tmax, HitRec &out) const override {  // accel.rs:59-83
        if (!bb.hit(r, tmin, tmax)) return false;
        HitRec rec_left, rec_right;
        bool hl = left->hit(r, tmin, tmax, rec_left);
        float tmax_new = hl ? rec_left.t : tmax;
        bool hr = right->hit(r, tmin, tmax_new, rec_right);
        if (hl && hr) {
            if (rec_left.t < rec_right.t) out = rec_left;
            else out = rec_right;
            return true;
        } else if (hl) { out = rec_left; return true; }
        else if (hr) { out = rec_right; return true; }
        return false;
    }
};

// ---------------------------------------------------------------------------- scene graph -> object tree
struct Scene {
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    HittableP world;
    std::shared_ptr<HittableList> lights;
    const vk_scene_desc *d = nullptr;
    std::vector<HittableP> memo[16];

    bool fail(const std::string &m) { g_err = m; return false; }

    HittableP build_ref(vk_ref ref, int depth) {
        if (depth > 4096) { g_err = "scene graph too deep / cyclic"; return nullptr; }
        uint32_t kind = VK_REF_KIND(ref), idx = VK_REF_INDEX(ref);
        bool flip = (ref & VK_REF_FLIP) != 0;
        HittableP base = build_unflipped(kind, idx, depth);
        if (!base) return nullptr;
        if (flip) {
            auto f = std::make_shared<FlipFace>();
            f->ptr = base;
            return f;
        }
        return base;
    }

    HittableP build_unflipped(uint32_t kind, uint32_t idx, int depth) {
        if (kind >= 16) return nullptr;
        auto &m = memo[kind];
        if (idx < m.size() && m[idx]) return m[idx];
        HittableP out;
        switch (kind) {
            case VK_KIND_BVH: {
                if (idx >= d->n_bvh) { g_err = "bvh index out of range"; return nullptr; }
                const vk_bvh_node &n = d->bvh[idx];
                auto b = std::make_shared<BVHNode>();
                b->bb.min = Vec3(n.bb_min[0], n.bb_min[1], n.bb_min[2]);
                b->bb.max = Vec3(n.bb_max[0], n.bb_max[1], n.bb_max[2]);
                b->left = build_ref(n.left, depth + 1);
                b->right = build_ref(n.right, depth + 1);
                if (!b->left || !b->right) return nullptr;
                out = b;
                break;
            }
            case VK_KIND_SPHERE: {
                if (idx >= d->n_spheres) { g_err = "sphere index out of range"; return nullptr; }
                const vk_sphere &s = d->spheres[idx];
                if (s.material >= d->n_materials) { g_err = "material index out of range"; return nullptr; }
                auto o = std::make_shared<Sphere>();
                o->center = Vec3(s.center[0], s.center[1], s.center[2]);
                o->radius = s.radius; o->material = materials[s.material].get(); o->mi = s.material;
                out = o;
                break;
            }
            case VK_KIND_MOVING_SPHERE: {
                if (idx >= d->n_moving_spheres) { g_err = "moving sphere index out of range"; return nullptr; }
                const vk_moving_sphere &s = d->moving_spheres[idx];
                if (s.material >= d->n_materials) { g_err = "material index out of range"; return nullptr; }
                auto o = std::make_shared<MovingSphere>();
                o->center0 = Vec3(s.center0[0], s.center0[1], s.center0[2]);
                o->center1 = Vec3(s.center1[0], s.center1[1], s.center1[2]);
                o->time0 = s.time0; o->time1 = s.time1; o->radius = s.radius;
                o->material = materials[s.material].get(); o->mi = s.material;
                out = o;
                break;
            }
            case VK_KIND_RECT: {
                if (idx >= d->n_rects) { g_err = "rect index out of range"; return nullptr; }
                const vk_rect &s = d->rects[idx];
                if (s.material >= d->n_materials) { g_err = "material index out of range"; return nullptr; }
                if (s.axis0 > 2 || s.axis1 > 2 || s.axis2 > 2) { g_err = "rect axis out of range"; return nullptr; }
                auto o = std::make_shared<Rect>();
                o->c0 = s.c0; o->c1 = s.c1; o->d0 = s.d0; o->d1 = s.d1; o->k = s.k;
                o->axis0 = s.axis0; o->axis1 = s.axis1; o->axis2 = s.axis2;
                o->mat = materials[s.material].get(); o->mi = s.material;
                out = o;
                break;
            }
            case VK_KIND_LIST: {
                if (idx >= d->n_lists) { g_err = "list index out of range"; return nullptr; }
                const vk_list &l = d->lists[idx];
                if ((uint64_t)l.first + l.count > d->n_list_items) { g_err = "list items out of range"; return nullptr; }
                auto o = std::make_shared<HittableList>();
                for (uint32_t i = 0; i < l.count; i++) {
                    HittableP c = build_ref(d->list_items[l.first + i], depth + 1);
                    if (!c) return nullptr;
                    o->items.push_back(c);
                }
                out = o;
                break;
            }
            case VK_KIND_MEDIUM: {
                if (idx >= d->n_media) { g_err = "medium index out of range"; return nullptr; }
                const vk_medium &s = d->media[idx];
                if (s.material >= d->n_materials) { g_err = "material index out of range"; return nullptr; }
                auto o = std::make_shared<ConstantMedium>();
                o->boundary = build_ref(s.boundary, depth + 1);
                if (!o->boundary) return nullptr;
                o->neg_inv_density = s.neg_inv_density;
                o->phase_function = materials[s.material].get(); o->mi = s.material;
                out = o;
                break;
            }
            case VK_KIND_TRANSLATE: {
                if (idx >= d->n_translates) { g_err = "translate index out of range"; return nullptr; }
                const vk_translate &s = d->translates[idx];
                auto o = std::make_shared<Translate>();
                o->ptr = build_ref(s.child, depth + 1);
                if (!o->ptr) return nullptr;
                o->offset = Vec3(s.offset[0], s.offset[1], s.offset[2]);
                out = o;
                break;
            }
            case VK_KIND_ROTATE: {
                if (idx >= d->n_rotates) { g_err = "rotate index out of range"; return nullptr; }
                const vk_rotate &s = d->rotates[idx];
                if (s.axis > 2) { g_err = "rotate axis out of range"; return nullptr; }
                auto o = std::make_shared<Rotate>();
                o->ptr = build_ref(s.child, depth + 1);
                if (!o->ptr) return nullptr;
                o->axis = (int)s.axis; o->sin_theta = s.sin_theta; o->cos_theta = s.cos_theta;
                out = o;
                break;
            }
            default:
                g_err = "unknown hittable kind";
                return nullptr;
        }
        if (m.size() <= idx) m.resize(idx + 1);
        m[idx] = out;
        return out;
    }

    bool build(const vk_scene_desc *desc) {
        d = desc;
        if (!desc) return fail("null scene desc");
        if (desc->abi_version != VK_ABI_VERSION) return fail("abi version mismatch");
        // textures (children must have smaller... any index; two passes)
        textures.resize(desc->n_textures);
        for (uint32_t i = 0; i < desc->n_textures; i++) {
            const vk_texture &t = desc->textures[i];
            switch (t.kind) {
                case VK_TEX_SOLID: { auto p = new SolidColor; p->color_value = Vec3(t.color[0], t.color[1], t.color[2]);
                    textures[i].reset(p); break; }
                case VK_TEX_CHECKER: { textures[i].reset(new Checker); break; }
                case VK_TEX_IMAGE: {
                    if (t.a >= desc->n_images) return fail("image index out of range");
                    auto p = new ImageTexture; p->buf = desc->images[t.a].rgb; p->width = desc->images[t.a].width;
                    p->height = desc->images[t.a].height;
                    if (!p->buf || !p->width || !p->height) { delete p; return fail("empty image"); }
                    textures[i].reset(p); break;
                }
                case VK_TEX_NOISE: {
                    if (t.a >= desc->n_perlins) return fail("perlin index out of range");
                    auto p = new NoiseTexture; p->scale = t.scale;
                    const vk_perlin &pl = desc->perlins[t.a];
                    for (int k = 0; k < 256; k++) {
                        p->noise.random_data[k] = Vec3(pl.ranvec[k][0], pl.ranvec[k][1], pl.ranvec[k][2]);
                        p->noise.perm_x[k] = pl.perm_x[k] & 255; p->noise.perm_y[k] = pl.perm_y[k] & 255;
                        p->noise.perm_z[k] = pl.perm_z[k] & 255;
                    }
                    textures[i].reset(p); break;
                }
                default: return fail("unknown texture kind");
            }
        }
        for (uint32_t i = 0; i < desc->n_textures; i++) {
            const vk_texture &t = desc->textures[i];
            if (t.kind == VK_TEX_CHECKER) {
                if (t.a >= desc->n_textures || t.b >= desc->n_textures) return fail("checker child out of range");
                auto c = static_cast<Checker *>(textures[i].get());
                c->odd = textures[t.a].get(); c->even = textures[t.b].get();
            }
        }
        materials.resize(desc->n_materials);
        for (uint32_t i = 0; i < desc->n_materials; i++) {
            const vk_material &m = desc->materials[i];
            bool needs_tex = m.kind == VK_MAT_LAMBERTIAN || m.kind == VK_MAT_METAL || m.kind == VK_MAT_DIFFUSE_LIGHT ||
                m.kind == VK_MAT_ISOTROPIC;
            if (needs_tex && m.texture >= desc->n_textures) return fail("texture index out of range");
            switch (m.kind) {
                case VK_MAT_LAMBERTIAN: { auto p = new Lambertian; p->albedo = textures[m.texture].get(); materials[i].reset(p); break; }
                case VK_MAT_METAL: { auto p = new Metal; p->albedo = textures[m.texture].get(); p->fuzz = m.param; materials[i].reset(p);
                    break; }
                case VK_MAT_DIELECTRIC: { auto p = new Dielectric; p->ref_idx = m.param; materials[i].reset(p); break; }
                case VK_MAT_DIFFUSE_LIGHT: { auto p = new DiffuseLight; p->emit = textures[m.texture].get(); materials[i].reset(p); break; }
                case VK_MAT_ISOTROPIC: { auto p = new Isotropic; p->albedo = textures[m.texture].get(); materials[i].reset(p); break; }
                case VK_MAT_SPEC_DIFFUSE: { auto p = new SpecDiffuse; p->pct = m.param; materials[i].reset(p); break; }
                default: return fail("unknown material kind");
            }
        }
        for (uint32_t i = 0; i < desc->n_materials; i++) {
            const vk_material &m = desc->materials[i];
            if (m.kind == VK_MAT_SPEC_DIFFUSE) {
                if (m.a >= desc->n_materials || m.b >= desc->n_materials) return fail("spec_diffuse child out of range");
                auto p = static_cast<SpecDiffuse *>(materials[i].get());
                p->specular = materials[m.a].get(); p->diffuse = materials[m.b].get();
            }
        }
        world = build_ref(desc->world, 0);
        if (!world) return false;
        lights = std::make_shared<HittableList>();
        for (uint32_t i = 0; i < desc->n_lights; i++) {
            HittableP l = build_ref(desc->lights[i], 0);
            if (!l) return false;
            lights->items.push_back(l);
        }
        return true;
    }
};

// ---------------------------------------------------------------------------- main.rs:56-121
struct Camera {
    Vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    float lens_radius, time0, time1;
    explicit Camera(const vk_camera &c) {
        origin = Vec3(c.origin[0], c.origin[1], c.origin[2]);
        lower_left_corner = Vec3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
        horizontal = Vec3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
        vertical = Vec3(c.vertical[0], c.vertical[1], c.vertical[2]);
        u = Vec3(c.u[0], c.u[1], c.u[2]); v = Vec3(c.v[0], c.v[1], c.v[2]); w = Vec3(c.w[0], c.w[1], c.w[2]);
        lens_radius = c.lens_radius; time0 = c.time0; time1 = c.time1;
    }
    Ray get_ray(float s, float t) const {  // main.rs:111-120
        Vec3 rd = random_in_unit_disk() * lens_radius;
        Vec3 offset = u * rd.x + v * rd.y;
        Vec3 o = origin + offset;
        Vec3 dir = lower_left_corner + horizontal * s + vertical * t - origin - offset;
        float time = gen_range(time0, time1);
        return Ray(o, dir, time);
    }
};

struct RenderCtx {
    const Scene *scene;
    uint32_t max_depth;
    uint32_t integrator, background;
    Vec3 background_color;
};

inline Vec3 background_of(const RenderCtx &c, const Ray &r) {
    if (c.background == VK_BACKGROUND_SKY) {  // InOneWeekend sky (not at HEAD; see SURVEY §8a "integrator variants")
        Vec3 unit_direction = r.direction.unit_vector();
        float t = 0.5f * (unit_direction.y + 1.0f);
        return Vec3::new_const(1.0f) * (1.0f - t) + Vec3(0.5f, 0.7f, 1.0f) * t;
    }
    return c.background_color;  // main.rs:124
}

// (counters only) does the ray have a NaN / infinite direction or origin?  Dielectric::scatter's refract() produces such
// directions just past the critical angle (sqrt of a rounding-negative number, util.rs:18-23)
static inline bool nonfinite_ray(const Ray &r) {
    float s = std::fabs(r.origin.x) + std::fabs(r.origin.y) + std::fabs(r.origin.z);
    return !(r.direction.length2() < INFINITY) || !(s < INFINITY);
}

// main.rs:123-153
Vec3 ray_color(const RenderCtx &ctx, Ray r, uint32_t depth) {
    if (depth > ctx.max_depth) return Vec3::new_const(0.0f);  // main.rs:126-128
    g_cnt.segments++;
    g_nonfinite_segment = nonfinite_ray(r) ? 1u : 0u;   // counters only: such a ray passes every AxisBB::hit and fails every Sphere::hit
    HitRec c;
    if (ctx.scene->world->hit(r, 0.001f, INFINITY, c)) {  // main.rs:130
        g_cnt.n_closest++;
        Vec3 emitted = c.material->emitted(c, c.u, c.v, c.p);  // main.rs:131
        ScatterRec srec;
        if (c.material->scatter_with_pdf(r, c, srec)) {  // main.rs:132
            if (srec.has_specular) {  // main.rs:134-137
                return srec.attenuation * ray_color(ctx, srec.specular_ray, depth + 1);
            }
            HittablePDF p_important(ctx.scene->lights.get(), c.p);  // main.rs:139
            MixturePDF p(&p_important, 0.5f, srec.pdf.get(), 0.5f);  // main.rs:140
            Ray scattered(c.p, p.generate(), r.time);                // main.rs:142
            float pdf = p.value(scattered.direction);                // main.rs:143
            return emitted + srec.attenuation * c.material->scattering_pdf(r, c, scattered) *
                                 ray_color(ctx, scattered, depth + 1) / pdf;  // main.rs:144-146
        } else {
            return emitted;  // main.rs:147-149
        }
    } else {
        return background_of(ctx, r);  // main.rs:150-152
    }
}

// InOneWeekend / TheNextWeek integrator on Material::scatter (material.rs:21-28 and overrides)
Vec3 ray_color_scatter(const RenderCtx &ctx, Ray r, uint32_t depth) {
    if (depth > ctx.max_depth) return Vec3::new_const(0.0f);
    g_cnt.segments++;
    g_nonfinite_segment = nonfinite_ray(r) ? 1u : 0u;   // counters only: such a ray passes every AxisBB::hit and fails every Sphere::hit
    HitRec c;
    if (ctx.scene->world->hit(r, 0.001f, INFINITY, c)) {
        g_cnt.n_closest++;
        Vec3 emitted = c.material->emitted(c, c.u, c.v, c.p);
        Vec3 attenuation;
        Ray scattered;
        if (c.material->scatter(r, c, attenuation, scattered)) {
            return emitted + attenuation * ray_color_scatter(ctx, scattered, depth + 1);
        } else {
            return emitted;
        }
    } else {
        return background_of(ctx, r);
    }
}

inline Vec3 trace_sample(const RenderCtx &ctx, const Camera &cam, const vk_render_params &p, uint32_t x, uint32_t y) {
    // main.rs:187-190
    float u = ((float)x + gen_f32()) / (float)(p.width - 1);
    float v = ((float)y + gen_f32()) / (float)(p.height - 1);
    Ray ray = cam.get_ray(u, v);
    if (ctx.integrator == VK_INTEGRATOR_PDF) return ray_color(ctx, ray, 1);
    return ray_color_scatter(ctx, ray, 1);
}

bool check_params(const vk_camera *cam, const vk_render_params *p) {
    if (!cam || !p) { g_err = "null argument"; return false; }
    if (p->width < 2 || p->height < 2) { g_err = "width/height must be >= 2"; return false; }
    if (p->samples_per_pixel == 0) { g_err = "samples_per_pixel must be > 0"; return false; }
    if (!(cam->time0 < cam->time1)) { g_err = "camera time0 >= time1 (gen_range would panic, main.rs:118)"; return false; }
    if (p->integrator > 1 || p->background > 1) { g_err = "bad integrator/background"; return false; }
    return true;
}

inline void add_counters(oracle_counters &a, const oracle_counters &b) {
    uint64_t *pa = reinterpret_cast<uint64_t *>(&a);
    const uint64_t *pb = reinterpret_cast<const uint64_t *>(&b);
    for (size_t i = 0; i < sizeof(oracle_counters) / sizeof(uint64_t); i++) pa[i] += pb[i];
}

}  // namespace

extern "C" {

const char *oracle_last_error(void) { return g_err.c_str(); }

static int oracle_render_impl(const vk_scene_desc *desc, const vk_camera *cam_in, const vk_render_params *params,
                              float *rgb_out, int n_threads, oracle_counters *counters_out, float *per_sample_out);

int oracle_render(const vk_scene_desc *desc, const vk_camera *cam_in, const vk_render_params *params,
                  float *rgb_out, int n_threads, oracle_counters *counters_out) {
    return oracle_render_impl(desc, cam_in, params, rgb_out, n_threads, counters_out, nullptr);
}

int oracle_render_samples(const vk_scene_desc *desc, const vk_camera *cam_in, const vk_render_params *params,
                          float *rgb_out, float *per_sample_out, int n_threads) {
    return oracle_render_impl(desc, cam_in, params, rgb_out, n_threads, nullptr, per_sample_out);
}

static int oracle_render_impl(const vk_scene_desc *desc, const vk_camera *cam_in, const vk_render_params *params,
                              float *rgb_out, int n_threads, oracle_counters *counters_out, float *per_sample_out) {
    g_err.clear();
    if (!rgb_out) { g_err = "null output"; return VK_ERR_BAD_ARG; }
    if (!check_params(cam_in, params)) return VK_ERR_BAD_ARG;
    Scene scene;
    if (!scene.build(desc)) return VK_ERR_BAD_ARG;
    const vk_render_params p = *params;
    Camera cam(*cam_in);
    RenderCtx ctx;
    ctx.scene = &scene; ctx.max_depth = p.max_depth; ctx.integrator = p.integrator; ctx.background = p.background;
    ctx.background_color = Vec3(p.background_color[0], p.background_color[1], p.background_color[2]);
    if (n_threads < 1) n_threads = 1;
    const uint32_t tiles_x = (p.width + 7) / 8;
    const uint32_t tile_world = p.tile_world ? p.tile_world : 1;
    const uint32_t tiles_y = (p.height + 7) / 8;
    // rayon's par_iter_mut over pixels (main.rs:181) = dynamic stealing; the stand-in's unit is one 8x8 tile (a row would leave
    // 1080 units for 256 threads)
    std::atomic<uint32_t> next_tile(0);
    std::atomic<int> panicked(0);
    std::vector<oracle_counters> per_thread(n_threads);
    auto worker = [&](int tid) {
        g_cnt = oracle_counters();
        g_panic = false;
        for (;;) {
            uint32_t tile = next_tile.fetch_add(1);
            if (tile >= tiles_x * tiles_y) break;
            if (tile % tile_world != p.tile_rank % tile_world) continue;
            const uint32_t x0 = (tile % tiles_x) * 8, y0 = (tile / tiles_x) * 8;
            for (uint32_t y = y0; y < y0 + 8 && y < p.height; y++)
            for (uint32_t x = x0; x < x0 + 8 && x < p.width; x++) {
                uint32_t i = y * p.width + x;  // main.rs:182-183
                Vec3 c = Vec3::new_const(0.0f);
                for (uint32_t s = 0; s < p.samples_per_pixel; s++) {  // main.rs:186
                    vk::Rng rng = vk::rng_for_sample(p.seed, i, s);
                    g_rng = &rng;
                    Vec3 color = trace_sample(ctx, cam, p, x, y);
                    g_cnt.samples++;
                    if (per_sample_out) {
                        float *o = per_sample_out + ((size_t)i * p.samples_per_pixel + s) * 4;
                        o[0] = color.x; o[1] = color.y; o[2] = color.z;
                        uint32_t dr = rng.ctr;
                        memcpy(&o[3], &dr, 4);
                    }
                    if (std::isfinite(color.x) && std::isfinite(color.y) && std::isfinite(color.z)) {  // main.rs:192-194
                        c = c + color;
                    } else {
                        g_cnt.n_dropped++;
                    }
                }
                c = c / (float)p.samples_per_pixel;  // main.rs:196
                rgb_out[3 * (size_t)i + 0] = c.x; rgb_out[3 * (size_t)i + 1] = c.y; rgb_out[3 * (size_t)i + 2] = c.z;
            }
        }
        g_rng = nullptr;
        if (g_panic) panicked = 1;
        per_thread[tid] = g_cnt;
    };
    std::vector<std::thread> ths;
    for (int t = 1; t < n_threads; t++) ths.emplace_back(worker, t);
    worker(0);
    for (auto &t : ths) t.join();
    if (counters_out) {
        *counters_out = oracle_counters();
        for (auto &c : per_thread) add_counters(*counters_out, c);
    }
    if (panicked) { g_err = "reference would panic (unwrap on None)"; return VK_ERR_UNSUPPORTED; }
    return VK_OK;
}

int oracle_sample(const vk_scene_desc *desc, const vk_camera *cam_in, const vk_render_params *params,
                  uint32_t pixel, uint32_t sample, float rgb_out[3], uint32_t *draws_out) {
    g_err.clear();
    if (!check_params(cam_in, params)) return VK_ERR_BAD_ARG;
    Scene scene;
    if (!scene.build(desc)) return VK_ERR_BAD_ARG;
    Camera cam(*cam_in);
    RenderCtx ctx;
    ctx.scene = &scene; ctx.max_depth = params->max_depth; ctx.integrator = params->integrator; ctx.background = params->background;
    ctx.background_color = Vec3(params->background_color[0], params->background_color[1], params->background_color[2]);
    g_cnt = oracle_counters();
    vk::Rng rng = vk::rng_for_sample(params->seed, pixel, sample);
    g_rng = &rng;
    Vec3 c = trace_sample(ctx, cam, *params, pixel % params->width, pixel / params->width);
    g_rng = nullptr;
    rgb_out[0] = c.x; rgb_out[1] = c.y; rgb_out[2] = c.z;
    if (draws_out) *draws_out = rng.ctr;
    return VK_OK;
}

int oracle_hit(const vk_scene_desc *desc, const float origin[3], const float dir[3], float time,
               float tmin, float tmax, uint64_t seed, float rec_out[11]) {
    g_err.clear();
    Scene scene;
    if (!scene.build(desc)) return -1;
    vk::Rng rng = vk::rng_for_sample(seed, 0, 0);
    g_rng = &rng;
    g_cnt = oracle_counters();
    Ray r(Vec3(origin[0], origin[1], origin[2]), Vec3(dir[0], dir[1], dir[2]), time);
    HitRec rec;
    bool h = scene.world->hit(r, tmin, tmax, rec);
    g_rng = nullptr;
    if (h) {
        rec_out[0] = rec.p.x; rec_out[1] = rec.p.y; rec_out[2] = rec.p.z;
        rec_out[3] = rec.normal.x; rec_out[4] = rec.normal.y; rec_out[5] = rec.normal.z;
        rec_out[6] = rec.t; rec_out[7] = rec.u; rec_out[8] = rec.v; rec_out[9] = rec.front ? 1.0f : 0.0f;
        rec_out[10] = (float)rec.material_index;
    }
    return h ? 1 : 0;
}

void oracle_math(int op, const float *a, const float *b, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: out[i] = vk::sinf_(a[i]); break;
            case 1: out[i] = vk::cosf_(a[i]); break;
            case 2: out[i] = vk::logf_(a[i]); break;
            case 3: out[i] = vk::asinf_(a[i]); break;
            case 4: out[i] = vk::atan2f_(a[i], b[i]); break;
            case 5: out[i] = vk::pow5f_(a[i]); break;
            default: out[i] = 0.0f;
        }
    }
}

void oracle_draws(uint64_t seed, uint32_t pixel, uint32_t sample, int kind, float lo, float hi,
                  uint32_t n_index, float *out, size_t n) {
    vk::Rng rng = vk::rng_for_sample(seed, pixel, sample);
    for (size_t i = 0; i < n; i++) {
        if (kind == 0) out[i] = vk::gen_f32(rng);
        else if (kind == 1) out[i] = vk::gen_range(rng, lo, hi);
        else out[i] = (float)vk::gen_index(rng, n_index);
    }
}

}  // extern "C"
