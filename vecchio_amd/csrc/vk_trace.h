// vk_trace.h — per-lane path tracing logic of the megakernel (one ray per lane).
//
// This is NOT a translation of the reference's recursion: ray_color (main.rs:123-153) is
// an iterative throughput loop, BVHNode::hit (accel.rs:58-83) is a stack-free threaded
// pre-order walk over 32-byte items, hit records are DEFERRED (only t, primitive ref and
// instance are tracked during traversal; p/normal/uv/front are built once for the closest
// hit), AxisBB::hit (accel.rs:16-35) is decided with reciprocal multiplies and falls back to
// the reference's divisions only when the decision is within rounding distance.  What is
// kept bit-for-bit is every value that feeds control flow: t of every accepted hit, the hit
// record of the closest hit, every draw (order and count) and every scatter direction.
// Radiance arithmetic is re-associated (it never feeds control flow).
//
// Compiled for gfx950 by hipcc and for the host by g++ (tests/emu only), both with
// -ffp-contract=off.  F = compile-time feature mask (VKF_*) selecting the kernel variant.
#ifndef VK_TRACE_H
#define VK_TRACE_H

#include "../../include/vecchio_amd.h"
#include "vk_device_scene.h"
#include "vk_math.h"

namespace vkd {

using vk::Rng;

// ------------------------------------------------------------------ vec3.rs (value semantics identical)
struct V3 { float x, y, z; };
VK_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
VK_HD V3 v3s(float s) { return v3(s, s, s); }
VK_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
VK_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
VK_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
VK_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
VK_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
VK_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
VK_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VK_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VK_HD float length2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
VK_HD V3 unit(V3 a) { float n = sqrtf(length2(a)); return v3(a.x / n, a.y / n, a.z / n); }
VK_HD float comp(V3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
VK_HD V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }

constexpr float PI_F = 3.14159265358979323846f;
constexpr float T_MIN = 0.001f;  // main.rs:130

struct RenderConsts {   // per-launch constants (camera + params)
    vk_camera cam;
    uint32_t width, height, spp, max_depth;
    uint64_t seed;
    uint32_t integrator, background;
    float bg[3];
};

// ------------------------------------------------------------------ memory policies
// Hot records (items, spheres) come either from HBM/L2 (GlobalMem) or from the workgroup's
// LDS copy of the scene (LdsMem, set up by the kernel).
struct GlobalMem {
    // The traversal cursor (Lane::i, Lane::end, the skip links) counts in units of 1 << ISHIFT: plain item indices here; the
    // LDS-resident scene (vk_kernels.h LdsMem) counts in BYTES of its 16-byte-stride arrays, so that a box step needs no shift
    // to form its two ds_read addresses (the skip links are scaled once, when a workgroup stages the items).
    static constexpr uint32_t ISHIFT = 0;
    // the fused box test (set_space) is for scenes traversed from LDS: from global memory a box step waits for its gather, not
    // for the VALU, and the three extra registers are spills at the 64 VGPRs of the 8-waves-per-SIMD build
    static constexpr bool FUSED_BOX = false;
    const DItem *items; const DSphere *spheres; const uint32_t *sphere_mat; const DBox *boxes;
    VK_HD DBox box(uint32_t i) const { return boxes[i]; }
    VK_HD DItem item(uint32_t i) const { return items[i]; }
    VK_HD DSphere sphere(uint32_t i) const { return spheres[i]; }
    VK_HD uint32_t smat(uint32_t i) const { return sphere_mat[i]; }
    // the grid form's cell table and reference lists (DGrid)
    VK_HD uint32_t grid_cell(const DScene &S, uint32_t c) const { return S.grid_cells[c]; }
    VK_HD uint32_t grid_ref(const DScene &S, uint32_t k) const { return S.grid_refs[k]; }
};

// ------------------------------------------------------------------ per-lane state
struct Lane {
    // current-space ray (object space while inside an instance)
    V3 o, d; float time;
    // xnan: the additive part of the box test's margin (0 unfused), or NaN when its fast path must not be trusted
    V3 inv; float a; float xnan;
    V3 oi;                     // o * (1/d): fused box test only (set_space)
    V3 wo, wd;                 // world-space ray of this segment
    // traversal cursor
    uint32_t i, end, pend, pend2; int32_t cur_inst;
    // ... of the grid form (DGrid; grid_step below): i, end = the current cell's range in refs[]; cell = major index | minor index << 10
    // | last minor index of this column << 20, or GRID_LAST / GRID_DONE; the segment's parameter range inside the layer's dilated box
    // and the dilation itself
    uint32_t cell; float ta, tb, dl;
    // closest hit so far (deferred record)
    float T; uint32_t best_prim; int32_t best_inst; float best_aux;
    // path
    V3 thr, acc; uint32_t depth;
    Rng rng;
    uint32_t pixel, sample;
};

// the next f32 above a positive finite t (+inf stays +inf)
VK_HD float nextafter_up(float t) { return (t > 0.0f && t < INFINITY) ? vk::bits_f32(vk::f32_bits(t) + 1u) : t; }

// The sphere-only kernel variants run the FUSED box test (box_step_core): one fma per bound, t~ = fma(b, 1/d, -o/d), against
// the other variants' (b - o) * (1/d).  Both are decided against the reference's fl(fl(b - o) / d) with a margin and fall
// back to the reference's own division sequence inside it; the fused form needs three more registers per lane (o/d), which
// only the sphere-only variants have to spare, and its margin carries a term in |o/d| (cancellation when b*(1/d) ~ o/d).
template <uint32_t F, class Mem> constexpr bool fused_box() { return (F & ~(uint32_t)VKF_INTEG_PDF) == 0u && Mem::FUSED_BOX; }
template <uint32_t F> constexpr bool spheres_only() { return (F & ~(uint32_t)VKF_INTEG_PDF) == 0u; }

// gate_scale: 1, or DScene::gate_scale = 1 / (1 + t_pad) under exact re-treeing (see exact re-treeing below): the box test then runs
// on distances scaled by it, i.e. it compares the boxes with the closest hit so far times (1 + t_pad)
// TIGHT (sphere-only variants): the narrower range of ray components in which the sphere test divides through a shared reciprocal
template <bool FUSED = false, bool TIGHT = FUSED>
VK_HD void set_space(Lane &L, V3 o, V3 d, float gate_scale = 1.0f) {
    L.o = o; L.d = d;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VK_EXACT_INV)
    // v_rcp_f32 (1 ulp) instead of three correctly rounded divisions (~36 instructions per new ray and per instance entered or
    // left): the reciprocals only enter the box test's FAST path, whose margins (box_step_core) cover a reciprocal that is off by one
    // ulp — q~ then differs from the reference's fl(fl(b - o) / d) by < 4 * 2^-24 relative instead of 3 * 2^-24 (fused form: 5u|t| +
    // u|o/d| instead of 4u|t| + u|o/d|), against margins of 33 * 2^-24 * hi (+ 8u * max|o/d|) — and inside the margin the decision is
    // the reference's own division sequence either way.  tests/test_box_decision.py runs the 4 M cases with reciprocals perturbed by
    // +-1 ulp as well.  C2 +1.1 %, C4 +1.3 %.
    L.inv = v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
#else
    L.inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
#endif
    L.inv = L.inv * gate_scale;          // (times 1.0f is exact)
    L.a = length2(d);
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (FUSED) {
        // With u = 2^-24, r = fl(1/d), c = fl(o*r): fma(b, r, -c) = (1/d)(1+e2)(1+e4)(b - o - o*e3), the reference's value is
        // ((b - o)/d)(1+e0)(1+e1), all |e| <= u, so the two differ by at most 4u|t| + u|o/d| (+ second order).  As in the
        // unfused case (box_step_core) the decision hi > lo is then the reference's whenever |hi - lo| >= (8u*hi + 2u*max|o/d|)/(1 - 4u):
        // the test there uses 2e-6*hi + 2^-21*max|c|, four times that.  Rays whose 1/d or o/d leave the range where these
        // bounds hold without overflow go through the exact test every time (xnan = NaN).
        L.oi = v3(o.x * L.inv.x, o.y * L.inv.y, o.z * L.inv.z);
        float c = fmaxf(fmaxf(fabsf(L.oi.x), fabsf(L.oi.y)), fabsf(L.oi.z));
        // (1e-6 .. 1e6 and an origin below 1e9: also the range in which the sphere test may divide by |d|^2 with a shared reciprocal,
        // div_by_a below; a scattered direction has a component below 1e-6 three times in a million)
        bool ok = ax > 1e-6f && ax < 1e6f && ay > 1e-6f && ay < 1e6f && az > 1e-6f && az < 1e6f && c < 1e12f &&
                  fabsf(o.x) + fabsf(o.y) + fabsf(o.z) < 1e9f;
        L.xnan = ok ? c * 4.76837158203125e-7f : vk::bits_f32(0x7FC00000u);      // 2^-21
    } else {
        // reciprocal-multiply slab test is only trusted when 1/d is a full-precision normal number
        bool ok = ax > 1e-30f && ax < 1e30f && ay > 1e-30f && ay < 1e30f && az > 1e-30f && az < 1e30f;
        if (TIGHT) ok = ax > 1e-6f && ax < 1e6f && ay > 1e-6f && ay < 1e6f && az > 1e-6f && az < 1e6f &&
                        fabsf(o.x) + fabsf(o.y) + fabsf(o.z) < 1e9f;      // (see the fused form)
        L.xnan = ok ? 0.0f : vk::bits_f32(0x7FC00000u);
    }
}

// ------------------------------------------------------------------ instance transforms (hittable.rs:507-524,579-624,676-713,765-802)
VK_HD void apply_op(const DOp &op, V3 &o, V3 &d) {
    float s = op.a, c = op.b;
    switch (op.kind) {
        case OP_TRANSLATE: o = o - v3(op.a, op.b, op.c); break;                      // hittable.rs:508
        case OP_ROTATE_Y: {                                                          // hittable.rs:591-595
            float ox = c * o.x - s * o.z, oz = s * o.x + c * o.z; o.x = ox; o.z = oz;
            float dx = c * d.x - s * d.z, dz = s * d.x + c * d.z; d.x = dx; d.z = dz; break; }
        case OP_ROTATE_X: {                                                          // hittable.rs:680-684
            float oy = c * o.y + s * o.z, oz = -s * o.y + c * o.z; o.y = oy; o.z = oz;
            float dy = c * d.y + s * d.z, dz = -s * d.y + c * d.z; d.y = dy; d.z = dz; break; }
        default: {                                                                   // hittable.rs:769-773
            float ox = c * o.x + s * o.y, oy = -s * o.x + c * o.y; o.x = ox; o.y = oy;
            float dx = c * d.x + s * d.y, dy = -s * d.x + c * d.y; d.x = dx; d.y = dy; break; }
    }
}
VK_HD void unapply_op(const DOp &op, V3 &p, V3 &n) {
    float s = op.a, c = op.b;
    switch (op.kind) {
        case OP_TRANSLATE: p = p + v3(op.a, op.b, op.c); break;                      // hittable.rs:511
        case OP_ROTATE_Y: {                                                          // hittable.rs:603-607
            float px = c * p.x + s * p.z, pz = -s * p.x + c * p.z; p.x = px; p.z = pz;
            float nx = c * n.x + s * n.z, nz = -s * n.x + c * n.z; n.x = nx; n.z = nz; break; }
        case OP_ROTATE_X: {                                                          // hittable.rs:692-696
            float py = c * p.y - s * p.z, pz = s * p.y + c * p.z; p.y = py; p.z = pz;
            float ny = c * n.y - s * n.z, nz = s * n.y + c * n.z; n.y = ny; n.z = nz; break; }
        default: {                                                                   // hittable.rs:781-785
            float px = c * p.x - s * p.y, py = s * p.x + c * p.y; p.x = px; p.y = py;
            float nx = c * n.x - s * n.y, ny = s * n.x + c * n.y; n.x = nx; n.y = ny; break; }
    }
}
// ray of instance `inst` (-1 = world) re-derived from the world ray: same ops, same order,
// hence the same bits as when the instance was entered
VK_HD void ray_in_instance(const DScene &S, int32_t inst, V3 wo, V3 wd, V3 &o, V3 &d) {
    o = wo; d = wd;
    if (inst < 0) return;
    const DInstance &I = S.instances[inst];
    for (uint32_t l = 0; l <= I.depth; l++) {
        const DInstance &J = S.instances[I.chain[l]];
        for (uint32_t k = 0; k < J.n_ops; k++) apply_op(J.ops[k], o, d);
    }
}

// ------------------------------------------------------------------ primitive tests (t only)
// Sphere::hit, hittable.rs:65-95: half-b quadratic, strict bounds, near root first
VK_HD bool sphere_t(float cx, float cy, float cz, float r, V3 o, V3 d, float a, float tmin, float tmax, float &t) {
    V3 oc = o - v3(cx, cy, cz);
    float half_b = dot(oc, d);
    float c = length2(oc) - r * r;
    float disc = half_b * half_b - a * c;
    if (disc > 0.0f) {
        float root = sqrtf(disc);
        float t1 = (-half_b - root) / a;
        if (tmin < t1 && t1 < tmax) { t = t1; return true; }
        float t2 = (-half_b + root) / a;
        if (tmin < t2 && t2 < tmax) { t = t2; return true; }
    }
    return false;
}
// ---- division by a = |d|^2 with a shared reciprocal (sphere-only variants, vk_trace.h prim_step).  A leaf's two sphere tests divide by
// the same a up to four times; the compiler expands every f32 division into v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup
// (11 instructions).  Without the scaling and the fix-up — which only act on operands near the ends of the exponent range — that
// expansion is: y = rcp(a) refined once; q = n y; r = fma(-a, q, n); q = fma(r, y, q); r = fma(-a, q, n); q = fma(r, y, q): the
// correctly rounded quotient.  So the refined reciprocal is computed once per step and a quotient costs five instructions, for rays
// inside the range set_space trusts (|d| components in 1e-6 .. 1e6, origin below 1e9: xnan is not NaN) in scenes whose spheres lie
// below 2^30 (DScene::fast_div): then a is in 3e-12 .. 3e12, |n| < 2^54, and no intermediate leaves the normal range except towards
// zero, where the quotient is below tmin either way.  Any other lane divides.  The host build (tests/emu) always divides: the per-sample
// GPU parity tests and test_gpu_parity's 2^26-pair probe compare the two bit for bit.
VK_HD float refined_rcp(float a) {
#if defined(__HIP_DEVICE_COMPILE__)
    float y = __builtin_amdgcn_rcpf(a);
    const float e = __builtin_fmaf(-a, y, 1.0f);
    return __builtin_fmaf(e, y, y);
#else
    return 1.0f / a;
#endif
}
VK_HD float div_by_a(float n, float a, float y, bool fast) {
#if defined(__HIP_DEVICE_COMPILE__)
    float q = n * y;
    float r = __builtin_fmaf(-a, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-a, q, n);
    q = __builtin_fmaf(r, y, q);
    if (__builtin_amdgcn_ballot_w64(!fast) != 0ull) {      // (wave-uniform: no lane in a million takes it)
        asm volatile("");
        if (!fast) q = n / a;
    }
    return q;
#else
    (void)y; (void)fast;
    return n / a;
#endif
}
// sphere_t_tie with the quotients from div_by_a
VK_HD bool sphere_t_tie_y(float cx, float cy, float cz, float r, V3 o, V3 d, float a, float y, bool fast, float tmin, float tmax,
    float &t, bool &tie) {
    V3 oc = o - v3(cx, cy, cz);
    float half_b = dot(oc, d);
    float c = length2(oc) - r * r;
    float disc = half_b * half_b - a * c;
    tie = false;
    if (disc > 0.0f) {
        float root = sqrtf(disc);
        float t1 = div_by_a(-half_b - root, a, y, fast);
        if (tmin < t1 && t1 <= tmax) { t = t1; tie = t1 == tmax; return true; }
        float t2 = div_by_a(-half_b + root, a, y, fast);
        if (tmin < t2 && t2 <= tmax) { t = t2; tie = t2 == tmax; return true; }
    }
    return false;
}
// the same with t == tmax reported too (`tie`): the root Sphere::hit would have taken had its bound been inclusive.  The near
// root is taken when it is in (tmin, tmax]; a near root at tmax leaves no far root below tmax, so the choice is the reference's.
VK_HD bool sphere_t_tie(float cx, float cy, float cz, float r, V3 o, V3 d, float a, float tmin, float tmax, float &t, bool &tie) {
    V3 oc = o - v3(cx, cy, cz);
    float half_b = dot(oc, d);
    float c = length2(oc) - r * r;
    float disc = half_b * half_b - a * c;
    tie = false;
    if (disc > 0.0f) {
        float root = sqrtf(disc);
        float t1 = (-half_b - root) / a;
        if (tmin < t1 && t1 <= tmax) { t = t1; tie = t1 == tmax; return true; }
        float t2 = (-half_b + root) / a;
        if (tmin < t2 && t2 <= tmax) { t = t2; tie = t2 == tmax; return true; }
    }
    return false;
}
VK_HD V3 moving_center(const DMoving &m, float time) {  // hittable.rs:147-150
    V3 c0 = ld3(m.c0), c1 = ld3(m.c1);
    return c0 + (c1 - c0) * ((time - m.t0) / (m.t1 - m.t0));
}
// Rect::hit, hittable.rs:230-239: inclusive bounds
VK_HD bool rect_t(const DRect &q, V3 o, V3 d, float tmin, float tmax, float &t_out) {
    uint32_t a0 = q.axes & 3u, a1 = (q.axes >> 2) & 3u, a2 = (q.axes >> 4) & 3u;
    float t = (q.k - comp(o, a2)) / comp(d, a2);
    float a = comp(o, a0) + t * comp(d, a0);
    float b = comp(o, a1) + t * comp(d, a1);
    // one predicate instead of two early returns (the same comparisons, NaN behaviour included: a NaN fails none of the
    // reference's `<` / `>` rejections): on a wave some lane passes the first test nearly always, so the early return saved
    // nothing but cost an EXEC region each
    bool ok = !(t < tmin) & !(t > tmax) & !(a < q.c0) & !(a > q.c1) & !(b < q.d0) & !(b > q.d1);
    if (ok) t_out = t;
    return ok;
}
// Sphere / MovingSphere / Rect by dref
template <class Mem>
VK_HD bool simple_t(const DScene &S, const Mem &M, uint32_t ref, V3 o, V3 d, float a, float time, float tmin, float tmax, float &t) {
    uint32_t k = VKD_KIND(ref), idx = VKD_INDEX(ref);
    if (k == DK_SPHERE) { DSphere s = M.sphere(idx); return sphere_t(s.cx, s.cy, s.cz, s.r, o, d, a, tmin, tmax, t); }
    if (k == DK_RECT) return rect_t(S.rects[idx], o, d, tmin, tmax, t);
    if (k == DK_MOVING) { const DMoving &m = S.moving[idx]; V3 c = moving_center(m, time);
        return sphere_t(c.x, c.y, c.z, m.r, o, d, a, tmin, tmax, t); }
    return false;
}
// impl Hittable for Vec<Arc<..>>, hittable.rs:381-394: first wins ties (strict <)
template <class Mem>
VK_HD bool list_t(const DScene &S, const Mem &M, uint32_t list_ref, V3 o, V3 d, float a, float time, float tmin, float tmax, float &t,
    uint32_t &item) {
    DList l = S.lists[VKD_INDEX(list_ref)];
    float closest = tmax;
    bool found = false;
    for (uint32_t j = 0; j < l.count; j++) {
        uint32_t r = S.list_refs[l.first + j];
        float tt;
        if (simple_t(S, M, r, o, d, a, time, tmin, closest, tt)) {
            if (tt < closest) { closest = tt; item = r; found = true; }
        }
    }
    t = closest;
    return found;
}
// Boxy::hit = sides.hit (hittable.rs:362-365,381-394) for the canonical six rects, unrolled with
// compile-time axes.  Same tests in the same order as list_t over the six vk_rects: Rect::hit's
// inclusive bounds, then the list's strict `rec.t < closest_dist`.
template <int A0, int A1, int A2>
VK_HD void box_face(float k, float c0, float c1, float d0, float d1, V3 o, V3 d, float tmin, float &closest, uint32_t f, uint32_t &face,
    bool &found) {
    float t = (k - comp(o, A2)) / comp(d, A2);
    float a = comp(o, A0) + t * comp(d, A0);
    float b = comp(o, A1) + t * comp(d, A1);
    // Rect::hit's rejections and the list's strict `rec.t < closest` as one predicate (see rect_t)
    bool ok = !(t < tmin) & !(t > closest) & !(a < c0) & !(a > c1) & !(b < d0) & !(b > d1) & (t < closest);
    closest = ok ? t : closest; face = ok ? f : face; found = found | ok;
}
VK_HD bool box_t(const DBox &B, V3 o, V3 d, float tmin, float tmax, float &t, uint32_t &face) {
    float closest = tmax;
    bool found = false;
    box_face<0, 1, 2>(B.p1z, B.p0[0], B.p1x, B.p0[1], B.p1y, o, d, tmin, closest, 0u, face, found);
    box_face<0, 1, 2>(B.p0[2], B.p0[0], B.p1x, B.p0[1], B.p1y, o, d, tmin, closest, 1u, face, found);
    box_face<0, 2, 1>(B.p1y, B.p0[0], B.p1x, B.p0[2], B.p1z, o, d, tmin, closest, 2u, face, found);
    box_face<0, 2, 1>(B.p0[1], B.p0[0], B.p1x, B.p0[2], B.p1z, o, d, tmin, closest, 3u, face, found);
    box_face<1, 2, 0>(B.p1x, B.p0[1], B.p1y, B.p0[2], B.p1z, o, d, tmin, closest, 4u, face, found);
    box_face<1, 2, 0>(B.p0[0], B.p0[1], B.p1y, B.p0[2], B.p1z, o, d, tmin, closest, 5u, face, found);
    t = closest;
    return found;
}
// the vk_rect that face f of a DBox stands for
VK_HD DRect box_face_rect(const DBox &B, uint32_t f) {
    DRect q;
    uint32_t pair = f >> 1;                       // 0: XY, 1: XZ, 2: YZ
    bool hi = (f & 1u) == 0u;                     // even faces sit at p1, odd (FlipFace'd) ones at p0
    float p0x = B.p0[0], p0y = B.p0[1], p0z = B.p0[2];
    q.c0 = pair == 2 ? p0y : p0x; q.c1 = pair == 2 ? B.p1y : B.p1x;
    q.d0 = pair == 0 ? p0y : p0z; q.d1 = pair == 0 ? B.p1y : B.p1z;
    q.k = pair == 0 ? (hi ? B.p1z : p0z) : (pair == 1 ? (hi ? B.p1y : p0y) : (hi ? B.p1x : p0x));
    q.axes = pair == 0 ? (0u | (1u << 2) | (2u << 4)) : (pair == 1 ? (0u | (2u << 2) | (1u << 4)) : (1u | (2u << 2) | (0u << 4)));
    q.mat = B.mat; q._p = 0;
    return q;
}

// boundary.hit() for ConstantMedium (boundary is a Sphere, MovingSphere, Rect or a Boxy list)
template <class Mem>
VK_HD bool boundary_t(const DScene &S, const Mem &M, uint32_t ref, V3 o, V3 d, float a, float time, float tmin, float tmax, float &t,
    uint32_t &item) {
    if (VKD_KIND(ref) == DK_LIST) {
        bool h = list_t(S, M, ref, o, d, a, time, tmin, tmax, t, item);
        if (h) item ^= (ref & DREF_FLIP);
        return h;
    }
    item = ref;
    return simple_t(S, M, ref, o, d, a, time, tmin, tmax, t);
}

// ------------------------------------------------------------------ AxisBB::hit (accel.rs:16-35)
VK_HD bool slab_exact(const DItem &n, V3 o, V3 d, float tmin, float tmax) {
    float lo = tmin, hi = tmax;
    {
        float q0 = (n.mnx - o.x) / d.x, q1 = (n.mxx - o.x) / d.x;
        lo = fmaxf(fminf(q0, q1), lo); hi = fminf(fmaxf(q0, q1), hi);
        if (hi <= lo) return false;
    }
    {
        float q0 = (n.mny - o.y) / d.y, q1 = (n.mxy - o.y) / d.y;
        lo = fmaxf(fminf(q0, q1), lo); hi = fminf(fmaxf(q0, q1), hi);
        if (hi <= lo) return false;
    }
    {
        float q0 = (n.mnz - o.z) / d.z, q1 = (n.mxz - o.z) / d.z;
        lo = fmaxf(fminf(q0, q1), lo); hi = fminf(fmaxf(q0, q1), hi);
        if (hi <= lo) return false;
    }
    return true;
}
// ------------------------------------------------------------------ traversal
// Lane::cell: GRID_DONE = the lane walks a tree (no grid, or a segment walked again); GRID_LAST = a grid lane with no cell left to
// visit (what is left of its current list, i .. end, is still to be tested); below that: the packed cell
constexpr uint32_t GRID_DONE = 0xFFFFFFFFu, GRID_LAST = 0xFFFFFFFEu;
// (a reciprocal computed where it is needed — v_rcp_f32 on the device, 1 ulp — instead of the lane's L.inv: the grid kernels then do not
// keep the three reciprocals and three o/d products of the tree walk alive through their loops)
VK_HD float rcp_here(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VK_EXACT_INV)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

// exact re-treeing: outside the ball in which the gate lemma's bounds hold (DScene::trust_c0; never for r0 = +inf)
VK_HD bool origin_untrusted(const DScene &S, V3 o) {
    const V3 oc = o - v3(S.trust_c0[0], S.trust_c0[1], S.trust_c0[2]);
    return !(length2(oc) <= S.trust_r0sq);
}
VK_HD void grid_begin(Lane &L, const DScene &S);
// redo (exact re-treeing of a scene traversed from global memory, DScene::walk_start != 0): this lane walks the tree as handed over,
// items[0, walk_start - 1), on unscaled distances
template <uint32_t ISHIFT = 0, bool FUSED = false, bool TIGHT = FUSED>
VK_HD void begin_segment(Lane &L, const DScene &S, V3 o, V3 d, float time, bool redo = false) {
    L.wo = o; L.wd = d; L.time = time;
    // (both trees in items[]: a segment that starts outside the trusted ball is the handed-over tree's from the start — segment_unsafe
    // would send it there after a wasted walk)
    // ... and so is, in the near form, a PRIMARY ray when the host found the camera farther than `reach` from everything
    // (DScene::primary_ref: its rebuilt walk could never stand)
    if (TIGHT && S.walk_start != 0u) redo = redo || origin_untrusted(S, o) || (S.primary_ref != 0u && L.depth == 1u);      // (TIGHT: the sphere-only variants)
    set_space<FUSED, TIGHT>(L, o, d, redo ? 1.0f : S.gate_scale);
    L.i = redo ? 0u : (S.walk_start << ISHIFT); L.end = S.n_world_items << ISHIFT; L.pend = 0; L.pend2 = 0; L.cur_inst = -1;
    L.cell = GRID_DONE;
    // the grid form: refs[0, n_always) first (a segment walked again — scenes in global memory keep the tree as handed over in items[],
    // walk_start = its length — walks that tree instead)
    if (TIGHT && S.grid.nu != 0u && !redo) grid_begin(L, S);
    L.T = INFINITY; L.best_prim = 0; L.best_inst = -1; L.best_aux = 0.0f;
    // A ray with a NaN (or infinite) direction or origin hits EVERY box — f32::min/max drop the NaN quotients, accel.rs:21-31 —
    // and no sphere: Sphere::hit's discriminant is NaN (hittable.rs:66-70).  The reference walks its whole tree for such a ray
    // and returns None; in a scene of spheres only (nothing draws during traversal) the walk has no other effect, so it is
    // skipped.  They come from Dielectric::scatter's refract() just past the critical angle (sqrt of a rounding-negative
    // number, util.rs:18-23), 3 samples in 10^4 on the stress scene — each cost a full 1 M-item walk on the stress scene: a 1.7 s tail on
    // EVERY frame, whatever its length.
    if (S.features == 0u) {
        bool dead = !(L.a < INFINITY) || !(fabsf(o.x) + fabsf(o.y) + fabsf(o.z) < INFINITY);
        if (dead) { L.i = L.end; if (L.cell != GRID_DONE) L.cell = GRID_LAST; }
    }
}

template <uint32_t F, class Mem>
VK_HD void accept(Lane &L, float t, uint32_t prim, float aux) {
    L.T = t; L.best_prim = prim; L.best_inst = L.cur_inst; L.best_aux = aux;
}

// ---- exact ties.  In the reference an object hit at EXACTLY the closest-so-far t replaces it only when its own test
// accepts t == tmax: Rect::hit does (hittable.rs:232: `t > tmax` rejects), Sphere::hit (hittable.rs:75: `t < tmax`) and the
// list scan (hittable.rs:386: `rec.t < closest`) do not; BVHNode::hit then keeps the later one (accel.rs:73-77).  Outside a
// rebuilt block the walk IS the reference's order, so the tests below with want_tie = false are the reference's.  Inside a
// rebuilt block (vk_linearize.cpp) the same outcome is computed from the two objects' positions in the reference's order.
VK_HD uint32_t tie_entry(const DScene &S, uint32_t ref) {
    uint32_t k = VKD_KIND(ref), i = VKD_INDEX(ref);
    uint32_t id = k == DK_SPHERE ? i : (k == DK_RECT ? S.tie_base_rect + i : (k == DK_BOX ? S.tie_base_box + i : S.tie_base_list + i));
    return (k == DK_SPHERE || k == DK_RECT || k == DK_BOX || k == DK_LIST) ? S.tie_rank[id] : 0u;
}
// x was just hit at exactly L.T, the t of the current best w: does x replace it?
VK_HD bool tie_replaces(const Lane &L, const DScene &S, uint32_t x) {
    bool x_inclusive = VKD_KIND(x) == DK_RECT;
    if (!S.tie_rank || L.best_prim == 0u || L.best_inst != L.cur_inst) return x_inclusive;      // x is visited after w, as in the reference
    const uint32_t w = L.best_prim;
    uint32_t ex = tie_entry(S, x), ew = tie_entry(S, w);
    if ((ex >> 20) == 0u || (ex >> 20) != (ew >> 20)) return x_inclusive;                      // not both in one rebuilt block: ditto
    if (ex == ew) return false;                                                                // the same object again
    if (ex > ew) return x_inclusive;                                                           // the reference reaches x after w
    return VKD_KIND(w) != DK_RECT;                                                             // ... w after x: w replaced x only if Rect
}

// ConstantMedium::hit, hittable.rs:453-493 (draws ONE number inside traversal)
template <uint32_t F, class Mem>
VK_HD void medium_test(Lane &L, const DScene &S, const Mem &M, uint32_t ref) {
    const DMedium &m = S.media[VKD_INDEX(ref)];
    float t1, t2; uint32_t it;
    uint32_t bk = VKD_KIND(m.boundary);
    if (bk == DK_SPHERE || bk == DK_MOVING) {
        // boundary.hit(-inf, inf) then boundary.hit(t1 + 0.0001, inf) on a sphere: both calls solve the
        // same quadratic (hittable.rs:66-74), so the two roots are computed once; the second call
        // re-tests the near root against its new tmin and then takes the far one.
        float cx, cy, cz, r;
        if (bk == DK_SPHERE) { DSphere s = M.sphere(VKD_INDEX(m.boundary)); cx = s.cx; cy = s.cy; cz = s.cz; r = s.r; }
        else { const DMoving &mv = S.moving[VKD_INDEX(m.boundary)]; V3 c = moving_center(mv, L.time); cx = c.x; cy = c.y; cz = c.z;
            r = mv.r; }
        V3 oc = L.o - v3(cx, cy, cz);
        float half_b = dot(oc, L.d);
        float c = length2(oc) - r * r;
        float disc = half_b * half_b - L.a * c;
        if (!(disc > 0.0f)) return;
        float root = sqrtf(disc);
        float near_t = (-half_b - root) / L.a, far_t = (-half_b + root) / L.a;
        bool near_ok = -INFINITY < near_t && near_t < INFINITY;
        bool far_ok = -INFINITY < far_t && far_t < INFINITY;
        if (near_ok) t1 = near_t; else if (far_ok) t1 = far_t; else return;
        float tmin2 = t1 + 0.0001f;
        if (tmin2 < near_t && near_t < INFINITY) t2 = near_t;
        else if (tmin2 < far_t && far_t < INFINITY) t2 = far_t;
        else return;
    } else {
        if (!boundary_t(S, M, m.boundary, L.o, L.d, L.a, L.time, -INFINITY, INFINITY, t1, it)) return;
        if (!boundary_t(S, M, m.boundary, L.o, L.d, L.a, L.time, t1 + 0.0001f, INFINITY, t2, it)) return;
    }
    float e = t1, x = t2;
    if (e < T_MIN) e = T_MIN;
    if (x > L.T) x = L.T;
    if (e >= x) return;
    if (e < 0.0f) e = 0.0f;
    float ray_length = sqrtf(length2(L.d));
    float distance_inside = (x - e) * ray_length;
    float hit_distance = m.neg_inv_density * vk::logf_(vk::gen_f32(L.rng));
    if (hit_distance > distance_inside) return;
    float t = e + hit_distance / ray_length;
    // BVHNode::hit keeps the earlier hit only when l.t < r.t (accel.rs:73-77)
    if (L.best_prim != 0 && L.T < t) return;
    accept<F, Mem>(L, t, ref, t1);
}

template <uint32_t F, class Mem>
VK_HD void enter_instance(Lane &L, const DScene &S, uint32_t ref) {
    int32_t idx = (int32_t)VKD_INDEX(ref);
    const DInstance &I = S.instances[idx];
    V3 o = L.o, d = L.d;
    for (uint32_t k = 0; k < I.n_ops; k++) apply_op(I.ops[k], o, d);
    set_space<fused_box<F, Mem>()>(L, o, d);
    L.cur_inst = idx;
    L.pend2 = 0;   // the home leaf's right object is restored from home_pend on leave
    if (I.child_end > I.child_begin) { L.i = I.child_begin << Mem::ISHIFT; L.end = I.child_end << Mem::ISHIFT; L.pend = 0; }
    else { L.i = 0; L.end = 0; L.pend = I.child_ref; }
}
template <uint32_t F, class Mem>
VK_HD void leave_instance(Lane &L, const DScene &S) {
    const DInstance &I = S.instances[L.cur_inst];
    int32_t P = I.parent;
    L.i = I.home_next << Mem::ISHIFT; L.pend = I.home_pend; L.pend2 = 0;
    L.end = (P < 0 ? S.n_world_items : S.instances[P].child_end) << Mem::ISHIFT;
    L.cur_inst = P;
    V3 o, d;
    ray_in_instance(S, P, L.wo, L.wd, o, d);
    set_space<fused_box<F, Mem>()>(L, o, d);
}

template <uint32_t F, class Mem>
VK_HD void process_ref(Lane &L, const DScene &S, const Mem &M, uint32_t ref) {
    uint32_t k = VKD_KIND(ref), idx = VKD_INDEX(ref);
    float t;
    bool is_sphere = (k == DK_SPHERE);
    if (is_sphere || ((F & VKF_MOVING) && k == DK_MOVING)) {
        float cx, cy, cz, r;
        if (is_sphere) { DSphere s = M.sphere(idx); cx = s.cx; cy = s.cy; cz = s.cz; r = s.r; }
        else { const DMoving &m = S.moving[idx]; V3 c = moving_center(m, L.time); cx = c.x; cy = c.y; cz = c.z; r = m.r; }
        bool tie;
        if (sphere_t_tie(cx, cy, cz, r, L.o, L.d, L.a, T_MIN, L.T, t, tie)) {
            if (!tie || tie_replaces(L, S, ref)) accept<F, Mem>(L, t, ref, 0.0f);     // tie: rare, see tie_replaces
        }
        return;
    }
    if ((F & VKF_RECT) && k == DK_RECT) {
        if (rect_t(S.rects[idx], L.o, L.d, T_MIN, L.T, t)) {                          // inclusive bounds: t == L.T comes through
            if (t != L.T || tie_replaces(L, S, ref)) accept<F, Mem>(L, t, ref, 0.0f);
        }
        return;
    }
    if ((F & VKF_BOX) && k == DK_BOX) {
        uint32_t face = 0;
        DBox B = M.box(idx);
        // the list scan is strict against the tmax it was given; nextafter(T) as tmax lets a hit AT T through for the tie rule
        float tmax = S.tie_rank ? nextafter_up(L.T) : L.T;
        if (box_t(B, L.o, L.d, T_MIN, tmax, t, face)) {
            if (t < L.T || (t == L.T && tie_replaces(L, S, ref))) accept<F, Mem>(L, t, ref, vk::bits_f32(face));
        }
        return;
    }
    if ((F & VKF_LIST) && k == DK_LIST) {
        uint32_t item;
        float tmax = S.tie_rank ? nextafter_up(L.T) : L.T;
        if (list_t(S, M, ref, L.o, L.d, L.a, L.time, T_MIN, tmax, t, item)) {
            if (t < L.T || (t == L.T && tie_replaces(L, S, ref))) accept<F, Mem>(L, t, item ^ (ref & DREF_FLIP), 0.0f);
        }
        return;
    }
    if ((F & VKF_MEDIUM) && k == DK_MEDIUM) { medium_test<F, Mem>(L, S, M, ref); return; }
    if ((F & VKF_INSTANCE) && k == DK_INSTANCE) { enter_instance<F, Mem>(L, S, ref); return; }
}

// The threaded pre-order walk is split into two kinds of step so that a wave can run each
// kind with the lanes that need it (vk_api.hip schedules them by ballot):
//   box_step   one 32-byte item: box test, then advance / take the skip link; a leaf whose
//              box is hit queues its objects in pend (left) and pend2 (right)
//   prim_step  the object in pend: intersect it (or enter it, if it is an instance); the
//              right object runs after the left one and sees the tmax it left behind
//              (accel.rs:64-70)
// A lane has box work when pend == 0 and prim work when pend != 0; objects are always
// finished before the next item is fetched, which is the reference's order.
VK_HD bool traversing(const Lane &L) { return L.i < L.end || L.pend != 0 || L.cur_inst >= 0 || L.cell < GRID_LAST; }
VK_HD bool has_prim_work(const Lane &L) { return L.pend != 0; }
// Sphere / MovingSphere / Rect tests are ~50 instructions; Boxy, lists, media and instance entry cost several times that
// In the everything-variants a Boxy (kind 7: six rect tests, no draw, no change of space) counts as light and is served inside the
// box loop like a sphere or a rect: C3's 400 ground boxes then stop competing with the media and the instance entries for the
// heavy phase (C3 727 -> 766 Msamples/s).  The Cornell-type variants keep it heavy (C4 -2.9 % otherwise: its two boxes sit
// inside instances, i.e. behind a heavy step anyway).  A scheduling class only: results do not depend on it.
template <uint32_t F>
VK_HD bool prim_is_heavy(uint32_t ref) {
    return ((F & VKF_ALL_SCENE) == VKF_ALL_SCENE) ? (ref - ((uint32_t)DK_LIST << 28) < (3u << 28)) : (VKD_KIND(ref) >= DK_LIST);
}

// ------------------------------------------------------------------ exact re-treeing (DScene::t_pad > 0; scenes of spheres only)
// items[] is then a tree REBUILT over the reference's leaf units (vk_linearize.cpp rt_collect): every object is gated by the box the
// reference gates it with — grown a little — and boxes are nested.  DESIGN.md section 5 has the argument in full; in short:
//  * CANDIDATES.  For a ray, a sphere's candidate is the root Sphere::hit would report with tmax = +inf (the first root above tmin);
//    whatever tmax it is called with, Sphere::hit reports the candidate or nothing.  BVHNode::hit therefore ends with the minimum over
//    the candidates it ACCEPTS, and it accepts X only when the boxes above X pass, the last of which is X's unit's.
//  * SOUND GATES.  The rebuilt walk tests a unit when its grown box passes against T (1 + t_pad), T the closest hit so far.  The growth
//    and the padding are chosen (vk_linearize.h rt_unit_growth, the "gate lemma") so that, for a ray with ordinary components that
//    starts inside the scene's trusted ball, the gate of X passes whenever T exceeds X's candidate and the reference's box of the unit
//    passes at all.  By induction the rebuilt walk then ends with a T no larger than any candidate the reference could accept.
//  * SAFE WINNER.  If the rebuilt walk's winner W passes its OWN box (center -+ radius: inside every box the reference has above W) by a
//    margin, with its hit behind that box's entry, then the reference's walk reaches W whatever it found before, with a tmax that is
//    still >= t_W, and accepts it: both walks end with W.  (A miss is safe: no candidate exists.)
//  * Everything else — a winner that is not safe, a ray from outside the ball, a ray with a tiny, huge or non-finite component — is
//    decided by the tree as handed over: the segment is walked again in place where items[] holds both trees (DScene::walk_start,
//    scenes in global memory), or its sample is dropped and rendered by a second launch (vk_kernels.h, scenes staged in LDS).
// tests/test_retree.py compares per sample with the oracle on the handed-over tree; tests/test_gate_lemma.py attacks the lemma.
template <uint32_t F>
VK_HD float gate_of(const Lane &L, const DScene &S) {
    return ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u && S.t_pad > 0.0f && L.i >= S.walk_start) ? L.T * (1.0f + S.t_pad) : L.T;
}
// The near form's clearance test (DScene::clear_k): beyond `reach` the ray stays outside the box around the small spheres' surfaces grown by
// M = clear_k (D_far + clear_r2) + clear_slack, D_far = the origin's distance from the box's far corner — then no sphere farther than
// rho_near from the origin holds a candidate (docs/gate_lemma.md section 7).  A slab test in the ray's parameter: inside the grown box
// for t in [tin, tout]; clear iff that interval is empty or ends before reach.  `inv` are the lane's reciprocals (v_rcp_f32: 1 ulp, times
// gate_scale: another), so the parameters carry gate_scale and an error of 4 u |b - o| in position — a thousandth of the 2 % in clear_k.
// (Ordinary rays only: no zero component.)
VK_HD bool clear_of_small_spheres(const DScene &S, V3 o, V3 inv, float dn) {
    const float fx = fmaxf(fabsf(o.x - S.small_clo[0]), fabsf(o.x - S.small_chi[0]));
    const float fy = fmaxf(fabsf(o.y - S.small_clo[1]), fabsf(o.y - S.small_chi[1]));
    const float fz = fmaxf(fabsf(o.z - S.small_clo[2]), fabsf(o.z - S.small_chi[2]));
    const float dfar = sqrtf(fx * fx + fy * fy + fz * fz);
    const float m = __builtin_fmaf(S.clear_k, dfar + S.clear_r2, S.clear_slack);
    const float x0 = ((S.small_clo[0] - m) - o.x) * inv.x, x1 = ((S.small_chi[0] + m) - o.x) * inv.x;
    const float y0 = ((S.small_clo[1] - m) - o.y) * inv.y, y1 = ((S.small_chi[1] + m) - o.y) * inv.y;
    const float z0 = ((S.small_clo[2] - m) - o.z) * inv.z, z1 = ((S.small_chi[2] + m) - o.z) * inv.z;
    const float tin = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float tout = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    // (a NaN anywhere: not clear)
    return (tin > tout) || (tout * dn < S.reach * S.gate_scale);
}

// Must the segment just walked (on the rebuilt tree: the lane's reciprocals are the scaled ones) be decided by the tree as handed over?
// Decided on the scaled fast-path quantities with the box test's own margin on the side of "unsafe": a false positive costs one walk
// (or one sample) done twice.
template <uint32_t F, class Mem>
VK_HD bool segment_unsafe(const Lane &L, const DScene &S, const Mem &M) {
    if (!(S.t_pad > 0.0f)) return false;
    // the ray: inside the trusted ball, components the fast box test trusts (xnan is NaN otherwise)
    const bool outside = origin_untrusted(S, L.o);
    if (S.walk_start != 0u && outside) return false;       // (begin_segment: this one WAS walked on the tree as handed over)
    bool unsafe = outside || !(L.xnan == L.xnan);
    // the near form (DScene::reach > 0): own-box gates are sound for the spheres within rho_near of the origin only, and only a hit
    // within `reach` of the origin guarantees that no other sphere could hold a closer candidate (vk_linearize.cpp rt_grow_near: the
    // reach lemma); a miss (T = +inf) never stands
    if (S.reach > 0.0f) {
        const float dn = sqrtf(L.a);
        bool stands = L.T * dn <= S.reach;
        if (!stands) stands = clear_of_small_spheres(S, L.o, L.inv, dn);
        unsafe = unsafe || !stands;
    }
    if (L.best_prim != 0u) {
        const DSphere sp = M.sphere(VKD_INDEX(L.best_prim));
        const float bx0 = sp.cx - sp.r, bx1 = sp.cx + sp.r, by0 = sp.cy - sp.r, by1 = sp.cy + sp.r, bz0 = sp.cz - sp.r, bz1 = sp.cz + sp.r;
        float x0, x1, y0, y1, z0, z1;
        if (S.grid.nu != 0u) {
            // (the grid form: reciprocals made here, see rcp_here; the unfused form, whose error the margins below cover a fortiori)
            const float ix = rcp_here(L.d.x) * S.gate_scale, iy = rcp_here(L.d.y) * S.gate_scale, iz = rcp_here(L.d.z) * S.gate_scale;
            x0 = (bx0 - L.o.x) * ix; x1 = (bx1 - L.o.x) * ix;
            y0 = (by0 - L.o.y) * iy; y1 = (by1 - L.o.y) * iy;
            z0 = (bz0 - L.o.z) * iz; z1 = (bz1 - L.o.z) * iz;
        } else if constexpr (fused_box<F, Mem>()) {
            x0 = __builtin_fmaf(bx0, L.inv.x, -L.oi.x); x1 = __builtin_fmaf(bx1, L.inv.x, -L.oi.x);
            y0 = __builtin_fmaf(by0, L.inv.y, -L.oi.y); y1 = __builtin_fmaf(by1, L.inv.y, -L.oi.y);
            z0 = __builtin_fmaf(bz0, L.inv.z, -L.oi.z); z1 = __builtin_fmaf(bz1, L.inv.z, -L.oi.z);
        } else {
            x0 = (bx0 - L.o.x) * L.inv.x; x1 = (bx1 - L.o.x) * L.inv.x;
            y0 = (by0 - L.o.y) * L.inv.y; y1 = (by1 - L.o.y) * L.inv.y;
            z0 = (bz0 - L.o.z) * L.inv.z; z1 = (bz1 - L.o.z) * L.inv.z;
        }
        // AxisBB::hit of the own box with tmax = the winner's t (scaled like the reciprocals): max(entry, tmin) < min(exit, t), i.e.
        // entry < min(exit, t) and tmin < exit (tmin < t holds for every accepted hit) — as box_step_core decides it, but a decision
        // inside the margin counts as a miss.  (An origin inside the box has a negative entry: nothing about it is in doubt.)
        const float entry = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
        const float ex = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        const float hi = fminf(ex, L.T * S.gate_scale);
        const bool passes = (hi - entry > __builtin_fmaf(fabsf(hi), 4.0e-6f, L.xnan)) &&     // (false for a NaN margin)
                            (ex - S.tmin_gate > __builtin_fmaf(fabsf(ex), 4.0e-6f, L.xnan));
        bool ok = passes;
        if (!ok && L.xnan == L.xnan) {
            // inside the fast arithmetic's margin: the reference's own AxisBB::hit of the sphere's box with tmax = the winner's t.
            // No margin is needed there: the boxes around the sphere in the tree as handed over contain its own, fl((b - o) / d) is
            // monotone in b and the test is monotone in tmax, so they pass whenever this one does.  (A third of the segments the
            // margin alone calls unsafe are; the rest need no second walk / second launch.)
            DItem own;
            own.mnx = bx0; own.mxx = bx1; own.mny = by0; own.mxy = by1; own.mnz = bz0; own.mxz = bz1; own.w0 = 0u; own.w1 = 0u;
            ok = slab_exact(own, L.o, L.d, T_MIN, L.T);
        }
        if (!ok && L.xnan == L.xnan && S.unit_item != nullptr) {
            // Second chance: the box that gates the sphere in the tree as handed over — its leaf's (of a bare child of a node: that
            // node's; DScene::unit_item).  The reference reaches the sphere through that node and its ancestors, which contain it: if the leaf's box passes AxisBB::hit with tmax = the winner's t,
            // they all do (same monotonicity), whatever the sphere's own box says.  (The hit point of a FAR origin lies up to
            // eta(rho) off its sphere, sometimes outside the sphere's own box and still inside the leaf's: of the 1 M-sphere scene's
            // segments 0.66 % fail the first test, 0.47 % both — hits reported BEFORE the ray enters the leaf's box, which the reference
            // accepts or not depending on what it found earlier: the tree as handed over has to say.)
            const uint32_t ui = S.unit_item[VKD_INDEX(L.best_prim)];
            if (ui != 0xFFFFFFFFu) ok = slab_exact(S.unit_tree[ui], L.o, L.d, T_MIN, L.T);
        }
        unsafe = unsafe || !ok || !(sp.r > 0.0f);
    }
    return unsafe;
}

// One box step of a lane that HAS box work and is inside its range (pend == 0, i < end).
// The wave runs it under the EXEC mask of exactly those lanes (box_steps below): a lane that queues
// objects or reaches the end of its range drops out of the mask and costs nothing more, and the
// bookkeeping is a handful of VALU selects (a fully predicated version that kept all 64 lanes in
// EXEC and re-did item 0 on idle lanes spent 14 of its 45 VALU instructions on selects).
// fminf(a, t) for the loop-carried tmax t (positive, possibly +inf, never NaN), as a SIGNED INTEGER minimum of the bit patterns:
// fminf() makes the compiler re-quiet `t` with a v_max_f32 t, t in every box step (it cannot see through the loop that t is
// never a signalling NaN).  For t > 0 the integer order is the float order whenever a >= 0; a negative a (sign bit set) is
// the smaller one in both orders; a = +NaN (> +inf as an integer) yields t like fminf.  (a = -NaN would come through, where
// fminf gives t: then both bounds of the axis were NaN and the step takes the exact path whatever this returns.)
VK_HD float min_with_tmax(float a, float t) {
    int32_t ia = (int32_t)vk::f32_bits(a), it = (int32_t)vk::f32_bits(t);
    return vk::bits_f32((uint32_t)(ia < it ? ia : it));
}

template <uint32_t F, class Mem>
VK_HD bool box_step_core(Lane &L, const DScene &S, const Mem &M) {      // returns: a leaf's box was hit (objects queued in pend)
    DItem n = M.item(L.i);
    // sphere-only variants: tmin on the scale the lane's reciprocals are on (exact re-treeing, below; T_MIN itself otherwise)
    float tmin = ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) ? S.tmin_gate : T_MIN;
    if constexpr ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u && Mem::ISHIFT == 0u) {
        // a lane on the tree as handed over: the reference's own test
        if (S.walk_start != 0u) tmin = L.i < S.walk_start ? T_MIN : tmin;
    }
    // AxisBB::hit decided from reciprocal multiplies; same boolean as the reference's (see slab_exact):
    // both forms compute fl(b-o) identically and q~ = fl(fl(b-o)*fl(1/d)) differs from the reference's
    // fl(fl(b-o)/d) by < 3*2^-24 relative; the three per-axis early-outs equal one test
    // max(lo..) < min(hi..) because lo only grows and hi only shrinks.
    // (scalar on purpose: v_pk_add_f32/v_pk_mul_f32 on the (min,max) pair of each axis was measured SLOWER twice —
    // -5 % as compiler vectors (the broadcast operands get materialised as register pairs), -14 % as inline asm
    // with op_sel broadcasts of (o.x,o.y)/(inv.x,inv.y)/(o.z,inv.z): the packed ops do not issue at twice the rate here)
    float x0, x1, y0, y1, z0, z1;
    if constexpr (fused_box<F, Mem>()) {     // one fma per bound (set_space<true> explains the margin)
        x0 = __builtin_fmaf(n.mnx, L.inv.x, -L.oi.x); x1 = __builtin_fmaf(n.mxx, L.inv.x, -L.oi.x);
        y0 = __builtin_fmaf(n.mny, L.inv.y, -L.oi.y); y1 = __builtin_fmaf(n.mxy, L.inv.y, -L.oi.y);
        z0 = __builtin_fmaf(n.mnz, L.inv.z, -L.oi.z); z1 = __builtin_fmaf(n.mxz, L.inv.z, -L.oi.z);
    } else {
        x0 = (n.mnx - L.o.x) * L.inv.x; x1 = (n.mxx - L.o.x) * L.inv.x;
        y0 = (n.mny - L.o.y) * L.inv.y; y1 = (n.mxy - L.o.y) * L.inv.y;
        z0 = (n.mnz - L.o.z) * L.inv.z; z1 = (n.mxz - L.o.z) * L.inv.z;
    }
    float lo = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));    // >= tmin > 0, never NaN
    float hi = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), min_with_tmax(fmaxf(z0, z1), L.T));    // <= T, never NaN
    // lo >= T_MIN > 0, so with e = 3*2^-24 the sign of the exact (hi - lo) equals the sign of this one whenever
    // |hi - lo| > 2e*hi/(1-e) ~ 3.6e-7*hi (also for hi <= 0: a certain miss).  The test below asks for 2e-6*hi,
    // and is false (-> exact fallback) when xnan is NaN, i.e. for rays whose 1/d is not a full-precision normal
    // number.  `>=` so that hi = +inf (always-hit leaves while T is still infinite) stays on the fast path.
    float dlt = hi - lo;
    bool h = dlt > 0.0f;
    const float margin = __builtin_fmaf(hi, 2.0e-6f, L.xnan);
#if defined(__HIP_DEVICE_COMPILE__)
    // One lane in a hundred thousand gets here: the wave asks first whether ANY of its lanes does (the compare's lane mask,
    // one scalar branch) and only then opens an EXEC region — which otherwise costs three scalar instructions in every step.
    // (The empty asm keeps the compiler from folding the two tests back into one region.)
    if (__builtin_amdgcn_fcmpf(fabsf(dlt), margin, 12 /* unordered or less than: !(a >= b) */) != 0ull) {
        asm volatile("");
        if (!(fabsf(dlt) >= margin)) h = slab_exact(n, L.o, L.d, T_MIN, gate_of<F>(L, S));
    }
#else
    if (!(fabsf(dlt) >= margin))
        h = slab_exact(n, L.o, L.d, T_MIN, gate_of<F>(L, S));         // within rounding distance: the reference's divisions
#endif
    bool inner = (n.w0 >> 28) == 0u;
    bool leaf_hit = h && !inner;
    L.i = (inner && !h) ? n.w0 : L.i + (1u << Mem::ISHIFT);   // inner: hit -> left subtree, miss -> skip link (in cursor units)
    L.pend = leaf_hit ? n.w0 : 0u;                       // leaf: left object first, then the right one
    L.pend2 = n.w1;                                      // only read after pend, i.e. after a leaf hit
    return leaf_hit;
}

// range end of the lane's current item range: wave-uniform when the scene has no instances
template <uint32_t F, class Mem>
VK_HD uint32_t range_end(const Lane &L, const DScene &S) { return (F & VKF_INSTANCE) ? L.end : (S.n_world_items << Mem::ISHIFT); }

// N box steps, each run by the lanes that still have box work: nested ifs, i.e. one shrinking EXEC mask
template <uint32_t F, class Mem, int N>
VK_HD void box_steps(Lane &L, const DScene &S, const Mem &M, bool go) {
    if (go) {
        bool queued = box_step_core<F, Mem>(L, S, M);   // (the mask is at hand: cheaper than comparing pend with 0 again)
        // (&: one mask, one branch)
        if (N > 1) box_steps<F, Mem, (N > 1 ? N - 1 : 1)>(L, S, M, (!queued) & (L.i < range_end<F, Mem>(L, S)));
    }
}

// sequential form of one step (CPU emulator); `on` = this lane has box work
template <uint32_t F, class Mem>
VK_HD void box_step(Lane &L, const DScene &S, const Mem &M, bool on) {
    bool at_end = L.i >= L.end;
    if (F & VKF_INSTANCE) {
        if (on && at_end) { if (L.cur_inst >= 0) leave_instance<F, Mem>(L, S); return; }   // rare
    }
    if (on && !at_end) box_step_core<F, Mem>(L, S, M);
}

template <uint32_t F, class Mem>
VK_HD void prim_step(Lane &L, const DScene &S, const Mem &M) {
    uint32_t ref = L.pend;
    if ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) {
        // sphere-only variants: both objects of the leaf in ONE step (left first, the right one sees the tmax the left one
        // left behind, accel.rs:64-70): the lane is back in the box loop after one primitive phase instead of two
        uint32_t ref2 = L.pend2;
        L.pend = 0; L.pend2 = 0;
        // both records are fetched before the first test so that their two memory (or LDS) latencies overlap
        DSphere sa = M.sphere(VKD_INDEX(ref));
        DSphere sb = M.sphere(VKD_INDEX(ref2 ? ref2 : ref));
        float t; bool tie;
        const float ya = refined_rcp(L.a);
        const bool fast = S.fast_div != 0u && (L.xnan == L.xnan);
        if (sphere_t_tie_y(sa.cx, sa.cy, sa.cz, sa.r, L.o, L.d, L.a, ya, fast, T_MIN, L.T, t, tie)) {
            if (!tie || tie_replaces(L, S, ref)) accept<F, Mem>(L, t, ref, 0.0f);
        }
        if (ref2) {
            if (sphere_t_tie_y(sb.cx, sb.cy, sb.cz, sb.r, L.o, L.d, L.a, ya, fast, T_MIN, L.T, t, tie)) {
                if (!tie || tie_replaces(L, S, ref2)) accept<F, Mem>(L, t, ref2, 0.0f);
            }
        }
        return;
    }
    L.pend = L.pend2;
    L.pend2 = 0;
    process_ref<F, Mem>(L, S, M, ref);
    // a light object followed by another light one (a leaf of two spheres / rects): both in this step, as above; and an
    // instance that holds ONE object (a rotated, translated Boxy: enter_instance queued it in pend): entered and tested in one
    // step instead of two heavy phases
    const bool chain_light = !prim_is_heavy<F>(ref) && L.pend != 0u && !prim_is_heavy<F>(L.pend);
    const bool chain_inst = (F & VKF_INSTANCE) != 0u && VKD_KIND(ref) == DK_INSTANCE && L.pend != 0u;
    if (chain_light || chain_inst) {
        uint32_t ref2 = L.pend;
        L.pend = 0;
        process_ref<F, Mem>(L, S, M, ref2);
    }
}

// ---- The grid form's walk (DGrid, docs/gate_lemma.md section 8).  One step: queue the next two references of the current cell, or
// move on — to the next cell of this column of cells, to the next column along the ray's major axis (x or z, whichever component is
// larger), or out.  A column is the strip U0 <= u <= U1 of the grid; the ray is inside the strip dilated by e = dl for t in
// [t_lo, t_hi] (clipped to the segment's range [ta, min(tb, T)]), and over that range its minor coordinate covers [v_lo, v_hi]: the
// cells of the column within e of that are visited.  Every cell within e of the ray's path is: a point of the path lies in some
// column's dilated strip at a parameter inside that column's range.  The reciprocals are the lane's (1 ulp, times gate_scale: undone
// here), floor() of a quotient may land a cell off: dl carries 2 m_reg for that (vk_linearize.cpp rt_build_grid).
// the segment's set-up (begin_segment): its parameter range inside the layer's dilated box, the dilation, the first column
VK_HD void grid_begin(Lane &L, const DScene &S) {
    const DGrid &G = S.grid;
    L.i = 0u; L.end = G.n_always;
    const bool mz = fabsf(L.d.z) > fabsf(L.d.x);
    const float unscale = 1.0f + S.t_pad;                 // (the reciprocals set_space has just made carry gate_scale)
    // the dilation this origin can need at most (the far corner of the layer's box), the segment's range inside the box dilated by
    // that, the dilation its far end needs, the range again across the layer
    const float fx = fmaxf(fabsf(L.o.x - G.lo[0]), fabsf(L.o.x - G.hi[0])), fy = fmaxf(fabsf(L.o.y - G.lo[1]), fabsf(L.o.y - G.hi[1]));
    const float fz = fmaxf(fabsf(L.o.z - G.lo[2]), fabsf(L.o.z - G.hi[2]));
    const float dlc = __builtin_fmaf(G.k, sqrtf(fx * fx + fy * fy + fz * fz) + G.r2, G.slack);
    const float ix_ = L.inv.x * unscale, iy_ = L.inv.y * unscale, iz_ = L.inv.z * unscale;
    const float x0 = ((G.lo[0] - dlc) - L.o.x) * ix_, x1 = ((G.hi[0] + dlc) - L.o.x) * ix_;
    const float y0 = ((G.lo[1] - dlc) - L.o.y) * iy_, y1 = ((G.hi[1] + dlc) - L.o.y) * iy_;
    const float z0 = ((G.lo[2] - dlc) - L.o.z) * iz_, z1 = ((G.hi[2] + dlc) - L.o.z) * iz_;
    float ta = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    float tb = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    float dl = fminf(__builtin_fmaf(G.k, __builtin_fmaf(tb, sqrtf(L.a), G.r2), G.slack), dlc);
    if (!(dl >= 0.0f)) dl = dlc;
    const float w0 = ((G.lo[1] - dl) - L.o.y) * iy_, w1 = ((G.hi[1] + dl) - L.o.y) * iy_;
    ta = fmaxf(ta, fminf(w0, w1)); tb = fminf(tb, fmaxf(w0, w1));
    L.ta = ta; L.tb = tb; L.dl = dl;
    if (G.n_gated != 0u) {
        // the large spheres' common gate: a candidate of one of them lies within dlc of it, hence of their box
        const float p0 = ((G.alo[0] - dlc) - L.o.x) * ix_, p1 = ((G.ahi[0] + dlc) - L.o.x) * ix_;
        const float q0 = ((G.alo[1] - dlc) - L.o.y) * iy_, q1 = ((G.ahi[1] + dlc) - L.o.y) * iy_;
        const float r0 = ((G.alo[2] - dlc) - L.o.z) * iz_, r1 = ((G.ahi[2] + dlc) - L.o.z) * iz_;
        const float gin = fmaxf(fmaxf(fminf(p0, p1), fminf(q0, q1)), fmaxf(fminf(r0, r1), 0.0f));
        const float gout = fminf(fminf(fmaxf(p0, p1), fmaxf(q0, q1)), fmaxf(r0, r1));
        if (gin > gout * 1.000001f + 1e-30f) L.end = G.n_always - G.n_gated;      // (a NaN anywhere: tested)
    }
    if (!(ta <= tb)) { L.cell = GRID_LAST; return; }       // (the path never comes near the layer: only refs[0, n_always))
    const float ou = mz ? L.o.z : L.o.x, du = mz ? L.d.z : L.d.x, gu = mz ? G.ov : G.ou;
    const int nmaj = (int)(mz ? G.nv : G.nu);
    const float ua = __builtin_fmaf(du, ta, ou) - (du > 0.0f ? dl : -dl);
    const int i0 = (int)fminf(fmaxf(floorf((ua - gu) * G.inv_cell), 0.0f), (float)(nmaj - 1));
    // "the cell after the last one of column i0 - dir": the first grid_advance enters column i0
    L.cell = ((uint32_t)(i0 - (du > 0.0f ? 1 : -1)) & 1023u) | (1u << 10);
}
template <class Mem>
VK_HD void grid_advance(Lane &L, const DScene &S, const Mem &M) {
    const DGrid &G = S.grid;
    const bool mz = fabsf(L.d.z) > fabsf(L.d.x);
    const float ou = mz ? L.o.z : L.o.x, ov = mz ? L.o.x : L.o.z, du = mz ? L.d.z : L.d.x, dv = mz ? L.d.x : L.d.z;
    const float gu = mz ? G.ov : G.ou, gv = mz ? G.ou : G.ov;
    const int nmaj = (int)(mz ? G.nv : G.nu), nmin = (int)(mz ? G.nu : G.nv);
    const int dir = du > 0.0f ? 1 : -1;
    // (the major index is kept in 10 bits, two's complement: -1 = 1023 is the column before the first one)
    int i = (int)(L.cell & 1023u); if (i == 1023) i = -1;
    int j = (int)((L.cell >> 10) & 1023u) + 1, jhi = (int)((L.cell >> 20) & 1023u);
    if (j > jhi) {
        // the next column; ONE per step (a column outside the segment's range leaves the state "after its last cell")
        i += dir;
        if (i < 0 || i >= nmaj) { L.cell = GRID_LAST; return; }
        const float e = L.dl, ru = rcp_here(du);
        const float tend = fminf(L.tb, L.T);
        const float U0 = __builtin_fmaf((float)i, G.cell, gu), U1 = U0 + G.cell;
        const float tA = ((U0 - e) - ou) * ru, tB = ((U1 + e) - ou) * ru;
        const float t_lo = fmaxf(L.ta, fminf(tA, tB)), t_hi = fminf(tend, fmaxf(tA, tB));
        if (!(t_lo <= t_hi)) {
            if (!(fminf(tA, tB) <= tend)) { L.cell = GRID_LAST; return; }       // the path ends before this column
            L.cell = ((uint32_t)i & 1023u) | (1u << 10);                          // (the column lies before the range's start)
            L.i = 0u; L.end = 0u;
            return;
        }
        const float v0 = __builtin_fmaf(dv, t_lo, ov), v1 = __builtin_fmaf(dv, t_hi, ov);
        j = (int)fminf(fmaxf(floorf(((fminf(v0, v1) - e) - gv) * G.inv_cell), 0.0f), (float)(nmin - 1));
        jhi = (int)fminf(fmaxf(floorf(((fmaxf(v0, v1) + e) - gv) * G.inv_cell), 0.0f), (float)(nmin - 1));
    }
    const uint32_t c = mz ? (uint32_t)j * G.nv + (uint32_t)i : (uint32_t)i * G.nv + (uint32_t)j;
    L.i = M.grid_cell(S, c); L.end = M.grid_cell(S, c + 1u);
    L.cell = (uint32_t)i | ((uint32_t)j << 10) | ((uint32_t)jhi << 20);
}
template <class Mem>
VK_HD bool grid_step(Lane &L, const DScene &S, const Mem &M) {       // returns: references were queued
    if (!(L.i < L.end) && L.cell < GRID_LAST) grid_advance(L, S, M);
    if (L.i < L.end) {
        L.pend = M.grid_ref(S, L.i);
        L.pend2 = L.i + 1u < L.end ? M.grid_ref(S, L.i + 1u) : 0u;
        L.i = L.i + 2u < L.end ? L.i + 2u : L.end;
        return true;
    }
    return false;
}

// sequential form (CPU emulator): one step of whichever kind is due
template <uint32_t F, class Mem>
VK_HD void traverse_step(Lane &L, const DScene &S, const Mem &M) {
    if (has_prim_work(L)) prim_step<F, Mem>(L, S, M);
    else if (spheres_only<F>() && L.cell != GRID_DONE) (void)grid_step(L, S, M);
    else box_step<F, Mem>(L, S, M, true);
}

// ------------------------------------------------------------------ deferred hit record
struct Rec { V3 p, n; float u, v; bool front; uint32_t mat; };

VK_HD void spherical(V3 p, float &u, float &v) {  // hittable.rs:54-61
    float phi = vk::atan2f_(p.z, p.x);
    float theta = vk::asinf_(p.y);
    u = 1.0f - ((phi + PI_F) / (2.0f * PI_F));
    v = (theta + PI_F / 2.0f) / PI_F;
}
VK_HD void face(V3 d, V3 outward, V3 &n, bool &front) {  // hittable.rs:23-30
    front = dot(d, outward) < 0.0f;
    n = front ? outward : -outward;
}

// Rect::hit's record (hittable.rs:240-255)
VK_HD void rect_record(const DRect &q, V3 o, V3 d, float t, Rec &R) {
    uint32_t a0 = q.axes & 3u, a1 = (q.axes >> 2) & 3u, a2 = (q.axes >> 4) & 3u;
    float a = comp(o, a0) + t * comp(d, a0);
    float b = comp(o, a1) + t * comp(d, a1);
    R.u = (a - q.c0) / (q.c1 - q.c0);
    R.v = (b - q.d0) / (q.d1 - q.d0);
    R.p = o + d * t;
    V3 outward = v3(a2 == 0 ? 1.0f : 0.0f, a2 == 1 ? 1.0f : 0.0f, a2 == 2 ? 1.0f : 0.0f);
    face(d, outward, R.n, R.front);
    R.mat = q.mat;
}

// record of a Sphere/MovingSphere/Rect hit at t in the ray's own space
template <class Mem>
VK_HD void simple_record(const DScene &S, const Mem &M, uint32_t ref, V3 o, V3 d, float time, float t, bool want_uv, Rec &R) {
    uint32_t k = VKD_KIND(ref), idx = VKD_INDEX(ref);
    R.u = 0.0f; R.v = 0.0f;
    if (k == DK_RECT) {
        rect_record(S.rects[idx], o, d, t, R);
    } else {
        V3 c; float r;
        if (k == DK_SPHERE) { DSphere s = M.sphere(idx); c = v3(s.cx, s.cy, s.cz); r = s.r; R.mat = M.smat(idx); }
        else { const DMoving &m = S.moving[idx]; c = moving_center(m, time); r = m.r; R.mat = m.mat; }
        R.p = o + d * t;
        V3 outward = (R.p - c) / r;
        face(d, outward, R.n, R.front);
        if (want_uv) spherical(outward, R.u, R.v);
    }
    if (ref & DREF_FLIP) R.front = !R.front;
}

VK_HD bool mat_wants_uv(const DMaterial &m) {
    return m.kind == VK_MAT_SPEC_DIFFUSE || (m.kind != VK_MAT_DIELECTRIC && (m.tex_kind == VK_TEX_IMAGE || m.tex_kind == VK_TEX_CHECKER));
}

template <uint32_t F, class Mem>
VK_HD void build_record(const Lane &L, const DScene &S, const Mem &M, Rec &R) {
    V3 o = L.wo, d = L.wd;
    if (F & VKF_INSTANCE) ray_in_instance(S, L.best_inst, L.wo, L.wd, o, d);
    uint32_t ref = L.best_prim;
    uint32_t k = VKD_KIND(ref);
    if ((F & VKF_MEDIUM) && k == DK_MEDIUM) {       // hittable.rs:479-489
        const DMedium &m = S.media[VKD_INDEX(ref)];
        R.p = o + d * L.T;
        R.n = v3(1.0f, 0.0f, 0.0f);
        R.front = true;
        R.mat = m.mat;
        R.u = 0.0f; R.v = 0.0f;
        if ((F & VKF_TEXTURES) && mat_wants_uv(S.materials[m.mat])) {   // rec1.u, rec1.v of the boundary's entry hit
            float a = length2(d), t1; uint32_t it;
            if (boundary_t(S, M, m.boundary, o, d, a, L.time, -INFINITY, INFINITY, t1, it)) {
                Rec B;
                simple_record(S, M, it, o, d, L.time, t1, true, B);
                R.u = B.u; R.v = B.v;
            }
        }
        if (ref & DREF_FLIP) R.front = !R.front;
    } else if ((F & VKF_BOX) && k == DK_BOX) {
        uint32_t f = vk::f32_bits(L.best_aux);
        DRect q = box_face_rect(M.box(VKD_INDEX(ref)), f);
        rect_record(q, o, d, L.T, R);
        if (f & 1u) R.front = !R.front;              // odd sides are FlipFace-wrapped (hittable.rs:327-352)
        if (ref & DREF_FLIP) R.front = !R.front;
    } else {
        bool want_uv = false;
        if (F & VKF_TEXTURES) {
            uint32_t mi = (k == DK_SPHERE) ? M.smat(VKD_INDEX(ref))
                                           : (k == DK_RECT ? S.rects[VKD_INDEX(ref)].mat : S.moving[VKD_INDEX(ref)].mat);
            want_uv = mat_wants_uv(S.materials[mi]);
        }
        simple_record(S, M, ref, o, d, L.time, L.T, want_uv, R);
    }
    if (F & VKF_INSTANCE) {
        if (L.best_inst >= 0) {
            const DInstance &I = S.instances[L.best_inst];
            for (int32_t l = (int32_t)I.depth; l >= 0; l--) {
                const DInstance &J = S.instances[I.chain[l]];
                for (int32_t kk = (int32_t)J.n_ops - 1; kk >= 0; kk--) {
                    unapply_op(J.ops[kk], R.p, R.n);
                    // direction of the ray AFTER op (l,kk), re-derived from the world direction
                    V3 oo = L.wo, dd = L.wd;
                    for (int32_t l2 = 0; l2 <= l; l2++) {
                        const DInstance &J2 = S.instances[I.chain[l2]];
                        int32_t last = (l2 == l) ? kk : (int32_t)J2.n_ops - 1;
                        for (int32_t k2 = 0; k2 <= last; k2++) apply_op(J2.ops[k2], oo, dd);
                    }
                    face(dd, R.n, R.n, R.front);   // set_face_normal(&moved_r / &rotated_r, normal)
                }
                if (J.flip) R.front = !R.front;
            }
        }
    }
}

// ------------------------------------------------------------------ textures (material.rs:228-434)
VK_HD float perlin_noise(const DPerlin &P, V3 p) {  // material.rs:392-413 + perlin_interp 331-352
    float u = p.x - floorf(p.x), v = p.y - floorf(p.y), w = p.z - floorf(p.z);
    uint32_t i = vk::usize_low8(floorf(p.x)), j = vk::usize_low8(floorf(p.y)), k = vk::usize_low8(floorf(p.z));
    float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
    float accum = 0.0f;
    for (uint32_t di = 0; di < 2; di++)
        for (uint32_t dj = 0; dj < 2; dj++)
            for (uint32_t dk = 0; dk < 2; dk++) {
                uint32_t h = (uint32_t)P.perm_x[(i + di) & 255u] ^ (uint32_t)P.perm_y[(j + dj) & 255u] ^
                             (uint32_t)P.perm_z[(k + dk) & 255u];
                V3 c = ld3(P.ranvec[h]);
                float fi = (float)di, fj = (float)dj, fk = (float)dk;
                V3 wv = v3(u - fi, v - fj, w - fk);
                accum += (fi * uu + (1.0f - fi) * (1.0f - uu)) * (fj * vv + (1.0f - fj) * (1.0f - vv)) *
                         (fk * ww + (1.0f - fk) * (1.0f - ww)) * dot(c, wv);
            }
    return accum;
}
VK_HD float perlin_turb(const DPerlin &P, V3 p, int depth) {  // material.rs:379-390
    float accum = 0.0f, weight = 1.0f;
    V3 tp = p;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(P, tp);
        weight *= 0.5f;
        tp = tp * 2.0f;
    }
    return fabsf(accum);
}
// Non-solid textures (checker / image / Perlin noise).  Out of line on the device: it is
// reached at most once per bounce, and inlining its 7x8-corner noise and f64 trigonometry
// into the megakernel is what pushes the whole kernel into spilling.
// A turbulence value worked out ahead of the material code (vk_kernels.h cooperative_turb: the seven octaves of ONE lane's
// perlin_turb spread over seven lanes of the wave): valid for texture `tex` at the point (px, py, pz) only; tex = ~0 = none.
struct PreTurb { uint32_t tex; float val, px, py, pz; };
VK_HD PreTurb no_pre_turb() { PreTurb t; t.tex = 0xFFFFFFFFu; t.val = 0.0f; t.px = 0.0f; t.py = 0.0f; t.pz = 0.0f; return t; }

VK_COLD V3 texture_value(const DTexture *textures, const DImage *images, const uint8_t *image_bytes, const DPerlin *perlins,
                         uint32_t tex, float u, float v, float px, float py, float pz,
                         uint32_t pre_tex, float pre_val, float ppx, float ppy, float ppz) {
    V3 p = v3(px, py, pz);
    for (int guard = 0; guard < 16; guard++) {
        const DTexture &t = textures[tex];
        if (t.kind == VK_TEX_SOLID) return v3(t.r, t.g, t.b);
        if (t.kind == VK_TEX_CHECKER) {  // material.rs:250-258
            float sins = vk::sinf_(10.0f * p.x) * vk::sinf_(10.0f * p.y) * vk::sinf_(10.0f * p.z);
            tex = sins < 0.0f ? t.a : t.b_;
            continue;
        }
        if (t.kind == VK_TEX_IMAGE) {  // material.rs:283-303
            const DImage &im = images[t.a];
            float uc = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
            float vc0 = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            float vc = 1.0f - vc0;
            uint32_t i = vk::sat_u32(uc * (float)im.width), j = vk::sat_u32(vc * (float)im.height);
            if (i >= im.width) i = im.width - 1;
            if (j >= im.height) j = im.height - 1;
            const uint8_t *pix = image_bytes + im.offset + ((uint64_t)j * im.width + i) * 3u;
            float cs = 1.0f / 255.0f;
            return v3(cs * (float)pix[0], cs * (float)pix[1], cs * (float)pix[2]);
        }
        // VK_TEX_NOISE, material.rs:430-434: Vec3::new_const(1.0) * 0.5 * (1.0 + sin(..))
        const DPerlin &P = perlins[t.a];
        // the value computed ahead is for this texture at this point
        const bool pre = tex == pre_tex && px == ppx && py == ppy && pz == ppz;
        float turb = pre_val;
        if (!pre) turb = perlin_turb(P, p, 7);
        float s = 1.0f + vk::sinf_(t.scale * p.z + 10.0f * turb);
        return (v3s(1.0f) * 0.5f) * s;
    }
    return v3s(0.0f);
}
template <uint32_t F>
VK_HD V3 material_color(const DScene &S, const DMaterial &m, const Rec &R, const PreTurb &pt) {
    if (!(F & VKF_TEXTURES) || m.tex_kind == VK_TEX_SOLID) return v3(m.r, m.g, m.b);
    return texture_value(S.textures, S.images, S.image_bytes, S.perlins, m.tex, R.u, R.v, R.p.x, R.p.y, R.p.z, pt.tex, pt.val, pt.px,
        pt.py, pt.pz);
}

// ------------------------------------------------------------------ samplers (util.rs:31-63, material.rs:51-58)
VK_HD V3 random_in_unit_sphere(Rng &g) {
    for (;;) {
        float x = vk::gen_pm1(g), y = vk::gen_pm1(g), z = vk::gen_pm1(g);      // gen_range(-1, 1)
        V3 p = v3(x, y, z);
        if (length2(p) >= 1.0f) continue;
        return p;
    }
}
// A point of random_in_unit_sphere computed ahead by the wave (vk_kernels.h cooperative_ball: the generator is counter-based, so the
// candidates of one lane's rejection loop can be drawn by other lanes).  `skip` draws of the lane's stream are accounted for; `have`:
// they end with the accepted point `p` (nothing computed ahead: skip = 0, have = 0, and the loop runs here).
struct PreBall { uint32_t skip, have; V3 p; };
VK_HD PreBall no_pre_ball() { PreBall b; b.skip = 0u; b.have = 0u; b.p = v3s(0.0f); return b; }
VK_HD V3 ball_sample(Rng &g, const PreBall &pb) {
    g.ctr += pb.skip;
    if (pb.have) return pb.p;
    return random_in_unit_sphere(g);
}
// one candidate of that loop, from draws ctr+1 .. ctr+3 of the stream `key`
VK_HD V3 ball_candidate(uint64_t key, uint32_t ctr, bool &inside) {
    Rng g; g.key = key; g.ctr = ctr;
    float x = vk::gen_pm1(g), y = vk::gen_pm1(g), z = vk::gen_pm1(g);
    V3 p = v3(x, y, z);
    inside = !(length2(p) >= 1.0f);
    return p;
}
VK_HD V3 random_in_unit_disk(Rng &g) {
    for (;;) {
        float x = vk::gen_pm1(g), y = vk::gen_pm1(g);
        V3 p = v3(x, y, 0.0f);
        if (length2(p) >= 1.0f) continue;
        return p;
    }
}
VK_HD V3 random_cosine_direction(Rng &g) {
    float r1 = vk::gen_f32(g), r2 = vk::gen_f32(g);
    float z = sqrtf(1.0f - r2);
    float phi = 2.0f * r1 * PI_F;
    vk::SinCos sc = vk::sincosf_(phi);
    float x = sc.c * sqrtf(r2);
    float y = sc.s * sqrtf(r2);
    return v3(x, y, z);
}
VK_HD V3 lambertian_random(Rng &g) {
    float a = vk::gen_0_to(g, 2.0f * PI_F);                   // gen_range(0, 2 pi)
    float z = vk::gen_pm1(g);                                 // gen_range(-1, 1)
    float r = sqrtf(1.0f - z * z);
    vk::SinCos sc = vk::sincosf_(a);
    return v3(r * sc.c, r * sc.s, z);
}
VK_HD V3 reflect(V3 v, V3 n) { return v - n * dot(v, n) * 2.0f; }  // util.rs:14-16
VK_HD V3 refract(V3 uv, V3 n, float eta) {                          // util.rs:18-23
    float cos_theta = -dot(uv, n);
    V3 par = (uv + n * cos_theta) * eta;
    V3 perp = n * -sqrtf(1.0f - length2(par));
    return par + perp;
}
VK_HD float schlick(float cosine, float ref_idx) {                   // util.rs:25-29
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * vk::pow5f_(1.0f - cosine);
}
struct Onb { V3 u, v, w; };
VK_HD Onb onb_from_w(V3 n) {                                         // util.rs:99-110
    Onb o;
    o.w = unit(n);
    V3 a = fabsf(o.w.x) > 0.9f ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
    o.v = unit(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
}
VK_HD V3 onb_local(const Onb &o, V3 a) { return o.u * a.x + o.v * a.y + o.w * a.z; }

// ------------------------------------------------------------------ light sampling (hittable.rs pdf_value / random)
// pdf_value of an un-flipped Rect / Sphere
template <class Mem>
VK_HD float leaf_pdf_value(const DScene &S, const Mem &M, uint32_t ref, V3 o, V3 v) {
    uint32_t k = VKD_KIND(ref), idx = VKD_INDEX(ref);
    float t;
    if (k == DK_RECT) {                            // hittable.rs:271-282
        const DRect &q = S.rects[idx];
        if (rect_t(q, o, v, T_MIN, INFINITY, t)) {
            uint32_t a2 = (q.axes >> 4) & 3u;
            V3 outward = v3(a2 == 0 ? 1.0f : 0.0f, a2 == 1 ? 1.0f : 0.0f, a2 == 2 ? 1.0f : 0.0f), n; bool fr;
            face(v, outward, n, fr);
            float area = (q.c1 - q.c0) * (q.d1 - q.d0);
            float distance_squared = t * t * length2(v);
            float cosine = fabsf(dot(v, n)) / sqrtf(length2(v));
            return distance_squared / (cosine * area);
        }
        return 0.0f;
    }
    if (k == DK_SPHERE) {                          // hittable.rs:104-113
        DSphere s = M.sphere(idx);
        if (sphere_t(s.cx, s.cy, s.cz, s.r, o, v, length2(v), T_MIN, INFINITY, t)) {
            float cos_theta_max = sqrtf(1.0f - s.r * s.r / length2(v3(s.cx, s.cy, s.cz) - o));
            float solid_angle = 2.0f * PI_F * (1.0f - cos_theta_max);
            return 1.0f / solid_angle;
        }
        return 0.0f;
    }
    return 0.0f;                                   // trait default, hittable.rs:36-38
}
template <class Mem>
VK_HD float object_pdf_value(const DScene &S, const Mem &M, uint32_t ref, V3 o, V3 v) {
    if (ref & DREF_FLIP) return 0.0f;              // FlipFace does not forward pdf_value (trait default)
    if (VKD_KIND(ref) == DK_LIST) {                // Boxy::pdf_value -> Vec::pdf_value, hittable.rs:371-373,420-427
        DList l = S.lists[VKD_INDEX(ref)];
        float weight = 1.0f / (float)l.count;
        float sum = 0.0f;
        for (uint32_t j = 0; j < l.count; j++) {
            uint32_t r = S.list_refs[l.first + j];
            float pv = (r & DREF_FLIP) ? 0.0f : leaf_pdf_value(S, M, r, o, v);
            sum += weight * pv;
        }
        return sum;
    }
    return leaf_pdf_value(S, M, ref, o, v);
}
VK_HD V3 random_to_sphere(Rng &g, float radius, float distance_squared) {  // hittable.rs:123-134 (sic: 1-z*z)
    float r1 = vk::gen_f32(g), r2 = vk::gen_f32(g);
    float z = 1.0f + r2 * (sqrtf(1.0f - radius * radius / distance_squared) - 1.0f);
    float phi = 2.0f * PI_F * r1;
    vk::SinCos sc = vk::sincosf_(phi);
    float x = sc.c * (1.0f - z * z);
    float y = sc.s * (1.0f - z * z);
    return v3(x, y, z);
}
template <class Mem>
VK_HD V3 object_random(const DScene &S, const Mem &M, Rng &g, uint32_t ref, V3 o) {
    if (!(ref & DREF_FLIP)) {
        uint32_t k = VKD_KIND(ref), idx = VKD_INDEX(ref);
        if (k == DK_LIST) {                        // Vec::random, hittable.rs:429-433
            DList l = S.lists[idx];
            if (l.count == 0) return v3(1.0f, 0.0f, 0.0f);
            ref = S.list_refs[l.first + vk::gen_index(g, l.count)];
            if (ref & DREF_FLIP) return v3(1.0f, 0.0f, 0.0f);
            k = VKD_KIND(ref); idx = VKD_INDEX(ref);
        }
        if (k == DK_RECT) {                        // hittable.rs:284-292
            const DRect &q = S.rects[idx];
            uint32_t a0 = q.axes & 3u, a1 = (q.axes >> 2) & 3u;
            float ra = vk::gen_range(g, q.c0, q.c1);
            float rb = vk::gen_range(g, q.d0, q.d1);
            float px = a0 == 0 ? ra : (a1 == 0 ? rb : q.k);
            float py = a0 == 1 ? ra : (a1 == 1 ? rb : q.k);
            float pz = a0 == 2 ? ra : (a1 == 2 ? rb : q.k);
            return v3(px, py, pz) - o;
        }
        if (k == DK_SPHERE) {                      // hittable.rs:115-120
            DSphere s = M.sphere(idx);
            V3 direction = v3(s.cx, s.cy, s.cz) - o;
            float distance_squared = length2(direction);
            Onb uvw = onb_from_w(direction);
            return onb_local(uvw, random_to_sphere(g, s.r, distance_squared));
        }
    }
    return v3(1.0f, 0.0f, 0.0f);                   // trait default, hittable.rs:39-41
}

// ------------------------------------------------------------------ camera + sample start (main.rs:111-120,187-189)
// The new ray is returned, not installed: the kernel installs the rays of refilled lanes and of lanes whose
// path continues with ONE begin_segment (its three exact reciprocals are ~60 instructions per call site).
VK_HD void start_sample_core(Lane &L, const RenderConsts &C, uint32_t x, uint32_t y, uint32_t sample, V3 &o, V3 &d, float &time) {
    uint32_t pixel = y * C.width + x;                         // main.rs:182-183
    L.pixel = pixel; L.sample = sample;
    L.rng = vk::rng_for_sample(C.seed, pixel, sample);
    float u = ((float)x + vk::gen_f32(L.rng)) / (float)(C.width - 1);
    float v = ((float)y + vk::gen_f32(L.rng)) / (float)(C.height - 1);
    V3 rd = random_in_unit_disk(L.rng) * C.cam.lens_radius;
    V3 offset = ld3(C.cam.u) * rd.x + ld3(C.cam.v) * rd.y;
    V3 org = ld3(C.cam.origin);
    o = org + offset;
    d = ld3(C.cam.lower_left_corner) + ld3(C.cam.horizontal) * u + ld3(C.cam.vertical) * v - org - offset;
    time = vk::gen_range(L.rng, C.cam.time0, C.cam.time1);
    L.thr = v3s(1.0f); L.acc = v3s(0.0f); L.depth = 1;
}
template <uint32_t F = VKF_ALL_SCENE, class Mem = GlobalMem>
VK_HD void start_sample(Lane &L, const DScene &S, const RenderConsts &C, uint32_t x, uint32_t y, uint32_t sample) {
    V3 o, d; float time;
    start_sample_core(L, C, x, y, sample, o, d, time);
    begin_segment<Mem::ISHIFT, fused_box<F, Mem>(), spheres_only<F>()>(L, S, o, d, time);
}

VK_HD V3 background_of(const RenderConsts &C, V3 ud) {       // ud = unit(ray direction), only read for the sky
    if (C.background == VK_BACKGROUND_SKY) {
        float t = 0.5f * (ud.y + 1.0f);
        return v3s(1.0f) * (1.0f - t) + v3(0.5f, 0.7f, 1.0f) * t;
    }
    return v3(C.bg[0], C.bg[1], C.bg[2]);
}

// Called when traversal of the current segment has finished.  Returns true when the path continues with
// the ray (no, nd, ntime) (the caller installs it with begin_segment), false when the path ended (L.acc is
// its radiance).
template <uint32_t F, class Mem>
VK_HD bool shade_core(Lane &L, const DScene &S, const Mem &M, const RenderConsts &C, V3 &no, V3 &ndir, float &ntime,
    const PreTurb &pt = no_pre_turb(), const PreBall &pb = no_pre_ball()) {
    const bool miss = L.best_prim == 0;
    Rec R;
    const DMaterial *m = S.materials;
    uint32_t k0 = VK_MAT_LAMBERTIAN;
    if (!miss) {
        build_record<F, Mem>(L, S, M, R);
        // sphere-only variants: the material record by SPHERE index (DScene::sphere_material), one gather instead of two dependent ones
        // (sphere -> material index -> record): the 1 M-sphere scene +1.5 %, C2 +-0
        if constexpr ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) m = &S.sphere_material[VKD_INDEX(L.best_prim)];
        else m = &S.materials[R.mat];
        k0 = m->kind;
    }
    V3 rd = L.wd;                                             // `r` of ray_color is the world-space ray
    // unit_vector(r.direction) is what the sky, Metal and Dielectric start from (main.rs IOW sky; material.rs:119,
    // 155,182): one copy of its three divisions + square root for the whole wave instead of one per branch
    V3 ud = rd;
    {
        if (F & VKF_SPEC_DIFFUSE) k0 = (k0 == VK_MAT_SPEC_DIFFUSE) ? (uint32_t)VK_MAT_METAL : k0;   // its children may need it
        bool need_ud = miss ? (C.background == VK_BACKGROUND_SKY) : (k0 == VK_MAT_METAL || k0 == VK_MAT_DIELECTRIC);
        if (need_ud) ud = unit(rd);
    }
    if (miss) {                                               // miss: main.rs:150-152
        L.acc = L.acc + L.thr * background_of(C, ud);
        return false;
    }
    ntime = L.time;
    if (!(F & VKF_INTEG_PDF)) {
        // emitted + attenuation * ray_color(scattered): Material::emitted + Material::scatter
        V3 emitted = v3s(0.0f);
        V3 atten;
        bool scattered = true;
        uint32_t kind = m->kind;
        if (kind == VK_MAT_LAMBERTIAN) {                      // material.rs:85-90
            ndir = R.n + lambertian_random(L.rng);
            atten = material_color<F>(S, *m, R, pt);
        } else if (kind == VK_MAT_METAL) {                    // material.rs:118-132
            V3 reflected = reflect(ud, R.n);
            ndir = reflected + ball_sample(L.rng, pb) * m->param;      // (no draw before it: see cooperative_ball)
            atten = material_color<F>(S, *m, R, pt);
            scattered = dot(ndir, R.n) > 0.0f;
        } else if (kind == VK_MAT_DIELECTRIC) {               // material.rs:150-175
            atten = v3s(1.0f);
            float eta = R.front ? 1.0f / m->param : m->param;
            float cos_theta = fminf(dot(-ud, R.n), 1.0f);
            float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            if (eta * sin_theta > 1.0f) ndir = reflect(ud, R.n);
            else {
                float reflect_prob = schlick(cos_theta, eta);
                if (vk::gen_f32(L.rng) < reflect_prob) ndir = reflect(ud, R.n);
                else ndir = refract(ud, R.n, eta);
            }
        } else if (kind == VK_MAT_ISOTROPIC) {                // material.rs:442-446
            ndir = ball_sample(L.rng, pb);
            atten = material_color<F>(S, *m, R, pt);
        } else {                                              // DiffuseLight: material.rs:215-225
            scattered = false;
            if (kind == VK_MAT_DIFFUSE_LIGHT && R.front) emitted = material_color<F>(S, *m, R, pt);
        }
        L.acc = L.acc + L.thr * emitted;
        if (!scattered) return false;
        L.thr = L.thr * atten;
    } else {
        // HEAD integrator, main.rs:131-149
        V3 emitted = v3s(0.0f);
        if (m->kind == VK_MAT_DIFFUSE_LIGHT) {
            if (R.front) emitted = material_color<F>(S, *m, R, pt);
            L.acc = L.acc + L.thr * emitted;                  // scatter_with_pdf is None: return emitted
            return false;
        }
        const DMaterial *outer = m;
        if (F & VKF_SPEC_DIFFUSE) {
            for (int guard = 0; guard < 8 && m->kind == VK_MAT_SPEC_DIFFUSE; guard++) {   // material.rs:475-483
                uint32_t pick = vk::gen_f32(L.rng) < m->param ? (m->ab & 0xFFFFu) : (m->ab >> 16);
                m = &S.materials[pick];
            }
            if (m->kind == VK_MAT_DIFFUSE_LIGHT) { L.acc = L.acc + L.thr * emitted; return false; }
        }
        uint32_t kind = m->kind;
        if (kind == VK_MAT_METAL) {                           // material.rs:134-141 (Ray::new: time 0, never absorbs)
            V3 reflected = reflect(ud, R.n);
            // (computed ahead only where the hit's own material is the Metal: no SpecDiffuse draw before it)
            ndir = reflected + ball_sample(L.rng, pb) * m->param;
            ntime = 0.0f;
            L.thr = L.thr * material_color<F>(S, *m, R, pt);         // specular: emitted is NOT added (main.rs:134-137)
        } else if (kind == VK_MAT_DIELECTRIC) {               // material.rs:177-206
            float eta = R.front ? 1.0f / m->param : m->param;
            float cos_theta = fminf(dot(-ud, R.n), 1.0f);
            float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            if (eta * sin_theta > 1.0f) ndir = reflect(ud, R.n);
            else {
                float reflect_prob = schlick(cos_theta, eta);
                if (vk::gen_f32(L.rng) < reflect_prob) ndir = reflect(ud, R.n);
                else ndir = refract(ud, R.n, eta);
            }
        } else {                                              // Lambertian / Isotropic: material.rs:92-98,448-454
            V3 atten = material_color<F>(S, *m, R, pt);
            Onb uvw = onb_from_w(R.n);                        // CosinePDF::new(rec.normal)
            // MixturePDF::generate, util.rs:177-185
            if (vk::gen_f32(L.rng) < 0.5f) {
                // HittablePDF::generate -> lights.random(o): Vec::random, hittable.rs:429-433
                uint32_t li = vk::gen_index(L.rng, S.n_lights);
                ndir = object_random(S, M, L.rng, S.lights[li], R.p);
            } else {
                ndir = onb_local(uvw, random_cosine_direction(L.rng));
            }
            // MixturePDF::value, util.rs:173-175
            float weight = 1.0f / (float)S.n_lights;          // Vec::pdf_value, hittable.rs:420-427
            float lsum = 0.0f;
            for (uint32_t j = 0; j < S.n_lights; j++) lsum += weight * object_pdf_value(S, M, S.lights[j], R.p, ndir);
            float cosv = dot(unit(ndir), uvw.w);              // CosinePDF::value, util.rs:134-142
            float cpdf = cosv <= 0.0f ? 0.0f : cosv / PI_F;
            float pdf = 0.5f * lsum + 0.5f * cpdf;
            // scattering_pdf of the OUTER material (main.rs:145; SpecDiffuse forwards to its diffuse child)
            const DMaterial *sm = outer;
            if (F & VKF_SPEC_DIFFUSE) {
                for (int guard = 0; guard < 8 && sm->kind == VK_MAT_SPEC_DIFFUSE; guard++) sm = &S.materials[sm->ab >> 16];
            }
            float spdf = 0.0f;
            if (sm->kind == VK_MAT_LAMBERTIAN || sm->kind == VK_MAT_ISOTROPIC) {   // material.rs:100-108,456-464
                float cs = dot(R.n, unit(ndir));
                spdf = cs < 0.0f ? 0.0f : cs / PI_F;
            }
            L.acc = L.acc + L.thr * emitted;
            L.thr = (L.thr * (atten * spdf)) / pdf;
        }
    }
    L.depth += 1;
    if (L.depth > C.max_depth) {                              // main.rs:126-128: the next call returns 0
        L.acc = L.acc + L.thr * v3s(0.0f);                    // keeps the reference's 0*inf / 0/0 -> NaN drops
        return false;
    }
    no = R.p;
    return true;
}
template <uint32_t F, class Mem>
VK_HD bool shade(Lane &L, const DScene &S, const Mem &M, const RenderConsts &C) {
    V3 o, d; float time;
    if (!shade_core<F, Mem>(L, S, M, C, o, d, time)) return false;
    begin_segment<Mem::ISHIFT, fused_box<F, Mem>(), spheres_only<F>()>(L, S, o, d, time);
    return true;
}

}  // namespace vkd
#endif
