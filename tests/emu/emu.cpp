// emu.cpp — TEST TOOL: runs the megakernel's per-lane logic (vecchio_amd/csrc/vk_trace.h) and
// the lineariser on the host, one sample at a time, so the iterative/deferred/threaded
// formulation can be compared with the recursive oracle WITHOUT a GPU.  It is built only
// under tests/, is not part of libvecchio_amd.so and is never reachable from the C ABI.
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <atomic>
#include <vector>
#include <type_traits>

#include "../../vecchio_amd/csrc/vk_linearize.h"
#include "../../vecchio_amd/csrc/vk_trace.h"

#include <cstdlib>
#include <cstdio>

using namespace vkd;

static thread_local std::string g_err;
static std::atomic<uint64_t> g_redo(0), g_segments(0);      // exact re-treeing: segments walked twice / all segments
// what the device's walk visits (bench.py: the algorithmic bytes of the walk performed, next to the oracle's on the tree handed over):
// box tests (one per item stepped over) and sphere tests of scenes of spheres only, second walks and samples rendered again included
static std::atomic<uint64_t> g_box_tests(0), g_sphere_tests(0);

// VK_RETREE=0/1 forces re-treeing off / on (same switch as the device library); default: vk_scene_desc.flags
static int linearize_env(const vk_scene_desc *desc, LinearScene &LS, std::string &err) {
    LinearizeOptions opt;
    if (const char *e = getenv("VK_RETREE")) opt.retree = (e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
    if (const char *e = getenv("VK_GATE_GROW")) opt.gate_grow = e[0] != '0';
    if (const char *e = getenv("VK_T_PAD")) opt.t_pad = (float)atof(e);
    if (const char *e = getenv("VK_GATE_PROOF")) opt.want_proof = e[0] != '0';      // (as the library: prefers, does not allow)
    if (const char *e = getenv("VK_EMPIRICAL_TREES")) opt.allow_empirical = e[0] == '1';
    if (const char *e = getenv("VK_NEAR_FORM")) opt.near_form = e[0] != '0';
    if (const char *e = getenv("VK_UNIT_FORM")) opt.unit_form = e[0] != '0';
    if (const char *e = getenv("VK_NEAR_FIRST")) opt.near_first = e[0] != '0';
    if (const char *e = getenv("VK_GRID_FORM")) opt.grid_form = e[0] != '0';
    return linearize(desc, LS, err, opt);
}
// EMU_GLOBAL_VARIANT=1: scenes of spheres only as the device runs them from GLOBAL memory (unfused box test; exact re-treeing with both
// trees in one item array, early segments walked again in place) instead of as it runs them from LDS (fused test, queued samples)
static bool global_variant() { const char *e = getenv("EMU_GLOBAL_VARIANT"); return e && e[0] == '1'; }
static DScene scene_view(const LinearScene &LS, std::vector<DItem> &both) {
    DScene S = LS.host_view();
    // EMU_GRID=0: the rebuilt TREE of a world that has a grid too (what the device does with such a world when it is too large for LDS)
    if (const char *e = getenv("EMU_GRID")) { if (e[0] == '0') drop_grid(S); }
    if (S.grid.nu != 0u && !LS.ref_items.empty()) {
        use_grid(S);
        // (test switch: no dilation — the walk then visits the cells on the ray's exact path only, and part F's control shows what is lost)
        if (const char *e = getenv("EMU_GRID_NO_DILATION")) { if (e[0] == '1') { S.grid.k = 0.0f; S.grid.slack = 0.0f; } }
        // the grid form: a failed segment requeues its sample (trace_one: the whole sample again on reference_view), whether the device
        // walks the grid from LDS or from global memory
        return S;
    }
    // (the near form whose reach does not span its small spheres: both trees in items[], as the device keeps it)
    if ((global_variant() || (LS.near_form && !LS.near_spans)) && !LS.ref_items.empty()) {
        uint32_t ws = 0;
        both = LS.combined_items(ws);
        S.items = both.data(); S.n_items = (uint32_t)both.size(); S.n_world_items = (uint32_t)both.size(); S.walk_start = ws;
        S.unit_tree = both.data();      // (the tree as handed over comes first, item for item)
        S.ref_items = nullptr; S.n_ref_items = 0;
        if (const char *e = getenv("EMU_PRIMARY_REF")) S.primary_ref = e[0] == '1';       // (what vk_api.hip decides per frame)
    }
    return S;
}

// the same records as GlobalMem, with the fused box test the device runs on LDS-resident scenes (vk_trace.h set_space)
struct FusedMem : GlobalMem { static constexpr bool FUSED_BOX = true; };

// the scene as handed over: what a sample dropped by exact re-treeing is rendered on (vk_api.hip builds the same view for the
// second launch)
static DScene reference_view(const DScene &S) {
    DScene r = S;
    r.items = S.ref_items; r.n_items = S.n_ref_items; r.n_world_items = S.n_ref_items;
    r.ref_items = nullptr; r.n_ref_items = 0; r.t_pad = 0.0f; r.gate_scale = 1.0f; r.tmin_gate = T_MIN; r.tie_rank = nullptr;
    r.grid.nu = 0u;
    return r;
}

template <uint32_t F, class Mem = GlobalMem>
static void trace_one(const DScene &S, const GlobalMem &M0, const RenderConsts &C, uint32_t pixel, uint32_t sample, V3 &rgb,
    uint32_t &draws, uint64_t *steps) {
    Mem M; static_cast<GlobalMem &>(M) = M0;
    Lane L;
    start_sample<F, Mem>(L, S, C, pixel % C.width, pixel / C.width, sample);
    // the segment just walked was walked on the tree as handed over (a primary ray under DScene::primary_ref starts there: begin_segment)
    bool on_ref = S.primary_ref != 0u && S.walk_start != 0u && spheres_only<F>();
    for (;;) {
        uint64_t nb = 0, ns = 0;
        while (traversing(L)) {
            if (has_prim_work(L)) ns += ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u && L.pend2) ? 2u : 1u;
            else if (L.i < L.end) nb++;       // (a step at the end of an instance's range reads no item: it leaves the instance)
            traverse_step<F, Mem>(L, S, M);
            if (steps) (*steps)++;
        }
        g_box_tests += nb; g_sphere_tests += ns;
        const bool early = !on_ref && segment_unsafe<F, Mem>(L, S, M);
        if (early && getenv("EMU_REDO_REASONS")) {      // (diagnostics: why the segment does not stand)
            const float dn = sqrtf(L.a);
            const bool outside = origin_untrusted(S, L.o), odd = !(L.xnan == L.xnan), far = S.reach > 0.0f && !(L.T * dn <= S.reach);
            fprintf(stderr, "REDO depth %u outside %d odd %d beyond_reach %d miss %d T|d| %.4g prim %08x o (%.4g %.4g %.4g) u (%.3g %.3g %.3g)\n", L.depth, (int)outside,
                (int)odd, (int)far, (int)(L.best_prim == 0u), (double)(L.T * dn), L.best_prim, L.o.x, L.o.y, L.o.z, L.d.x / dn, L.d.y / dn, L.d.z / dn);
        }
        if (getenv("EMU_TRACE")) fprintf(stderr, "  seg depth %u o (%.9g %.9g %.9g) d (%.9g %.9g %.9g) T %.9g prim %08x early %d on_ref %d t_pad %g\n", L.depth,
            L.o.x, L.o.y, L.o.z, L.d.x, L.d.y, L.d.z, L.T, L.best_prim, (int)early, (int)on_ref, S.t_pad);
        on_ref = false;
        if (S.walk_start != 0u && early) {                      // exact re-treeing, both trees in items[] (the device's global-memory
            g_redo++;                                           // scenes): this segment again, on the tree as handed over
            begin_segment<Mem::ISHIFT, fused_box<F, Mem>(), spheres_only<F>()>(L, S, L.o, L.d, L.time, true);
            on_ref = true;
            continue;
        }
        if (early) {                       // exact re-treeing: the whole sample again on the tree as handed over (the device's LDS scenes)
            DScene Sr = reference_view(S);
            GlobalMem Mr = M0; Mr.items = Sr.items;
            g_redo++;
            trace_one<F, Mem>(Sr, Mr, C, pixel, sample, rgb, draws, steps);
            return;
        }
        g_segments++;
        if (!shade<F, Mem>(L, S, M, C)) break;
    }
    rgb = L.acc;
    draws = L.rng.ctr;
}

static RenderConsts make_consts(const vk_camera *cam, const vk_render_params *p) {
    RenderConsts C;
    C.cam = *cam;
    C.width = p->width; C.height = p->height; C.spp = p->samples_per_pixel; C.max_depth = p->max_depth;
    C.seed = p->seed; C.integrator = p->integrator; C.background = p->background;
    C.bg[0] = p->background_color[0]; C.bg[1] = p->background_color[1]; C.bg[2] = p->background_color[2];
    return C;
}

// AxisBB::hit property test: the kernel's box step (fast path + margin, exact fallback) against the reference's division
// sequence alone, for n (box, ray, tmax) triples.  fused = 0: the (b - o) * (1/d) form of the general variants; 1: the
// fma(b, 1/d, -o/d) form of the sphere-only variants.  decisions[i] bit0 = kernel's answer, bit1 = slab_exact's answer,
// bit2 = the kernel took the exact fallback.
// perturb != 0: the three reciprocals 1/d are moved by -1, 0 or +1 ulp (pseudo-randomly per case and axis) before the step: what the
// device's v_rcp_f32 (1 ulp) may hand the fast path instead of the correctly rounded quotient (vk_trace.h set_space)
template <uint32_t F, bool FUSED>
static void box_decisions(const float *boxes, const float *rays, size_t n, uint8_t *decisions, uint32_t perturb = 0) {
    for (size_t k = 0; k < n; k++) {
        DItem it;
        it.mnx = boxes[k * 6 + 0]; it.mxx = boxes[k * 6 + 1]; it.mny = boxes[k * 6 + 2]; it.mxy = boxes[k * 6 + 3];
        it.mnz = boxes[k * 6 + 4]; it.mxz = boxes[k * 6 + 5];
        it.w0 = ((uint32_t)DK_SPHERE << 28) | 1u; it.w1 = 0;      // a leaf: pend != 0 afterwards <=> box hit
        Lane L;
        memset(&L, 0, sizeof(L));
        V3 o = v3(rays[k * 7 + 0], rays[k * 7 + 1], rays[k * 7 + 2]), d = v3(rays[k * 7 + 3], rays[k * 7 + 4], rays[k * 7 + 5]);
        set_space<FUSED>(L, o, d);
        if (perturb) {
            uint64_t h = vk::mix64((uint64_t)perturb * 0x9E3779B97F4A7C15ull + k);
            float *iv[3] = {&L.inv.x, &L.inv.y, &L.inv.z};
            for (int a = 0; a < 3; a++) {
                int step = (int)((h >> (8 * a)) % 3u) - 1;
                uint32_t b = vk::f32_bits(*iv[a]);
                if (std::isfinite(*iv[a]) && *iv[a] != 0.0f) *iv[a] = vk::bits_f32(b + (uint32_t)step);     // +-1 ulp in magnitude
            }
            if (FUSED && !std::isnan(L.xnan)) {      // o/d and the margin's |o/d| term follow the reciprocal actually used
                L.oi = v3(o.x * L.inv.x, o.y * L.inv.y, o.z * L.inv.z);
                L.xnan = fmaxf(fmaxf(fabsf(L.oi.x), fabsf(L.oi.y)), fabsf(L.oi.z)) * 4.76837158203125e-7f;
            }
        }
        L.T = rays[k * 7 + 6];
        L.i = 0; L.end = 1; L.pend = 0;
        typename std::conditional<FUSED, FusedMem, GlobalMem>::type M;
        M.items = &it; M.spheres = nullptr; M.sphere_mat = nullptr; M.boxes = nullptr;
        DScene S0; memset(&S0, 0, sizeof(S0)); S0.gate_scale = 1.0f; S0.tmin_gate = T_MIN;
        box_step_core<F, decltype(M)>(L, S0, M);
        bool fast = L.pend != 0;
        bool exact = slab_exact(it, o, d, T_MIN, rays[k * 7 + 6]);
        // was the fallback taken?  the margin test exactly as box_step_core writes it
        float x0, x1, y0, y1, z0, z1;
        if (FUSED) {
            x0 = __builtin_fmaf(it.mnx, L.inv.x, -L.oi.x); x1 = __builtin_fmaf(it.mxx, L.inv.x, -L.oi.x);
            y0 = __builtin_fmaf(it.mny, L.inv.y, -L.oi.y); y1 = __builtin_fmaf(it.mxy, L.inv.y, -L.oi.y);
            z0 = __builtin_fmaf(it.mnz, L.inv.z, -L.oi.z); z1 = __builtin_fmaf(it.mxz, L.inv.z, -L.oi.z);
        } else {
            x0 = (it.mnx - o.x) * L.inv.x; x1 = (it.mxx - o.x) * L.inv.x;
            y0 = (it.mny - o.y) * L.inv.y; y1 = (it.mxy - o.y) * L.inv.y;
            z0 = (it.mnz - o.z) * L.inv.z; z1 = (it.mxz - o.z) * L.inv.z;
        }
        float lo = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), T_MIN));
        float hi = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), rays[k * 7 + 6]));
        float dlt = hi - lo;
        bool fb = !(fabsf(dlt) >= __builtin_fmaf(hi, 2.0e-6f, L.xnan));
        decisions[k] = (uint8_t)((fast ? 1 : 0) | (exact ? 2 : 0) | (fb ? 4 : 0));
    }
}
extern "C" {

const char *emu_last_error(void) { return g_err.c_str(); }

// exact re-treeing: samples rendered again on the tree as handed over since the last call (and segments walked)
void emu_take_redo_stats(uint64_t out[2]) { out[0] = g_redo.exchange(0); out[1] = g_segments.exchange(0); }
// box tests and (scenes of spheres only) sphere tests of every walk since the last call
void emu_take_visit_counts(uint64_t out[2]) { out[0] = g_box_tests.exchange(0); out[1] = g_sphere_tests.exchange(0); }

// the full-feature variants, and for scenes of spheres only the sphere-only ones (as the device library picks them: they
// run the fused box test, vk_trace.h set_space); the Cornell-type variants in between are the same code as the full ones
static const uint32_t FALL = VKF_ALL_SCENE;
static const uint32_t FPDF = VKF_ALL_SCENE | VKF_INTEG_PDF;
static void trace_any(const DScene &S, const GlobalMem &M, const RenderConsts &C, uint32_t integrator, uint32_t pixel, uint32_t sample,
    V3 &rgb, uint32_t &draws, uint64_t *steps) {
    bool lean = S.features == 0u && !getenv("VK_FORCE_FULL_VARIANT");
    const bool glob = lean && (global_variant() || S.walk_start != 0u);      // (both trees in items[]: the device's global-memory form)
    if (integrator == VK_INTEGRATOR_PDF) {
        if (glob) trace_one<VKF_INTEG_PDF, GlobalMem>(S, M, C, pixel, sample, rgb, draws, steps);
        else if (lean) trace_one<VKF_INTEG_PDF, FusedMem>(S, M, C, pixel, sample, rgb, draws, steps);
        else trace_one<FPDF>(S, M, C, pixel, sample, rgb, draws, steps);
    } else {
        if (glob) trace_one<0u, GlobalMem>(S, M, C, pixel, sample, rgb, draws, steps);
        else if (lean) trace_one<0u, FusedMem>(S, M, C, pixel, sample, rgb, draws, steps);
        else trace_one<FALL>(S, M, C, pixel, sample, rgb, draws, steps);
    }
}

int emu_sample(const vk_scene_desc *desc, const vk_camera *cam, const vk_render_params *p, uint32_t pixel, uint32_t sample,
               float rgb[3], uint32_t *draws) {
    LinearScene LS;
    int st = linearize_env(desc, LS, g_err);
    if (st != VK_OK) return st;
    std::vector<DItem> both;
    DScene S = scene_view(LS, both);
    GlobalMem M{S.items, S.spheres, S.sphere_mat, S.boxes};
    RenderConsts C = make_consts(cam, p);
    V3 c; uint32_t dr;
    trace_any(S, M, C, p->integrator, pixel, sample, c, dr, nullptr);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    if (draws) *draws = dr;
    return VK_OK;
}

// debug: the primary ray of (pixel, sample) and the closest hit of its first segment: out = o3, d3, T, best_prim (bits)
int emu_first_hit(const vk_scene_desc *desc, const vk_camera *cam, const vk_render_params *p, uint32_t pixel, uint32_t sample,
    float out[8]) {
    LinearScene LS;
    int st = linearize_env(desc, LS, g_err);
    if (st != VK_OK) return st;
    std::vector<DItem> both;
    DScene S = scene_view(LS, both);
    GlobalMem M{S.items, S.spheres, S.sphere_mat, S.boxes};
    RenderConsts C = make_consts(cam, p);
    Lane L;
    start_sample<FALL>(L, S, C, pixel % C.width, pixel / C.width, sample);
    out[0] = L.o.x; out[1] = L.o.y; out[2] = L.o.z; out[3] = L.d.x; out[4] = L.d.y; out[5] = L.d.z;
    while (traversing(L)) traverse_step<FALL, GlobalMem>(L, S, M);
    out[6] = L.T; memcpy(&out[7], &L.best_prim, 4);
    return VK_OK;
}

// per-sample outputs: out_rgbd[(pixel*spp + s)*4 + {0,1,2}] = radiance, [3] = draw count (as float bits of uint)
int emu_render(const vk_scene_desc *desc, const vk_camera *cam, const vk_render_params *p, float *rgb_out,
               float *per_sample_out, int n_threads, uint64_t *steps_out, uint32_t *info_out) {
    LinearScene LS;
    int st = linearize_env(desc, LS, g_err);
    if (st != VK_OK) return st;
    if (p->integrator == VK_INTEGRATOR_PDF && LS.lights.empty()) { g_err =
        "PDF integrator needs a non-empty lights list (hittable.rs:431 would panic)"; return VK_ERR_UNSUPPORTED; }
    std::vector<DItem> both;
    DScene S = scene_view(LS, both);
    GlobalMem M{S.items, S.spheres, S.sphere_mat, S.boxes};
    RenderConsts C = make_consts(cam, p);
    if (info_out) { info_out[0] = S.n_items; info_out[1] = LS.n_prims; info_out[2] = (uint32_t)LS.instances.size();
        info_out[3] = LS.features; }
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint32_t> next_row(0);
    std::atomic<uint64_t> total_steps(0);
    auto worker = [&]() {
        uint64_t steps = 0;
        for (;;) {
            uint32_t y = next_row.fetch_add(1);
            if (y >= p->height) break;
            for (uint32_t x = 0; x < p->width; x++) {
                uint32_t pix = y * p->width + x;
                V3 sum = v3s(0.0f);
                for (uint32_t s = 0; s < p->samples_per_pixel; s++) {
                    V3 c; uint32_t dr;
                    trace_any(S, M, C, p->integrator, pix, s, c, dr, &steps);
                    if (per_sample_out) {
                        float *o = per_sample_out + ((size_t)pix * p->samples_per_pixel + s) * 4;
                        o[0] = c.x; o[1] = c.y; o[2] = c.z; memcpy(&o[3], &dr, 4);
                    }
                    if (std::isfinite(c.x) && std::isfinite(c.y) && std::isfinite(c.z)) sum = sum + c;
                }
                sum = sum / (float)p->samples_per_pixel;
                rgb_out[(size_t)pix * 3 + 0] = sum.x; rgb_out[(size_t)pix * 3 + 1] = sum.y; rgb_out[(size_t)pix * 3 + 2] = sum.z;
            }
        }
        total_steps += steps;
    };
    std::vector<std::thread> ths;
    for (int t = 1; t < n_threads; t++) ths.emplace_back(worker);
    worker();
    for (auto &t : ths) t.join();
    if (steps_out) *steps_out = total_steps.load();
    if (getenv("VK_EMU_STATS")) fprintf(stderr, "emu: %llu segments, %llu samples rendered again on the reference tree\n", (unsigned long long)g_segments.load(), (unsigned long long)g_redo.load());
    return VK_OK;
}

// the box step's clamp against the loop-carried tmax (vk_trace.h min_with_tmax): out[i] = min_with_tmax(a[i], t[i])
void emu_min_with_tmax(const float *a, const float *t, size_t n, float *out) {
    for (size_t k = 0; k < n; k++) out[k] = min_with_tmax(a[k], t[k]);
}
void emu_box_decisions(const float *boxes, const float *rays, size_t n, uint8_t *decisions, int fused) {
    if (fused) box_decisions<0u, true>(boxes, rays, n, decisions);
    else box_decisions<VKF_ALL_SCENE, false>(boxes, rays, n, decisions);
}
void emu_box_decisions_rcp(const float *boxes, const float *rays, size_t n, uint8_t *decisions, int fused, uint32_t perturb) {
    if (fused) box_decisions<0u, true>(boxes, rays, n, decisions, perturb);
    else box_decisions<VKF_ALL_SCENE, false>(boxes, rays, n, decisions, perturb);
}

}  // extern "C"

// ------------------------------------------------------------------ the gate lemma of exact re-treeing, attacked
// with the kernel's own arithmetic (vk_trace.h sphere_t_tie, slab_exact) and the lineariser's own bound (vk_linearize.h rt_eta,
// rt_unit_growth).  tests/test_gate_lemma.py drives these.
namespace {
struct Lcg {        // splitmix-style stream: reproducible, no <random> distributions whose output differs between libraries
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(vk::mix64(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull)) {}
    uint64_t next() { return vk::mix64(s += 0x9E3779B97F4A7C15ull); }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double log_uni(double lo, double hi) { return std::exp(std::log(lo) + uni() * (std::log(hi) - std::log(lo))); }
};
// the candidate of Sphere::hit (hittable.rs:65-95): the root it reports when tmax = +inf
bool candidate(const float c[3], float r, V3 o, V3 d, float &t) {
    bool tie;
    return sphere_t_tie(c[0], c[1], c[2], r, o, d, length2(d), T_MIN, INFINITY, t, tie);
}
typedef long double LD;
}  // namespace

extern "C" {

// Part A.  K = |dist(H, centre) - R| * R / (u (rho + R)^2) of both roots, H = o + t d in extended precision: the largest over n
// pseudo-random configurations (near, far, inside, grazing, axis-parallel) followed by `climb` steps of bit-level hill climbing from the
// worst one.  The lemma's constant is RT_KAPPA / u = 32.
double emu_lemma_residual(uint64_t n, uint64_t seed, uint64_t climb) {
    const LD U = ldexpl(1.0L, -24);
    struct Case { float c[3], r, o[3], d[3]; };
    auto keff = [&](const Case &k) -> LD {
        V3 o = v3(k.o[0], k.o[1], k.o[2]), d = v3(k.d[0], k.d[1], k.d[2]);
        const float a = length2(d);
        V3 oc = o - v3(k.c[0], k.c[1], k.c[2]);
        const float half_b = dot(oc, d), c = length2(oc) - k.r * k.r, disc = half_b * half_b - a * c;
        if (!(disc > 0.0f)) return 0;
        const float root = sqrtf(disc);
        const float ts[2] = {(-half_b - root) / a, (-half_b + root) / a};
        const LD ocx = (LD)k.o[0] - k.c[0], ocy = (LD)k.o[1] - k.c[1], ocz = (LD)k.o[2] - k.c[2];
        const LD rho = sqrtl(ocx * ocx + ocy * ocy + ocz * ocz);
        LD best = 0;
        for (float t : ts) {
            if (!(t > T_MIN) || !std::isfinite(t)) continue;
            const LD hx = ocx + (LD)t * k.d[0], hy = ocy + (LD)t * k.d[1], hz = ocz + (LD)t * k.d[2];
            const LD q = fabsl(sqrtl(hx * hx + hy * hy + hz * hz) - k.r) * k.r / (U * (rho + k.r) * (rho + k.r));
            if (q > best) best = q;
        }
        return best;
    };
    Lcg g(seed);
    LD worst = 0; Case wc{};
    for (uint64_t it = 0; it < n; it++) {
        Case k;
        const double R = g.log_uni(1e-3, 1e5);
        const double cm = g.uni() < 0.3 ? 0.0 : g.log_uni(1e-2, 1e5);
        double cd[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
        const double cn = std::sqrt(cd[0] * cd[0] + cd[1] * cd[1] + cd[2] * cd[2]) + 1e-30;
        for (int a = 0; a < 3; a++) k.c[a] = (float)(cd[a] / cn * cm);
        k.r = (float)R;
        const int mode = (int)(g.uni() * 5);
        double m = mode == 0 ? g.log_uni(1.0 + 1e-7, 1.01) : (mode == 1 ? g.log_uni(1e-3, 1.0) : (mode == 2 ? g.log_uni(1.0, 100.0) : g.log_uni(1.0, 1e4)));
        if (mode == 4) m = 1.0;
        double od[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
        if (g.uni() < 0.2) od[(int)(g.uni() * 3) % 3] = 0.0;
        const double on = std::sqrt(od[0] * od[0] + od[1] * od[1] + od[2] * od[2]) + 1e-30;
        for (int a = 0; a < 3; a++) k.o[a] = (float)((double)k.c[a] + od[a] / on * m * R);
        const double p = g.uni() < 0.5 ? (1.0 + (g.uni() - 0.5) * g.log_uni(1e-9, 1e-1)) : g.uni() * 1.05;      // perpendicular offset / R
        double e1[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
        const double dp = (e1[0] * od[0] + e1[1] * od[1] + e1[2] * od[2]) / (on * on);
        for (int a = 0; a < 3; a++) e1[a] -= dp * od[a];
        const double en = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) + 1e-30;
        double dd[3];
        for (int a = 0; a < 3; a++) dd[a] = (double)k.c[a] + e1[a] / en * p * R - k.o[a];
        const double dn = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) + 1e-30, dl = g.log_uni(1e-3, 1e3);
        for (int a = 0; a < 3; a++) k.d[a] = (float)(dd[a] / dn * dl);
        const LD q = keff(k);
        if (q > worst) { worst = q; wc = k; }
    }
    for (uint64_t it = 0; it < climb; it++) {
        Case nx = wc;
        float *f = reinterpret_cast<float *>(&nx);
        const int idx = (int)(g.uni() * 10);
        uint32_t b; memcpy(&b, &f[idx], 4);
        int step = (int)(g.uni() * 64) - 32; if (g.uni() < 0.3) step *= 1000;
        b += (uint32_t)step; memcpy(&f[idx], &b, 4);
        if (!std::isfinite(f[idx]) || !(nx.r > 0.0f)) continue;
        const LD q = keff(nx);
        if (q > worst) { worst = q; wc = nx; }
    }
    return (double)worst;
}

// Part B.  Units of two spheres of radius 0.2 (the second one `len` away: long units are what strains the gate), their box as
// BVHNode::new computes it; rays from inside the trusted ball (centre = the first sphere's, radius ball_r0) aimed to GRAZE the first
// sphere near the top face of its box, nearly parallel to it (the configuration in which a hit can precede the box entry by the most),
// and rays aimed anywhere at the unit.  For every ray that passes the unit's box as handed over and has a candidate on the first
// sphere — i.e. that BVHNode::hit can accept — the gate must pass with tmax = next(t) (1 + pad): counts = {such rays, gates that
// failed}.  grow != 0: the gate's box is the unit's grown by rt_unit_growth (the proven form), else the unit's own (the empirical form).
// viol = the failing case with the largest relative earliness: c1(3) r1 c2(3) r2 o(3) d(3) t entry llc(3).
int emu_gate_soundness(uint64_t n, uint64_t seed, int grow, float pad, double ball_r0, uint64_t counts[2], float viol[19]) {
    Lcg g(seed);
    counts[0] = counts[1] = 0;
    double worst = 0.0;
    for (uint64_t it = 0; it < n; it++) {
        const float R = 0.2f;
        const double len = g.log_uni(0.5, 120.0);
        float c[2][3] = {{0.0f, 0.2f, 0.0f}, {(float)len, 0.2f, (float)((g.uni() - 0.5) * 2.0)}};
        float r[2] = {R, R};
        DItem U; memset(&U, 0, sizeof(U));
        U.mnx = fminf(c[0][0] - R, c[1][0] - R); U.mxx = fmaxf(c[0][0] + R, c[1][0] + R);
        U.mny = fminf(c[0][1] - R, c[1][1] - R); U.mxy = fmaxf(c[0][1] + R, c[1][1] + R);
        U.mnz = fminf(c[0][2] - R, c[1][2] - R); U.mxz = fmaxf(c[0][2] + R, c[1][2] + R);
        RtDomain dom; dom.c0[0] = c[0][0]; dom.c0[1] = c[0][1]; dom.c0[2] = c[0][2]; dom.r0 = ball_r0;
        DItem G = U;
        if (grow) {
            const float mn[3] = {U.mnx, U.mny, U.mnz}, mx[3] = {U.mxx, U.mxy, U.mxz};
            const double gd = rt_unit_growth(mn, mx, 2, c, r, dom, pad);
            if (gd < 0.0) continue;                        // (the lineariser would shrink the ball)
            const float gg = std::nextafter((float)gd, INFINITY);
            G.mnx = std::nextafter(U.mnx - gg, -INFINITY); G.mxx = std::nextafter(U.mxx + gg, INFINITY);
            G.mny = std::nextafter(U.mny - gg, -INFINITY); G.mxy = std::nextafter(U.mxy + gg, INFINITY);
            G.mnz = std::nextafter(U.mnz - gg, -INFINITY); G.mxz = std::nextafter(U.mxz + gg, INFINITY);
        }
        // the ray
        V3 o, d;
        const double rho = g.log_uni(1.0, ball_r0 * 0.999);
        if (g.uni() < 0.8) {
            // through P, just above (or below) the top of the first sphere, descending slowly along +x
            const double delta = (g.uni() < 0.5 ? 1.0 : -1.0) * g.log_uni(1e-8, 0.5) * R;
            const double P[3] = {(g.uni() - 0.5) * 0.4, 0.4 + delta, (g.uni() - 0.5) * 0.4};
            const double theta = g.log_uni(1e-7, 0.3), zs = (g.uni() - 0.5) * g.log_uni(1e-6, 0.1);
            double dir[3] = {1.0, -theta, zs};
            const double dn = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            const double dl = g.log_uni(1e-2, 1e3);
            o = v3((float)(P[0] - dir[0] / dn * rho), (float)(P[1] - dir[1] / dn * rho), (float)(P[2] - dir[2] / dn * rho));
            d = v3((float)(dir[0] / dn * dl), (float)(dir[1] / dn * dl), (float)(dir[2] / dn * dl));
        } else {
            double od[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
            const double on = std::sqrt(od[0] * od[0] + od[1] * od[1] + od[2] * od[2]) + 1e-30;
            o = v3((float)(c[0][0] + od[0] / on * rho), (float)(c[0][1] + od[1] / on * rho), (float)(c[0][2] + od[2] / on * rho));
            const double tgt[3] = {c[0][0] + (g.uni() - .5) * 0.44, c[0][1] + (g.uni() - .5) * 0.44, c[0][2] + (g.uni() - .5) * 0.44};
            const double dl = g.log_uni(1e-2, 1e2);
            double dd[3] = {tgt[0] - o.x, tgt[1] - o.y, tgt[2] - o.z};
            const double dn = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) + 1e-30;
            d = v3((float)(dd[0] / dn * dl), (float)(dd[1] / dn * dl), (float)(dd[2] / dn * dl));
        }
        // (a direction a camera can produce: lower_left_corner - origin in f32, main.rs:115-119 — so that a failing ray can be rendered)
        const V3 llc = v3(o.x + d.x, o.y + d.y, o.z + d.z);
        d = v3(llc.x - o.x, llc.y - o.y, llc.z - o.z);
        {   // inside the trusted ball (after the rounding of o)
            const double ox = o.x - dom.c0[0], oy = o.y - dom.c0[1], oz = o.z - dom.c0[2];
            if (!(ox * ox + oy * oy + oz * oz <= ball_r0 * ball_r0 * 0.998)) continue;
        }
        float t;
        if (!candidate(c[0], R, o, d, t)) continue;
        if (!slab_exact(U, o, d, T_MIN, INFINITY)) continue;          // BVHNode::hit never tests the unit for this ray
        counts[0]++;
        const float T = nextafter_up(t);
        if (slab_exact(G, o, d, T_MIN, T * (1.0f + pad))) continue;
        counts[1]++;
        // how early: the entry into the unit's box against t
        float lo = T_MIN;
        { const float q0 = (U.mnx - o.x) / d.x, q1 = (U.mxx - o.x) / d.x; lo = fmaxf(fminf(q0, q1), lo); }
        { const float q0 = (U.mny - o.y) / d.y, q1 = (U.mxy - o.y) / d.y; lo = fmaxf(fminf(q0, q1), lo); }
        { const float q0 = (U.mnz - o.z) / d.z, q1 = (U.mxz - o.z) / d.z; lo = fmaxf(fminf(q0, q1), lo); }
        double early = (double)lo / (double)t - 1.0;
        // (preferred for rendering: ordinary magnitudes — every direction component above 2e-6, so that the kernel's fast box test is
        // trusted for the ray (vk_trace.h set_space), a direction of length 10..300, early by a factor of 3+)
        if (fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)) > 2.0e-6f && length2(d) > 100.0f && length2(d) < 1.0e5f && early > 3.0 &&
            early < 50.0) early += 1.0e6;
        if (early > worst && viol) {
            worst = early;
            const float v[19] = {c[0][0], c[0][1], c[0][2], R, c[1][0], c[1][1], c[1][2], R, o.x, o.y, o.z, d.x, d.y, d.z, t, lo,
                llc.x, llc.y, llc.z};
            memcpy(viol, v, sizeof(v));
        }
    }
    return 0;
}

// Part D (round 5: the REFUTATION of "gate every sphere by its own box").  The same question for a gate made of the sphere's OWN box,
// center -+ radius, grown by rt_unit_growth applied to that box (d* = sqrt(3) (R + g), growth ~ 5e-4 R): rays from rho = 50 .. ball_r0
// away that pass the sphere at 1 + delta radii from its centre, delta = 1e-4 .. 0.5.  From hundreds of radii away the f32 quadratic
// reports roots for lines that miss the sphere by a good part of its radius (eta(rho) = 32 u (rho + R)^2 / R is the bound; the measured
// constant is ~1), i.e. for lines that miss the grown own box ALTOGETHER — the far-origin rule of the gate lemma presumes that the ray
// enters the gate box, which the reference's unit guarantees (BVHNode::hit tests X only for rays that pass the unit's box) and the
// sphere's own box does not.  counts = {rays with a candidate, of which the own-box gate is closed}; viol = the failing ray that misses
// by the most: c(3) r o(3) d(3) t miss/R.
int emu_own_gate_soundness(uint64_t n, uint64_t seed, float pad, double ball_r0, uint64_t counts[2], float viol[12]) {
    Lcg g(seed);
    counts[0] = counts[1] = 0;
    double worst = 0.0;
    for (uint64_t it = 0; it < n; it++) {
        const float R = 0.2f;
        float c[1][3] = {{(float)((g.uni() - 0.5) * 1000.0), 0.2f, (float)((g.uni() - 0.5) * 1000.0)}};
        const float mn[3] = {c[0][0] - R, c[0][1] - R, c[0][2] - R}, mx[3] = {c[0][0] + R, c[0][1] + R, c[0][2] + R};
        RtDomain dom; dom.c0[0] = c[0][0]; dom.c0[1] = c[0][1]; dom.c0[2] = c[0][2]; dom.r0 = ball_r0;
        const double gd = rt_unit_growth(mn, mx, 1, c, &R, dom, pad);
        if (gd < 0.0) continue;
        const float gg = std::nextafter((float)gd, INFINITY);
        DItem G; memset(&G, 0, sizeof(G));
        G.mnx = std::nextafter(mn[0] - gg, -INFINITY); G.mxx = std::nextafter(mx[0] + gg, INFINITY);
        G.mny = std::nextafter(mn[1] - gg, -INFINITY); G.mxy = std::nextafter(mx[1] + gg, INFINITY);
        G.mnz = std::nextafter(mn[2] - gg, -INFINITY); G.mxz = std::nextafter(mx[2] + gg, INFINITY);
        const double rho = g.log_uni(50.0, ball_r0 * 0.99), delta = g.log_uni(1e-4, 0.5);
        double od[3] = {g.uni() - .5, (g.uni() - .5) * 0.6 + 0.3, g.uni() - .5};
        const double on = std::sqrt(od[0] * od[0] + od[1] * od[1] + od[2] * od[2]) + 1e-30;
        double e1[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
        const double dp = (e1[0] * od[0] + e1[1] * od[1] + e1[2] * od[2]) / (on * on);
        for (int a = 0; a < 3; a++) e1[a] -= dp * od[a];
        const double en = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) + 1e-30;
        const V3 o = v3((float)(c[0][0] + od[0] / on * rho), (float)(c[0][1] + od[1] / on * rho), (float)(c[0][2] + od[2] / on * rho));
        double dd[3];
        const double tgt[3] = {c[0][0] + e1[0] / en * R * (1.0 + delta), c[0][1] + e1[1] / en * R * (1.0 + delta), c[0][2] + e1[2] / en * R * (1.0 + delta)};
        dd[0] = tgt[0] - o.x; dd[1] = tgt[1] - o.y; dd[2] = tgt[2] - o.z;
        const double dn = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) + 1e-30, dl = g.log_uni(0.5, 20.0);
        const V3 d = v3((float)(dd[0] / dn * dl), (float)(dd[1] / dn * dl), (float)(dd[2] / dn * dl));
        float t;
        if (!candidate(c[0], R, o, d, t)) continue;
        counts[0]++;
        if (slab_exact(G, o, d, T_MIN, nextafter_up(t) * (1.0f + pad))) continue;
        counts[1]++;
        // how far the LINE (as f32 holds it) misses the sphere: distance of the centre from it, in extended precision
        const LD ocx = (LD)o.x - c[0][0], ocy = (LD)o.y - c[0][1], ocz = (LD)o.z - c[0][2];
        const LD a = (LD)d.x * d.x + (LD)d.y * d.y + (LD)d.z * d.z, hb = ocx * d.x + ocy * d.y + ocz * d.z;
        const LD m2 = ocx * ocx + ocy * ocy + ocz * ocz - hb * hb / a;
        const double miss = (double)(sqrtl(m2 > 0 ? m2 : 0) / R) - 1.0;
        if (miss > worst && viol) {
            worst = miss;
            const float v[12] = {c[0][0], c[0][1], c[0][2], R, o.x, o.y, o.z, d.x, d.y, d.z, t, (float)miss};
            memcpy(viol, v, sizeof(v));
        }
    }
    return 0;
}
// one ray against one sphere: out = {candidate exists, t, the own-box gate grown by rt_unit_growth passes for tmax = next(t) (1 + pad), growth}
int emu_own_gate_ray(const float c[3], float r, const float o[3], const float d[3], float pad, double ball_r0, float out[4]) {
    float cc[1][3] = {{c[0], c[1], c[2]}};
    const float mn[3] = {c[0] - r, c[1] - r, c[2] - r}, mx[3] = {c[0] + r, c[1] + r, c[2] + r};
    RtDomain dom; dom.c0[0] = c[0]; dom.c0[1] = c[1]; dom.c0[2] = c[2]; dom.r0 = ball_r0;
    const double gd = rt_unit_growth(mn, mx, 1, cc, &r, dom, pad);
    if (gd < 0.0) return 1;
    const float gg = std::nextafter((float)gd, INFINITY);
    DItem G; memset(&G, 0, sizeof(G));
    G.mnx = std::nextafter(mn[0] - gg, -INFINITY); G.mxx = std::nextafter(mx[0] + gg, INFINITY);
    G.mny = std::nextafter(mn[1] - gg, -INFINITY); G.mxy = std::nextafter(mx[1] + gg, INFINITY);
    G.mnz = std::nextafter(mn[2] - gg, -INFINITY); G.mxz = std::nextafter(mx[2] + gg, INFINITY);
    float t = 0.0f;
    const bool has = candidate(c, r, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), t);
    out[0] = has ? 1.0f : 0.0f; out[1] = t;
    out[2] = (has && slab_exact(G, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), T_MIN, nextafter_up(t) * (1.0f + pad))) ? 1.0f : 0.0f;
    out[3] = gg;
    return 0;
}

// The device-only arithmetic of the sphere test (vk_trace.h refined_rcp / div_by_a: quotients by |d|^2 through a shared reciprocal) exists
// under __HIP_DEVICE_COMPILE__ only, because it starts from v_rcp_f32, whose bits the ISA does not specify beyond "within 1 ulp".  What CAN
// be checked on the host is the algorithm's claim: for ANY starting reciprocal within 1 ulp of 1/a, one Newton step and the two
// fma corrections yield the correctly rounded quotient n / a, for a in the range set_space admits (3e-12 .. 3e12) and |n| < 2^54.
// Returns the number of (n, a, perturbation) triples out of `cases` x 3 whose result differs from n / a; worst[] = {n, a, got, want}.
uint64_t emu_div_by_a_model(uint64_t cases, uint64_t seed, float worst[4]) {
    Lcg g(seed);
    uint64_t bad = 0;
    for (uint64_t it = 0; it < cases; it++) {
        const float a = (float)g.log_uni(3.0e-12, 3.0e12);
        float n = (float)(g.log_uni(1.0e-12, 1.0e16) * (g.uni() < 0.5 ? -1.0 : 1.0));
        if (g.uni() < 0.1) n = a * (float)(g.uni() * 4.0 - 2.0);              // quotients near 1: the sphere test's t ~ 1 cases
        const float want = n / a;
        if (!std::isfinite(want) || std::fabs(want) < 1.0e-30f) continue;       // (underflowing quotients are below tmin either way)
        const float y0 = 1.0f / a;
        for (int p = -1; p <= 1; p++) {
            float y = vk::bits_f32(vk::f32_bits(y0) + (uint32_t)p);             // what v_rcp_f32 may return
            const float e = __builtin_fmaf(-a, y, 1.0f);
            y = __builtin_fmaf(e, y, y);                                        // refined_rcp
            float q = n * y;                                                    // div_by_a
            float r = __builtin_fmaf(-a, q, n);
            q = __builtin_fmaf(r, y, q);
            r = __builtin_fmaf(-a, q, n);
            q = __builtin_fmaf(r, y, q);
            if (vk::f32_bits(q) != vk::f32_bits(want)) {
                if (bad == 0 && worst) { worst[0] = n; worst[1] = a; worst[2] = q; worst[3] = want; }
                bad++;
            }
        }
    }
    return bad;
}

// Part E (round 5): the NEAR form's three claims (vk_linearize.cpp rt_grow_near), each attacked with the kernel's own arithmetic.
// mode 0 — NEAR GATES ARE SOUND: a sphere of radius R (1e-2 .. 1e2, anywhere within 1e3 of the coordinate origin), its own box grown by
//   1.25 eta(rho_near) + 8 ulps, rho_near = sqrt(0.8 growth R^2 / kappa) - R as the lineariser sets it; rays from rho <= rho_near (inside the
//   sphere, on it, grazing it where it touches its box, nearly parallel to a face).  Every ray with a candidate must pass the gate with
//   tmax = next(t) (1 + RT_PAD_NEAR).          counts = {candidates, gates closed}
// mode 1 — REACH: rays from rho_near .. 1e5 away that pass the sphere at up to 1 + 64 u (rho / R)^2 radii (where false roots live): a
//   candidate's distance t |d| must exceed reach = rho_near - (R + eta(rho_near)).          counts = {candidates, t |d| <= reach}
// mode 2 — CLEARANCE: a ray that the device's clear_of_small_spheres lets through (outside the box around the spheres' surfaces grown by
//   M for every s >= reach): no sphere with rho > rho_near may hold a candidate.  Rays at the test's boundary; spheres placed where they
//   hurt, at the box's point nearest the ray.          counts = {rays x spheres tried, candidates}
int emu_near_form_claims(int mode, uint64_t n, uint64_t seed, uint64_t counts[2], float viol[12]) {
    Lcg g(seed);
    counts[0] = counts[1] = 0;
    const double U24 = 1.0 / 16777216.0;
    for (uint64_t it = 0; it < n; it++) {
        const float R = (float)g.log_uni(1e-2, 1e2);
        const double rho_near = std::sqrt(0.8 * RT_NEAR_GROWTH * (double)R * R / RT_KAPPA) - R;
        const double eta_near = rt_eta(rho_near, R);
        const double reach = (rho_near - (R + eta_near)) * (1.0 - 1e-5);
        float c[3] = {(float)((g.uni() - .5) * 2e3), (float)((g.uni() - .5) * 2e3), (float)((g.uni() - .5) * 2e3)};
        double maxabs = 0.0;
        for (int a = 0; a < 3; a++) maxabs = std::max(maxabs, std::fabs((double)c[a]) + R);
        const float gg = std::nextafter((float)(1.25 * eta_near + 8.0 * U24 * maxabs), INFINITY);
        DItem G; memset(&G, 0, sizeof(G));
        G.mnx = std::nextafter(c[0] - R - gg, -INFINITY); G.mxx = std::nextafter(c[0] + R + gg, INFINITY);
        G.mny = std::nextafter(c[1] - R - gg, -INFINITY); G.mxy = std::nextafter(c[1] + R + gg, INFINITY);
        G.mnz = std::nextafter(c[2] - R - gg, -INFINITY); G.mxz = std::nextafter(c[2] + R + gg, INFINITY);
        V3 o, d;
        if (mode <= 1) {
            const double rho = mode == 0 ? (g.uni() < 0.2 ? g.uni() * 1.2 * R : g.log_uni(0.5 * R, rho_near * 0.9999))
                                         : g.log_uni(rho_near * 1.0001, 1e5 * R);
            double od[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
            if (g.uni() < 0.3) od[(int)(g.uni() * 3) % 3] *= 1e-4;                  // origins in a face's plane
            const double on = std::sqrt(od[0] * od[0] + od[1] * od[1] + od[2] * od[2]) + 1e-30;
            o = v3((float)(c[0] + od[0] / on * rho), (float)(c[1] + od[1] / on * rho), (float)(c[2] + od[2] / on * rho));
            // aim at a point at (1 + delta) R from the centre, on a direction perpendicular to the origin's
            double e1[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
            if (g.uni() < 0.5) { const int a = (int)(g.uni() * 3) % 3; e1[0] = e1[1] = e1[2] = 0.0; e1[a] = 1.0; }      // the box's face centres
            const double dp = (e1[0] * od[0] + e1[1] * od[1] + e1[2] * od[2]) / (on * on);
            for (int a = 0; a < 3; a++) e1[a] -= dp * od[a];
            const double en = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) + 1e-30;
            const double wob = 64.0 * U24 * (rho / R) * (rho / R) + 1e-6;
            const double delta = (g.uni() < 0.5 ? 1.0 : -1.0) * g.log_uni(1e-9, 1.0) * (mode == 0 ? 0.05 : wob) + (g.uni() < 0.2 ? -g.uni() : 0.0);
            double dd[3];
            for (int a = 0; a < 3; a++) dd[a] = c[a] + e1[a] / en * R * (1.0 + delta) - o.x * (a == 0) - o.y * (a == 1) - o.z * (a == 2);
            const double dn = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) + 1e-30, dl = g.log_uni(1e-2, 1e2);
            d = v3((float)(dd[0] / dn * dl), (float)(dd[1] / dn * dl), (float)(dd[2] / dn * dl));
            {   // ordinary rays only (set_space): the others are never trusted
                const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
                if (!(ax > 1e-6f && ax < 1e6f && ay > 1e-6f && ay < 1e6f && az > 1e-6f && az < 1e6f)) continue;
            }
            float t;
            if (!candidate(c, R, o, d, t)) continue;
            counts[0]++;
            bool bad;
            if (mode == 0) {
                const double ox = (double)o.x - c[0], oy = (double)o.y - c[1], oz = (double)o.z - c[2];
                if (std::sqrt(ox * ox + oy * oy + oz * oz) > rho_near) { counts[0]--; continue; }       // (after rounding of o)
                bad = !slab_exact(G, o, d, T_MIN, nextafter_up(t) * (1.0f + (float)RT_PAD_NEAR));
            } else {
                bad = (double)t * std::sqrt((double)length2(d)) <= reach;
            }
            if (bad) {
                if (counts[1] == 0 && viol) { const float v[12] = {c[0], c[1], c[2], R, o.x, o.y, o.z, d.x, d.y, d.z, t, (float)rho_near}; memcpy(viol, v, sizeof(v)); }
                counts[1]++;
            }
            continue;
        }
        // ---- mode 2: a box of centres (flat fields, rows, cubes: each extent 0 .. 1e3), the spheres' surfaces R around it; the DEVICE's own
        // test (vk_trace.h clear_of_small_spheres) with the lineariser's constants.  The ray is pushed to the test's boundary: its origin is
        // bisected along an axis between a position where the test refuses and one where it accepts.
        // (mode 3, the control: M replaced by MINUS a quarter of a radius, i.e. rays that dip into the box are let through: candidates DO
        // appear, the test has teeth)
        DScene S; memset(&S, 0, sizeof(S));
        double ext[3], blo[3], bhi[3];
        for (int a = 0; a < 3; a++) {
            ext[a] = g.uni() < 0.3 ? 0.0 : g.log_uni(1e-1, 1e3);
            const double mid = (g.uni() - .5) * 1e3;
            blo[a] = (float)(mid - 0.5 * ext[a]); bhi[a] = (float)(mid + 0.5 * ext[a]);
            S.small_clo[a] = std::nextafter((float)blo[a] - R, -INFINITY); S.small_chi[a] = std::nextafter((float)bhi[a] + R, INFINITY);
        }
        double mab = 0.0;
        for (int a = 0; a < 3; a++) mab = std::max(mab, std::max(std::fabs((double)S.small_clo[a]), std::fabs((double)S.small_chi[a])));
        const double bb = std::sqrt(RT_KAPPA);
        S.reach = (float)reach;
        S.clear_k = (float)(1.02 * bb / (1.0 - 2.0 * bb)); S.clear_r2 = (float)(2.0 * (double)R * (1.0 + 1e-6)); S.clear_slack = (float)(64.0 * U24 * mab);
        S.t_pad = (float)RT_PAD_NEAR; S.gate_scale = 1.0f / (1.0f + S.t_pad);
        if (mode == 3) { S.clear_k = 0.0f; S.clear_slack = -0.25f * R; }
        if (mode == 4) { S.clear_k = 0.0f; S.clear_slack = 0.0f; }              // (no margin at all: how much of it is needed?)
        // a ray through the box region, shallow against one of its faces more often than not
        const int ax = (int)(g.uni() * 3) % 3;
        double dir[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
        {
            const double hn = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2] - dir[ax] * dir[ax]) + 1e-30;
            if (g.uni() < 0.7) dir[ax] = (g.uni() < 0.5 ? 1.0 : -1.0) * hn * g.log_uni(1e-5, 1.0);
        }
        const double dn0 = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]) + 1e-30, dl = g.log_uni(1e-2, 1e2);
        d = v3((float)(dir[0] / dn0 * dl), (float)(dir[1] / dn0 * dl), (float)(dir[2] / dn0 * dl));
        {
            const float fx = fabsf(d.x), fy = fabsf(d.y), fz = fabsf(d.z);
            if (!(fx > 1e-6f && fx < 1e6f && fy > 1e-6f && fy < 1e6f && fz > 1e-6f && fz < 1e6f)) continue;
        }
        const float dnf = sqrtf(length2(d));
        // (the lane's reciprocals: 1 / d off by up to an ulp, as v_rcp_f32's are, times gate_scale)
        auto rcp = [&](float x) { float r = 1.0f / x; const double u = g.uni(); r = u < 0.25 ? std::nextafter(r, -INFINITY) : u < 0.5 ? std::nextafter(r, INFINITY) : r; return r * S.gate_scale; };
        const V3 inv = v3(rcp(d.x), rcp(d.y), rcp(d.z));
        // the ray passes through a point of the box at distance s0 from its origin (s0 around reach, or far beyond it) ...
        double q[3];
        for (int a = 0; a < 3; a++) q[a] = blo[a] + g.uni() * (bhi[a] - blo[a]);
        const double s0 = g.uni() < 0.5 ? reach * g.log_uni(0.5, 4.0) : g.log_uni(std::max(reach, 1e-3), 3e3);
        double o0[3], o1[3];
        for (int a = 0; a < 3; a++) o0[a] = q[a] - (double)d.x * (a == 0) / dnf * s0 - (double)d.y * (a == 1) / dnf * s0 - (double)d.z * (a == 2) / dnf * s0;
        // ... and is shifted along an axis until the test accepts
        const int sa = g.uni() < 0.7 ? ax : (int)(g.uni() * 3) % 3;
        const double sgn = g.uni() < 0.5 ? 1.0 : -1.0;
        for (int a = 0; a < 3; a++) o1[a] = o0[a] + (a == sa ? sgn * (ext[a] + 4.0 * R + 0.1 * (s0 + 3e3)) : 0.0);
        auto at = [&](double l) { return v3((float)(o0[0] + l * (o1[0] - o0[0])), (float)(o0[1] + l * (o1[1] - o0[1])), (float)(o0[2] + l * (o1[2] - o0[2]))); };
        double l0 = 0.0, l1 = 1.0;
        if (clear_of_small_spheres(S, at(0.0), inv, dnf)) l1 = 0.0;              // (already clear: leaves the box before reach)
        else if (!clear_of_small_spheres(S, at(1.0), inv, dnf)) continue;
        else for (int k = 0; k < 48; k++) { const double lm = 0.5 * (l0 + l1); if (clear_of_small_spheres(S, at(lm), inv, dnf)) l1 = lm; else l0 = lm; }
        o = at(l1);
        if (!clear_of_small_spheres(S, o, inv, dnf)) continue;
        // spheres where they hurt: centres at the box's point nearest the ray, for the ray's closest approach beyond reach and a few more
        double best_s = reach, best_d = INFINITY;
        for (int k = 0; k < 96; k++) {
            const double s = reach * std::pow(3e3 / std::max(reach, 1e-3) + 2.0, k / 95.0);
            double dist2 = 0.0;
            for (int a = 0; a < 3; a++) {
                const double pa = (a == 0 ? o.x + (double)d.x / dnf * s : a == 1 ? o.y + (double)d.y / dnf * s : o.z + (double)d.z / dnf * s);
                const double e = pa < blo[a] ? blo[a] - pa : pa > bhi[a] ? pa - bhi[a] : 0.0;
                dist2 += e * e;
            }
            if (dist2 < best_d) { best_d = dist2; best_s = s; }
        }
        for (int k = 0; k < 6; k++) {
            const double s = k < 3 ? best_s * (1.0 + (g.uni() - .5) * 0.05) : g.log_uni(std::max(reach, 1e-3), 3e3);
            if (s < reach) continue;
            float cc[3];
            for (int a = 0; a < 3; a++) {
                const double pa = (a == 0 ? o.x + (double)d.x / dnf * s : a == 1 ? o.y + (double)d.y / dnf * s : o.z + (double)d.z / dnf * s);
                cc[a] = (float)std::min(std::max(pa + (k % 3 == 0 ? 0.0 : (g.uni() - .5) * 0.5 * R), blo[a]), bhi[a]);
            }
            const double ox = (double)o.x - cc[0], oy = (double)o.y - cc[1], oz = (double)o.z - cc[2];
            if (std::sqrt(ox * ox + oy * oy + oz * oz) <= rho_near) continue;             // (a near sphere: behind a sound gate)
            counts[0]++;
            float t;
            if (candidate(cc, R, o, d, t)) {
                if (counts[1] == 0 && viol) { const float v[12] = {cc[0], cc[1], cc[2], R, o.x, o.y, o.z, d.x, d.y, d.z, t, (float)rho_near}; memcpy(viol, v, sizeof(v)); }
                counts[1]++;
            }
        }
    }
    return 0;
}

// closest hit of ONE segment as the device decides it (rebuilt walk, segment_unsafe, the tree as handed over where needed):
// out = {T, best_prim (bits), 1 if the tree as handed over decided}
int emu_hit(const vk_scene_desc *desc, const float o[3], const float d[3], float out[3]) {
    LinearScene LS;
    int st = linearize_env(desc, LS, g_err);
    if (st != VK_OK) return st;
    if (LS.features != 0u) { g_err = "emu_hit: scenes of spheres only"; return VK_ERR_UNSUPPORTED; }
    std::vector<DItem> both;
    DScene S = scene_view(LS, both);
    const bool glob = global_variant() || S.walk_start != 0u;
    auto run = [&](auto mem_tag) {
        using Mem = decltype(mem_tag);
        Mem M; static_cast<GlobalMem &>(M) = GlobalMem{S.items, S.spheres, S.sphere_mat, S.boxes};
        Lane L; memset(&L, 0, sizeof(L));
        begin_segment<Mem::ISHIFT, fused_box<0u, Mem>(), true>(L, S, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), 0.0f);
        while (traversing(L)) {
            const uint32_t before = L.pend; const float Tb = L.T;
            traverse_step<0u, Mem>(L, S, M);
            if (getenv("EMU_TRACE") && before != 0u) fprintf(stderr, "  leaf objects %08x: T %.9g -> %.9g\n", before, Tb, L.T);
        }
        float redo = 0.0f;
        if (segment_unsafe<0u, Mem>(L, S, M)) {
            redo = 1.0f;
            if (S.walk_start != 0u) {
                begin_segment<Mem::ISHIFT, fused_box<0u, Mem>(), true>(L, S, L.o, L.d, L.time, true);
                while (traversing(L)) traverse_step<0u, Mem>(L, S, M);
            } else {
                DScene Sr = reference_view(S);
                Mem Mr; static_cast<GlobalMem &>(Mr) = GlobalMem{Sr.items, Sr.spheres, Sr.sphere_mat, Sr.boxes};
                begin_segment<Mem::ISHIFT, fused_box<0u, Mem>(), true>(L, Sr, v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), 0.0f);
                while (traversing(L)) traverse_step<0u, Mem>(L, Sr, Mr);
            }
        }
        out[0] = L.T; memcpy(&out[1], &L.best_prim, 4); out[2] = redo;
    };
    if (glob) run(GlobalMem{}); else run(FusedMem{});
    return VK_OK;
}

// debug: the ancestors of sphere `sidx`'s leaf in the tree as handed over, and whether each box passes AxisBB::hit(T_MIN, tmax) for the ray
int emu_unit_chain(const vk_scene_desc *desc, uint32_t sidx, const float o[3], const float d[3], float tmax) {
    LinearScene LS;
    int st = linearize_env(desc, LS, g_err);
    if (st != VK_OK) return st;
    const uint32_t ui = LS.unit_item.empty() ? 0xFFFFFFFFu : LS.unit_item[sidx];
    fprintf(stderr, "sphere %u unit item %u of %zu\n", sidx, ui, LS.ref_items.size());
    if (ui == 0xFFFFFFFFu) return 0;
    const V3 O = v3(o[0], o[1], o[2]), D = v3(d[0], d[1], d[2]);
    for (uint32_t i = 0; i <= ui; i++) {
        const DItem &it = LS.ref_items[i];
        const bool inner = (it.w0 >> 28) == 0u;
        const bool anc = i == ui || (inner && it.w0 > ui);
        if (!anc) continue;
        fprintf(stderr, "  item %u %s box (%.9g %.9g %.9g)-(%.9g %.9g %.9g) w %08x %08x passes %d\n", i, inner ? "inner" : "leaf", it.mnx, it.mny, it.mnz,
            it.mxx, it.mxy, it.mxz, it.w0, it.w1, (int)slab_exact(it, O, D, T_MIN, tmax));
    }
    return 0;
}

// Part F of the gate lemma's tests (round 5): the GRID form finds every candidate.  n rays against the world of `desc` (spheres only,
// eligible for the grid form): the closest hit of the grid walk — begin_segment, grid_step, prim_step, exactly as the device runs them
// — against the closest hit over ALL spheres, tested one by one with the same Sphere::hit arithmetic.  Rays where they hurt: origins on
// spheres, in them, far away (out to 1e4), directions along the layer, and lines that pass a random sphere at (1 + delta) radii, delta
// where the f32 discriminant reports hits that are not there.  counts = {rays, rays with a hit, rays whose two answers differ};
// viol = the first such ray (o, d, T grid, T brute force).
int emu_grid_claims(const vk_scene_desc *desc, uint64_t n, uint64_t seed, uint64_t counts[3], float viol[8]) {
    LinearScene LS;
    int st = linearize_env(desc, LS, g_err);
    if (st != VK_OK) return st;
    if (LS.features != 0u || LS.grid.nu == 0u) { g_err = "emu_grid_claims: a world the grid form applies to"; return VK_ERR_UNSUPPORTED; }
    std::vector<DItem> both;
    DScene S = scene_view(LS, both);
    if (S.grid.nu == 0u) { g_err = "emu_grid_claims: the grid is switched off"; return VK_ERR_UNSUPPORTED; }
    FusedMem M; static_cast<GlobalMem &>(M) = GlobalMem{S.items, S.spheres, S.sphere_mat, S.boxes};
    Lcg g(seed);
    counts[0] = counts[1] = counts[2] = 0;
    const double U24 = 1.0 / 16777216.0;
    const uint32_t ns = S.n_spheres;
    for (uint64_t it = 0; it < n; it++) {
        const DSphere sp = S.spheres[(uint32_t)(g.uni() * ns) % ns];
        const double R = sp.r, c[3] = {sp.cx, sp.cy, sp.cz};
        // the origin: on a sphere, inside one, near, far
        const double pick = g.uni();
        const double rho = pick < 0.3 ? R * (1.0 + 1e-4) : pick < 0.4 ? R * g.uni() : pick < 0.7 ? g.log_uni(1.5 * R, 50.0) : g.log_uni(50.0, 1e4);
        double od[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
        if (g.uni() < 0.5) od[1] = std::fabs(od[1]) * (g.uni() < 0.5 ? 1.0 : 0.05);        // above the layer, often low above it
        const double on = std::sqrt(od[0] * od[0] + od[1] * od[1] + od[2] * od[2]) + 1e-30;
        const V3 o = v3((float)(c[0] + od[0] / on * rho), (float)(c[1] + od[1] / on * rho), (float)(c[2] + od[2] / on * rho));
        // the direction: anywhere, along the layer, or past ANOTHER sphere at (1 + delta) of its radius
        double dd[3];
        const double kind = g.uni();
        if (kind < 0.25) { dd[0] = g.uni() - .5; dd[1] = g.uni() - .5; dd[2] = g.uni() - .5; }
        else if (kind < 0.45) { dd[0] = g.uni() - .5; dd[1] = (g.uni() - .5) * 0.02; dd[2] = g.uni() - .5; }
        else {
            const DSphere tq = S.spheres[(uint32_t)(g.uni() * ns) % ns];
            const double tr = tq.r, tc[3] = {tq.cx, tq.cy, tq.cz};
            double oc[3] = {tc[0] - o.x, tc[1] - o.y, tc[2] - o.z};
            const double dist = std::sqrt(oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2]) + 1e-30;
            double e1[3] = {g.uni() - .5, g.uni() - .5, g.uni() - .5};
            const double dp = (e1[0] * oc[0] + e1[1] * oc[1] + e1[2] * oc[2]) / (dist * dist);
            for (int a = 0; a < 3; a++) e1[a] -= dp * oc[a];
            const double en = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) + 1e-30;
            const double wob = 64.0 * U24 * (dist / tr) * (dist / tr) + 1e-6;
            const double delta = (g.uni() < 0.5 ? 1.0 : -1.0) * g.log_uni(1e-9, 1.0) * wob;
            for (int a = 0; a < 3; a++) dd[a] = tc[a] + e1[a] / en * tr * (1.0 + delta) - (a == 0 ? o.x : a == 1 ? o.y : o.z);
        }
        const double dn = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) + 1e-30, dl = g.log_uni(1e-2, 1e2);
        const V3 d = v3((float)(dd[0] / dn * dl), (float)(dd[1] / dn * dl), (float)(dd[2] / dn * dl));
        Lane L; memset(&L, 0, sizeof(L));
        begin_segment<FusedMem::ISHIFT, true, true>(L, S, o, d, 0.0f);
        if (!(L.xnan == L.xnan)) continue;                  // (not an ordinary ray: never trusted, segment_unsafe)
        while (traversing(L)) traverse_step<0u, FusedMem>(L, S, M);
        const float Tg = L.T;
        // every sphere, one by one
        float Tb = INFINITY;
        const float ya = refined_rcp(L.a);
        for (uint32_t k = 0; k < ns; k++) {
            const DSphere q = S.spheres[k];
            float t; bool tie;
            if (sphere_t_tie_y(q.cx, q.cy, q.cz, q.r, L.o, L.d, L.a, ya, S.fast_div != 0u, T_MIN, Tb, t, tie)) Tb = t;
        }
        counts[0]++;
        if (Tb < INFINITY) counts[1]++;
        if (vk::f32_bits(Tg) != vk::f32_bits(Tb)) {
            if (counts[2] == 0 && viol) { const float v[8] = {o.x, o.y, o.z, d.x, d.y, d.z, Tg, Tb}; memcpy(viol, v, sizeof(v)); }
            counts[2]++;
        }
    }
    return 0;
}

}  // extern "C"
