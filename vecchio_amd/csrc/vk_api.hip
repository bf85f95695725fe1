// vk_api.hip — the C ABI of include/vecchio_amd.h (libvecchio_amd.so) around the HIP megakernel of vk_kernels.h.
//
// Kernel structure (gfx950 / CDNA4, wave64):
//   * persistent workgroups (256-1024 threads, sized by plan_residency; sphere-only scenes staged in LDS run as TWO concurrent
//     launches, one 1024-thread and one 768-thread workgroup per CU = seven waves per SIMD: launch_dual); each WAVE pulls work units =
//     (8x8-pixel tile, sample chunk) from a global atomic counter and hands them to its lanes sample by
//     sample; it pulls the next unit the moment the current one is handed out (no per-unit drain);
//   * one ray per lane.  A lane whose path ended takes the next (pixel, sample) through a ballot +
//     prefix-popcount ("active-ray compaction"); the RNG is keyed (seed, pixel, sample) and pixel sums are
//     64-bit fixed point (integer atomics: LDS per tile, flushed to the frame's accumulators once per
//     unit), so the image does not depend on which lane, wave, unit, tile partition or GPU traced a
//     sample, nor on completion order;
//   * a wave-level phase scheduler runs, each round, the code of the state most lanes are in: BOX
//     (box_steps: nested steps under one shrinking EXEC mask, light primitive tests inline), PRIM heavy,
//     SHADE + REFILL (out of line for the everything-variants); lane state that only shading needs
//     (throughput, RNG, depth, pixel) is parked in LDS between SHADE phases so the traversal loops fit
//     80 VGPRs (6 waves/SIMD; 72 = 7 for sphere-only scenes in LDS; 64 = 8 for sphere-only scenes traversed from global memory);
//   * traversal is the stack-free threaded walk of vk_trace.h; when the linear BVH + spheres + boxes
//     fit next to that per-wave state WITHOUT costing occupancy, every workgroup stages them into its
//     LDS (160 KB/CU) once and item fetches are ds_read_b128; otherwise they are L1/L2 gathers;
//   * scenes of spheres only are walked on a tree REBUILT over the reference's leaf units (vk_linearize.cpp), and the tree as handed
//     over decides every segment whose winner could depend on the visiting order (vk_trace.h segment_unsafe): walked again in place
//     where both trees sit in one array (global-memory scenes), or the sample is queued and rendered by a second launch of the same
//     kernel in list mode (LDS scenes: enqueue_render_f32);
//   * no MFMA: there is no dense contraction in a path tracer.
//
// Host side: one vk_scene per device (scene upload, per-launch scratch); vk_scene_create_multi = a group of
// them: tiles dealt over the devices, each on its own stream, slabs moved to devices[0] by peer copies,
// de-interleaved on device, one device-to-host copy.
//
// There is NO CPU fallback in this library: every entry point either runs on a gfx950
// device or returns an error.

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>      // declarations only: librccl.so is loaded on request (VK_SCENE_RCCL_GATHER), never linked

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/vecchio_amd.h"
#include "../../include/vecchio_amd_debug.h"
#include "vk_linearize.h"
#include "vk_kernels.h"

namespace {

thread_local std::string g_err = "";

int fail(int code, const std::string &m) { g_err = m; return code; }

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return fail(VK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- RCCL, loaded on first use (VK_SCENE_RCCL_GATHER: the in-library gather of a multi-device scene as grouped ncclSend / ncclRecv).
// Not linked: a single-device host never maps librccl.so.
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
    bool ok() const { return handle && CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString; }
};
const RcclApi &rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
#ifdef VK_DEBUG_LIB
        // the DEBUG build may be pointed at a test double (tests/mock_rccl: the gather's orchestration on a one-GPU box)
        if (const char *e = getenv("VK_RCCL_LIB")) a.handle = dlopen(e, RTLD_NOW | RTLD_LOCAL);
#endif
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            if (a.handle) break;
            a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!a.handle) { const char *e = dlerror(); a.why = std::string("librccl.so could not be loaded: ") + (e ? e : "?"); return a; }
#define VK_RCCL_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, #sym))
        VK_RCCL_SYM(CommInitAll, ncclCommInitAll); VK_RCCL_SYM(CommDestroy, ncclCommDestroy); VK_RCCL_SYM(GroupStart, ncclGroupStart);
        VK_RCCL_SYM(GroupEnd, ncclGroupEnd); VK_RCCL_SYM(Send, ncclSend); VK_RCCL_SYM(Recv, ncclRecv);
        VK_RCCL_SYM(GetErrorString, ncclGetErrorString);
#undef VK_RCCL_SYM
        if (!a.ok()) a.why = "librccl.so lacks one of ncclCommInitAll / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv";
        return a;
    }();
    return api;
}
#define RCCL_TRY(expr)                                                                                                              \
    do {                                                                                                                            \
        ncclResult_t _r = (expr);                                                                                                   \
        if (_r != ncclSuccess) return fail(VK_ERR_HIP, std::string(#expr) + ": " + rccl_api().GetErrorString(_r));                  \
    } while (0)

// nothing may unwind across the C boundary: every entry point that can allocate runs through this
template <class Fn>
int guarded(Fn &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(VK_ERR_OOM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(VK_ERR_BAD_ARG, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(VK_ERR_BAD_ARG, "internal error");
    }
}

// Diagnostic switches (environment), read ONCE per scene at creation: none changes results except the sum grouping of
// VK_CHUNK_CAP.  DESIGN.md §6 lists them.
struct EnvSwitches {
    bool force_full_variant = false;   // VK_FORCE_FULL_VARIANT=1: run the everything-kernel
    bool no_lds_scene = false;         // VK_NO_LDS_SCENE=1: traverse from global memory at full occupancy
    int max_waves_per_cu = 0;          // VK_MAX_WAVES_PER_CU=n: lower the occupancy
    int chunk_cap = 0;                 // VK_CHUNK_CAP=n: samples per pixel per work unit
    int shade_defer = 0;               // VK_SHADE_DEFER=n
    bool tile_order = true;            // VK_TILE_ORDER=0: raster order, no probe launch; =1: dearest-first also for whole frames
    bool tile_order_forced = false;
    int probe_spp = 0;                 // VK_PROBE_SPP=n
    int probe_depth = 0;               // VK_PROBE_DEPTH=n: depth limit of the probe launch's paths (default 16)
    int prim_weight = 0;               // VK_PRIM_WEIGHT=n
    bool order_reuse = true;           // VK_ORDER_REUSE=0: probe the tile costs in every frame of a partition
    bool dual_same_stream = false;     // VK_DUAL_SAME_STREAM=1 (tests): see launch_dual
    bool dual_debug = false;           // VK_DUAL_DEBUG=1: print each checked frame's unit split
    // VK_RETREE=0/1/2: nothing rebuilt / every draw-free subtree / exact re-treeing (default: vk_scene_desc.flags)
    int retree = -1;
    int redo_region_cap = 0;           // VK_REDO_REGION_CAP=n (tests): entries per queue between the two launches of exact re-treeing
    bool grid_global = false;          // VK_GRID_GLOBAL=1 (comparisons): the grid form also for scenes traversed from global memory
    bool no_grid = false;              // VK_NO_GRID=1 (comparisons): the tree forms of exact re-treeing where the grid form would do
    int near_lds = -1;                 // VK_NEAR_LDS=0/1 (comparisons): the near form of exact re-treeing from global memory / staged in LDS
    static int int_env(const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; }
    static EnvSwitches read() {
        EnvSwitches v;
        if (const char *e = getenv("VK_FORCE_FULL_VARIANT")) v.force_full_variant = e[0] == '1';
        if (const char *e = getenv("VK_NO_LDS_SCENE")) v.no_lds_scene = e[0] == '1';
        if (const char *e = getenv("VK_TILE_ORDER")) { v.tile_order = e[0] != '0'; v.tile_order_forced = e[0] == '1'; }
        if (const char *e = getenv("VK_RETREE")) v.retree = (e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
        if (const char *e = getenv("VK_DUAL_SAME_STREAM")) v.dual_same_stream = e[0] == '1';
        if (const char *e = getenv("VK_ORDER_REUSE")) v.order_reuse = e[0] != '0';
        if (const char *e = getenv("VK_DUAL_DEBUG")) v.dual_debug = e[0] == '1';
        v.max_waves_per_cu = int_env("VK_MAX_WAVES_PER_CU");
        v.chunk_cap = int_env("VK_CHUNK_CAP");
        v.shade_defer = int_env("VK_SHADE_DEFER");
        v.probe_spp = int_env("VK_PROBE_SPP");
        v.probe_depth = int_env("VK_PROBE_DEPTH");
        v.prim_weight = int_env("VK_PRIM_WEIGHT");
        v.redo_region_cap = int_env("VK_REDO_REGION_CAP");
        if (const char *e = getenv("VK_NEAR_LDS")) v.near_lds = e[0] != '0';
        if (const char *e = getenv("VK_NO_GRID")) v.no_grid = e[0] == '1';
        if (const char *e = getenv("VK_GRID_GLOBAL")) v.grid_global = e[0] == '1';
        return v;
    }
};

}  // namespace

// =========================================================================================
// One vk_scene = the linearised scene resident on ONE device plus the per-launch scratch of the (at most one)
// render in flight on it.  A multi-device scene (vk_scene_create_multi) is a group handle: `parts` holds one
// ordinary single-device scene per listed device, each with its own stream.
struct vk_scene {
    int device = 0;
    std::shared_ptr<const LinearScene> host;   // shared by the parts of a multi-device scene
    DScene dev;                // device pointers
    EnvSwitches env;
    std::vector<void *> allocs;
    uint32_t *counter = nullptr;
    float *fb = nullptr; size_t fb_bytes = 0;        // f32 framebuffer (vk_render; RGB8 output; the parts' render targets)
    uint8_t *fb8 = nullptr; size_t fb8_bytes = 0;    // RGB8 image for vk_render with VK_OUTPUT_RGB8
    long long *accum = nullptr; size_t accum_bytes = 0;   // fixed-point pixel sums of the render in flight
    float4 *debug = nullptr; size_t debug_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cus = 256;
    uint32_t lds_bytes = 0;    // hot-record bytes staged per workgroup (0 = not LDS resident)
    bool grid_on = false;      // the grid form of exact re-treeing is this scene's walk (DGrid)
    uint32_t grid_slots = 0;   // the grid form: size of the table [cells | refs] in 32-byte units (KArgs::lds_items of its launches)
    size_t hot_bytes = 0;      // items + spheres + boxes: what traversal gathers from
    uint32_t wg_threads = 512; // workgroup size chosen by plan_residency()
    uint32_t sphere_waves = 6; // waves per SIMD of the sphere-only variant (8 was measured 3 % slower: it spills)
    uint32_t wgs_per_cu = 2;
    // Seven waves per SIMD for sphere-only scenes staged in LDS: ONE 1024-thread and ONE 768-thread workgroup per CU, i.e. two
    // concurrent launches of the same kernel pulling from the same unit counter (plan_residency); the second one runs on `stream2`.
    bool dual_launch = false;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Self-check of the dual launch: the two launches must OVERLAP, or the first one does all the work at 16 waves per CU.  Each
    // launch counts the units it pulls (two words of the counter block); the 768-thread launch should get ~12/28 of them.  A frame in
    // which one launch got under a tenth counts as a strike; after two strikes in a row the scene uses the single-launch shape for
    // good.  Checked where the caller synchronises anyway (vk_scene_last_kernel_ms, vk_render).
    bool dual_last = false;        // the last render used the dual launch
    int dual_strikes = 0;
    bool last_timed = false;
    // Exact re-treeing (vk_trace.h): the scene's own tree is a rebuilt one; samples it cannot vouch for are queued by the first launch
    // and rendered by a second one on `ref_view`, the scene as handed over.  redo_count: REDO_REGIONS counters + the 3 plan words.
    bool exact = false;            // host->ref_items is there and the switch VK_EXACT_RETREE is not 0
    DScene ref_view;
    uint2 *redo_list = nullptr; size_t redo_bytes = 0;
    uint32_t *redo_count = nullptr;
    bool redo_last = false;        // the last render had a second launch
    unsigned long long *wave_times = nullptr;      // VK_WAVE_TIMES=1 (diagnostics)
    // The rebuilt tree is SUSPENDED for a while when a frame sends more than a quarter of its samples through the second launch, or
    // overflows the queues between the launches (the fallback launch then renders the frame a third time): the scene renders on the tree
    // as handed over until frame `exact_resume`, then tries again; every relapse doubles the pause (32 frames .. 4096).  An animation
    // that passes through one bad viewpoint loses the rebuilt tree for a few dozen frames, not for good.  The verdict on a frame is read
    // from `plan_host` (pinned; copied behind the frame's last kernel) when the NEXT frame is enqueued, without waiting, or where the
    // caller synchronises anyway (vk_scene_last_requeued_samples).
    uint64_t frame_no = 0, exact_resume = 0, exact_pause = 32;
    // Two verdict slots, used in turn: [4] words of a frame's redo_plan each, an event recorded right behind the copy, the samples of the
    // partition the frame covered.  A caller that always enqueues frame N + 1 before frame N has finished (vk_render_device in a
    // pipeline) still has frame N - 1's verdict taken when it enqueues frame N + 1: the verdict does not wait for the MOST RECENT frame.
    uint32_t *plan_host = nullptr;     // [2][4]
    hipEvent_t ev_plan[2] = {nullptr, nullptr};
    bool plan_pending[2] = {false, false};      // a frame with a second launch has been enqueued and its plan not judged yet
    uint64_t plan_samples[2] = {0, 0};
    int plan_last = 0;                 // the slot of the last frame with a second launch
    bool plan_copied = false;          // ... whose plan did travel to that slot
    uint64_t redo_last_samples = 0;    // samples of the partition the last render covered
    unsigned long long *phase_stats = nullptr;   // device, 24 counters (diagnostic kernel build)
    bool want_phase_stats = false;
    // heavy-first tile order: per-tile times of the probe launch and the order derived from them
    uint32_t *tile_cost = nullptr, *tile_order = nullptr, *order_hist = nullptr;
    size_t tile_cost_n = 0, tile_order_n = 0;
    // Frame-to-frame reuse of the order: the frames of an animation (and the steps of a benchmark) see nearly the same tile costs,
    // so the order found for a partition is kept while the partition's geometry is the same, the camera has hardly moved and the
    // order is younger than ORDER_MAX_AGE frames; then the probe launch and the three sorting kernels are skipped (~1.7 ms of a
    // 1/8 share of C2's 42 ms).  The order never changes a pixel, so a stale one only costs balance.
    struct {
        uint32_t width = 0, height = 0, rank = 0, world = 0, depth = 0, age = 0;
        float org[3] = {0, 0, 0}, llc[3] = {0, 0, 0};
        bool valid = false;
    } order_for;
    // ---- multi-device group (empty for an ordinary scene)
    std::vector<vk_scene *> parts;
    // a part's own stream, its slab (on its device), the slab's landing buffer on devices[0] and the event that says it landed
    hipStream_t stream = nullptr;
    uint8_t *slab = nullptr; size_t slab_bytes = 0;
    uint8_t *landing = nullptr; size_t landing_bytes = 0; int landing_device = 0;
    hipEvent_t ev_landed = nullptr;
    hipEvent_t ev_begin = nullptr;               // group: recorded on the caller's stream at the start of a frame
    // group, VK_SCENE_RCCL_GATHER: one communicator per part (rank j = devices[j]); empty = peer copies
    std::vector<ncclComm_t> comms;
};

namespace {

template <class T>
int upload(vk_scene *s, const std::vector<T> &v, const T *&dptr) {
    dptr = nullptr;
    size_t bytes = v.size() * sizeof(T);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    s->allocs.push_back(p);
    if (bytes) HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
    dptr = reinterpret_cast<const T *>(p);
    return VK_OK;
}

// grows a device buffer owned by the scene (never shrinks); the scene's device must be current
template <class T>
int ensure(T *&ptr, size_t &have, size_t need) {
    if (need <= have && ptr) return VK_OK;
    if (ptr) HIP_TRY(hipFree(ptr));
    ptr = nullptr; have = 0;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, need ? need : 16));
    ptr = reinterpret_cast<T *>(p); have = need;
    return VK_OK;
}

uint32_t pick_variant(const vk_scene *s) {
    const uint32_t features = s->host->features;
    const uint32_t F_CORNELL = VKF_RECT | VKF_LIST | VKF_INSTANCE | VKF_BOX;
    if (s->env.force_full_variant) return VKF_ALL_SCENE;   // diagnostics: cost of the general kernel
    if (features == 0) return 0u;
    if ((features & ~F_CORNELL) == 0) return F_CORNELL;
    return VKF_ALL_SCENE;
}

size_t per_wave_lds_bytes(uint32_t F) {   // cold lane state of one wave + its tile's fixed-point sums (64 x 3 x 8 B) + its unit state
    return ((F & VKF_INSTANCE) ? wave_block_floats<VKF_INSTANCE>() : wave_block_floats<0u>()) * sizeof(float);
}

// Waves per SIMD the variants with heavy primitives are built for (their __launch_bounds__): 80 VGPRs = 6, 96 = 5.
#ifndef VK_ALL_MINW
#define VK_ALL_MINW 6
#endif
#ifndef VK_CORNELL_MINW
#define VK_CORNELL_MINW 6
#endif
// LDS residency plan.  `hot` = bytes of items + spheres + boxes.  Measured on MI355X with the VALU-bound
// kernel: at EQUAL occupancy a scene staged in LDS beats the same scene read through L1/L2 by only 7 % (C2:
// 4.06 vs 3.77 Gsamples/s at 24 waves/CU; C4: no difference), while occupancy is worth much more (C3: the
// 118 KB scene in LDS leaves 11 waves/CU = 371 Msamples/s; from L2 at 16 waves/CU = 503).  So the scene is
// staged in LDS only when that costs no waves: choose the workgroup size (waves share one LDS copy) and
// workgroups per CU that reach the variant's full occupancy (24 waves/CU at 80 VGPRs) with the
// scene resident, else traverse from global memory at full occupancy.
void plan_residency(vk_scene *s, size_t hot) {
    const size_t pw = per_wave_lds_bytes(pick_variant(s));
    uint32_t best_waves = 0, best_wg = 0, best_n = 0;
    const bool spheres_only = pick_variant(s) == 0u;
    // The near form of exact re-treeing walks a failed segment again in place: both trees in items[], i.e. global memory — unless its
    // reach spans the small spheres' whole box: then hardly a segment fails (the InOneWeekend scene: 3 in 10^5), a failed one may as well
    // requeue its whole sample, and the scene is staged in LDS like any other (7 520 against the unit form's 7 285 Msamples/s at 256 spp)
    const bool near_needs_global = s->host->near_form && !s->grid_on && (s->env.near_lds >= 0 ? s->env.near_lds == 0 : !s->host->near_spans);
    const uint32_t per_simd = spheres_only ? s->sphere_waves : (pick_variant(s) == (uint32_t)VKF_ALL_SCENE ? (uint32_t)VK_ALL_MINW
                                                                                                           : (uint32_t)VK_CORNELL_MINW);   // = MINW of launch_variant
    uint32_t cap = 4 * per_simd;                                             // waves per CU the variant's register budget admits
    // Workgroups hold a multiple of 4 waves that divides evenly over the CU's four SIMDs: the dispatcher deals a workgroup's waves
    // round-robin, so e.g. two 10-wave workgroups land 3+3+2+2 twice and the second one does not fit beside the first at 5 per SIMD
    const uint32_t max_wg_waves = 12;                                        // <= the variants' __launch_bounds__ thread limit / 64
    { int v = s->env.max_waves_per_cu; if (v >= 4 && (uint32_t)v < cap) cap = (uint32_t)v; }   // diagnostics: lower the occupancy
    for (uint32_t n_wg = 1; n_wg <= 6; n_wg++) {
        size_t budget = LDS_PER_CU / n_wg;
        if (hot + 4 * pw > budget) break;
        uint32_t w = (uint32_t)std::min<size_t>(std::min<uint32_t>(max_wg_waves, cap / n_wg), (budget - hot) / pw);
        w &= ~3u;                          // (a multiple of four: see above)
        if (w < 4) break;
        if (w * n_wg > best_waves) { best_waves = w * n_wg; best_wg = w; best_n = n_wg; }
    }
    // Seven waves per SIMD (the sphere-only kernels need 72 VGPRs): 28 waves per CU cannot be two EQUAL workgroups — 14 waves land
    // 4+4+3+3 on the four SIMDs and the second workgroup does not fit beside the first — but they can be 16 + 12: a 1024-thread
    // workgroup (4 per SIMD) and a 768-thread one (3 per SIMD), from two concurrent launches.  Needs two LDS copies of the scene.
    s->dual_launch = false;
    // (the near form of exact re-treeing walks a failed segment again in place: both trees in items[], i.e. global memory)
    if (spheres_only && !s->env.no_lds_scene && !near_needs_global && s->env.max_waves_per_cu == 0 && !getenv("VK_NO_DUAL_LAUNCH") &&
        2 * hot + 28 * pw <= LDS_PER_CU) {
        s->lds_bytes = (uint32_t)hot; s->wg_threads = 768; s->wgs_per_cu = 2;      // (the single-launch shape: probe, STATS, tiny frames)
        s->dual_launch = true;
        return;
    }
    if (best_waves >= cap && !s->env.no_lds_scene && !near_needs_global) {
        s->lds_bytes = (uint32_t)hot; s->wg_threads = best_wg * 64; s->wgs_per_cu = best_n;
    } else {
        // 28 (sphere-only: seven 4-wave workgroups, 7 waves/SIMD) or 24 waves per CU
        s->lds_bytes = 0; s->wg_threads = 256; s->wgs_per_cu = spheres_only ? 7 : per_simd;
    }
}

template <uint32_t F, int MINW_SPHERES = 6>
int launch_variant(vk_scene *s, const KArgs &A, bool lds, dim3 grid, size_t shmem, hipStream_t st, bool cost) {
    // Register budget: every variant is held to 80 VGPRs = 6 waves per SIMD, 24 per CU.  The sphere-only kernels fit (76).  The
    // Cornell-type variants (Rect / list / Boxy / instance) need 96 and the everything-variants 120 to be free of spills, but both
    // gain more from the waves than they lose to the spills: C4 at 4 / 5 / 6 / 7 per SIMD 4 430 / 5 240 / 5 425 / 5 030 Msamples/s (45
    // spilled
    // registers at 6, 80 at 7, shading inline; out of line 5 340 at 6); C3, which waits for memory 44 % of the time, 642 / 695 / 726 at 4 /
    // 5 / 6
    // (13 spilled registers, 25 scratch instructions outside the box loop, shading out of line) and 695 at 7 (27 registers, 154).
    constexpr int MINW = ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) ? MINW_SPHERES
                                                                : ((F & VKF_ALL_SCENE) == VKF_ALL_SCENE ? VK_ALL_MINW : VK_CORNELL_MINW);
    // Sphere-only scenes traversed from GLOBAL memory (C5, 49 MB of items and spheres): every box step is a dependent gather there, so
    // waves in flight pay.  On the tree handed over 8 waves per SIMD / 64 VGPRs with the shading phase out of line were best (629 -> 685
    // Msamples/s over 6); on the rebuilt tree of exact re-treeing the walks are half as long and the 64-VGPR build's spills (72 B of
    // scratch per lane, 2 TB per frame) weigh more than the eighth wave: 6 / 7 / 8 waves per SIMD -> 1 112 / 1 163 / 1 076 Msamples/s.
    // Seven: 72 VGPRs, shading inline, seven 256-thread workgroups per CU.
    constexpr int MINW_G = ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) ? 7 : MINW;
    auto go = [&](auto kernel) -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(kernel, grid, dim3(s->wg_threads), shmem, st, A);
        return VK_OK;
    };
    int rc;
    if constexpr (F == 0u) {
        if (A.S.grid.nu != 0u) {      // the grid form of exact re-treeing (DGrid): worlds without lights, i.e. the scatter integrator's
            if (cost) rc = lds ? go(&render_kernel<F, true, MINW, false, true, true>) : go(&render_kernel<F, false, MINW_G, false, true, true>);
            else rc = lds ? go(&render_kernel<F, true, MINW, false, false, true>) : go(&render_kernel<F, false, MINW_G, false, false, true>);
            if (rc != VK_OK) return rc;
            HIP_TRY(hipGetLastError());
            return VK_OK;
        }
    }
    if (cost) rc = lds ? go(&render_kernel<F, true, MINW, false, true>) : go(&render_kernel<F, false, MINW_G, false, true>);
    else rc = lds ? go(&render_kernel<F, true, MINW, false, false>) : go(&render_kernel<F, false, MINW_G, false, false>);
    if (rc != VK_OK) return rc;
    HIP_TRY(hipGetLastError());
    return VK_OK;
}

// The dual launch of plan_residency: the same 7-waves-per-SIMD build of a sphere-only LDS variant, once with 1024-thread workgroups on
// `st` and once with 768-thread workgroups on the scene's second stream, one workgroup of each per CU; both pull units from A.counter.
template <uint32_t F, bool GRID = false>
int launch_dual(vk_scene *s, const KArgs &A, size_t per_wave, hipStream_t st) {
    auto kernel = &render_kernel<F, true, 7, false, false, GRID>;
    const size_t shm_a = s->lds_bytes + 16 * per_wave, shm_b = s->lds_bytes + 12 * per_wave;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_a));
    // (VK_DUAL_SAME_STREAM=1, tests: both launches on ONE stream, i.e. serialised — what the self-check must notice)
    hipStream_t st2 = s->env.dual_same_stream ? st : s->stream2;
    HIP_TRY(hipEventRecord(s->ev_fork, st));                    // everything enqueued so far (memsets of counter and sums)
    HIP_TRY(hipStreamWaitEvent(st2, s->ev_fork, 0));
    hipLaunchKernelGGL(kernel, dim3((unsigned)s->num_cus), dim3(1024), shm_a, st, A);
    hipLaunchKernelGGL(kernel, dim3((unsigned)s->num_cus), dim3(768), shm_b, st2, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev_join, st2));
    HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0));              // the resolve kernel waits for both
    s->dual_last = true;
    return VK_OK;
}

// cost = the probe build of the variant (per-tile times into A.tile_cost)
int launch_by_features(vk_scene *s, uint32_t F, const KArgs &A, bool lds, dim3 grid, size_t shmem, hipStream_t st, bool cost) {
    const uint32_t F_CORNELL = VKF_RECT | VKF_LIST | VKF_INSTANCE | VKF_BOX;
    switch (F) {
        case 0u: return launch_variant<0u>(s, A, lds, grid, shmem, st, cost);
        case VKF_INTEG_PDF: return launch_variant<VKF_INTEG_PDF>(s, A, lds, grid, shmem, st, cost);
        case F_CORNELL: return launch_variant<F_CORNELL>(s, A, lds, grid, shmem, st, cost);
        case F_CORNELL | VKF_INTEG_PDF: return launch_variant<(F_CORNELL | VKF_INTEG_PDF)>(s, A, lds, grid, shmem, st, cost);
        case VKF_ALL_SCENE: return launch_variant<VKF_ALL_SCENE>(s, A, lds, grid, shmem, st, cost);
        default: return launch_variant<(VKF_ALL_SCENE | VKF_INTEG_PDF)>(s, A, lds, grid, shmem, st, cost);
    }
}

int check_render_args(vk_scene *scene, const vk_camera *cam, const vk_render_params *p) {
    if (!scene || !cam || !p) return fail(VK_ERR_BAD_ARG, "null argument");
    if (p->width < 2 || p->height < 2) return fail(VK_ERR_BAD_ARG,
        "width and height must be >= 2 (u,v divide by width-1/height-1, main.rs:187-188)");
    if ((uint64_t)p->width * p->height > (1ull << 31) / 3 || p->width > 65535u || p->height > 65535u) return fail(VK_ERR_BAD_ARG,
        "image too large");
    if (p->samples_per_pixel == 0 || p->samples_per_pixel > (1u << 26)) return fail(VK_ERR_BAD_ARG, "samples_per_pixel must be in 1..2^26");
    if (!(cam->time0 < cam->time1)) return fail(VK_ERR_BAD_ARG, "camera time0 >= time1 (gen_range panics, main.rs:118)");
    if (p->integrator > VK_INTEGRATOR_SCATTER || p->background > VK_BACKGROUND_SKY) return fail(VK_ERR_BAD_ARG,
        "bad integrator/background");
    if (p->output_format > VK_OUTPUT_RGB8) return fail(VK_ERR_BAD_ARG, "bad output_format");
    uint32_t world = p->tile_world ? p->tile_world : 1;
    if (p->tile_rank >= world) return fail(VK_ERR_BAD_ARG, "tile_rank >= tile_world");
    const LinearScene &H = *scene->host;
    if (p->integrator == VK_INTEGRATOR_PDF && H.lights.empty())
        return fail(VK_ERR_UNSUPPORTED, "PDF integrator with an empty lights list (Vec::random unwraps None, hittable.rs:431)");
    if (p->integrator == VK_INTEGRATOR_SCATTER && (H.features & VKF_SPEC_DIFFUSE))
        return fail(VK_ERR_UNSUPPORTED,
            "SpecDiffuse has no Material::scatter (default impl unwraps a None specular ray, material.rs:21-28)");
    return VK_OK;
}

struct TileGeom {      // the tile partition of one call
    uint32_t tiles_x, tiles_y, tiles, rank, world, n_local;
    TileGeom(const vk_render_params *p) {
        tiles_x = (p->width + TILE - 1) / TILE; tiles_y = (p->height + TILE - 1) / TILE; tiles = tiles_x * tiles_y;
        world = p->tile_world ? p->tile_world : 1; rank = p->tile_rank;
        n_local = tiles > rank ? (tiles - rank + world - 1) / world : 0;
    }
};

template <int MODE>
int tile_move(const void *src, void *dst, const vk_render_params *p, const TileGeom &g, hipStream_t st) {
    if (g.n_local == 0) return VK_OK;
    size_t n = (size_t)g.n_local * 64u;
    hipLaunchKernelGGL(tile_move_kernel<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, p->width, p->height,
        g.tiles_x,
                       g.rank, g.world, g.n_local);
    HIP_TRY(hipGetLastError());
    return VK_OK;
}

uint64_t partition_samples(const vk_render_params *p, const TileGeom &g) {
    uint64_t px = 0;
    for (uint32_t t = g.rank; t < g.tiles; t += g.world) {
        uint32_t tx = (t % g.tiles_x) * TILE, ty = (t / g.tiles_x) * TILE;
        px += (uint64_t)std::min<uint32_t>(TILE, p->width - tx) * std::min<uint32_t>(TILE, p->height - ty);
    }
    return px * p->samples_per_pixel;
}

// number of sample chunks per tile.  Pixel sums are order independent, so this only sets the granularity of the work
// units (locality of a wave's samples against the spread of expensive tiles over many waves), never a pixel's value
uint32_t choose_chunks(const vk_scene *s, const vk_render_params *p) {
    // Samples per pixel per unit: a unit keeps a wave on one tile (coherent primary rays, LDS tile sums flushed once per
    // unit); small enough that tiles of very different cost (fog, glass, grazing rays over 1M spheres) are spread over many
    // waves and that small images still give ~64K units for ~6K waves.  (C2: 32 / 64 / 128 spp per unit -> 5 218 / 5 235 / 5 221.)
    uint64_t tiles = (uint64_t)((p->width + TILE - 1) / TILE) * ((p->height + TILE - 1) / TILE);
    uint64_t c = (uint64_t)p->samples_per_pixel * (tiles / (p->tile_world ? p->tile_world : 1u)) / 65536u;      // ~64K units per launch
    // One GPU: 64 (C2: 32 / 64 / 128 / 256 -> 5 218 / 5 235 / 5 221 / 4 960 Msamples/s).  One rank of N: the launch ends on the last
    // units of the rank's dearest tiles, so smaller ones (C2's 1/8 share: 64 -> 54.2 ms, 32 -> 53.1, 16 -> 52.9, 8 -> 53.5; ideal 50.0).
    // (seven waves per SIMD — the dual launch of sphere-only LDS scenes — like the smaller units too: 16 / 32 / 48 / 64 -> 6 805 / 6 836 /
    // 6 815 / 6 776 Msamples/s on C2 at full size)
    uint32_t cap = (p->tile_world > 1u || s->dual_launch) ? 32u : 64u;
    // (Round 1 and the first half of round 2 cut the units of scenes bigger than an XCD's L2 down to 8 spp "because tile costs are
    // skewed by orders of magnitude": the skew was NaN rays walking the whole million-item tree — see begin_segment in vk_trace.h.
    // Without them C5 prefers the common setting: 4 / 8 / 16 / 32 / 64 spp per unit -> 564 / 574 / 579 / 583 / 585 Msamples/s.)
    if (s->env.chunk_cap >= 1) cap = (uint32_t)s->env.chunk_cap;   // diagnostics
    uint32_t lo = cap < 32 ? cap : 32;
    uint32_t chunk_spp = (uint32_t)(c > cap ? cap : (c < lo ? lo : c));
    // the unit counter is 32 bits wide: tiles x chunks must stay below 2^32 (4096 x 4096 at 2^20 spp would not at 64 spp per unit)
    {
        const uint64_t n_local = tiles / (p->tile_world ? p->tile_world : 1u) + 1u;
        const uint64_t min_chunk = ((uint64_t)p->samples_per_pixel * n_local + 0xE0000000ull - 1u) / 0xE0000000ull;
        if (chunk_spp < min_chunk) chunk_spp = (uint32_t)min_chunk;
    }
    uint32_t n = (p->samples_per_pixel + chunk_spp - 1) / chunk_spp;
    if (n < 1) n = 1;
    return n;
}

// The verdict on a frame with a second launch (its plan has arrived in slot b of plan_host): see vk_scene::exact_resume.
void judge_frame(vk_scene *s, int b) {
    s->plan_pending[b] = false;
    const uint32_t requeued = s->plan_host[4 * b + 1], lost = s->plan_host[4 * b + 2];
    const uint64_t frame_samples = s->plan_samples[b];
    const bool heavy = (uint64_t)requeued * 4u > frame_samples && frame_samples >= (1u << 20);
    if (lost != 0u || heavy) {
        s->exact_resume = s->frame_no + s->exact_pause;
        fprintf(stderr, "vecchio_amd: exact re-treeing %s (%u of %llu samples requeued, %u did not fit); this scene renders on the tree as "
            "handed over for the next %llu frames\n", lost ? "overflowed its queues and the frame was rendered again" : "sent over a quarter of "
            "a frame through the second launch", requeued, (unsigned long long)frame_samples, lost, (unsigned long long)s->exact_pause);
        s->exact_pause = std::min<uint64_t>(s->exact_pause * 2u, 4096u);
    } else if (s->exact_pause > 32u) {
        s->exact_pause /= 2u;           // a clean frame on the rebuilt tree: relapses are forgiven step by step
    }
}

// Enqueues one render of this call's tile partition into the f32 framebuffer d_out (device memory of s->device) on `st`.
int enqueue_render_f32(vk_scene *s, const vk_camera *cam, const vk_render_params *p, float *d_out, hipStream_t st, bool want_debug,
    vk_stats *stats) {
    HIP_TRY(hipSetDevice(s->device));
    const TileGeom g(p);
    KArgs A;
    memset(&A, 0, sizeof(A));
    A.S = s->dev;
    // the verdicts that have arrived (never waits): the older slot first
    for (int k = 1; k <= 2; k++) {
        const int b = (s->plan_last + k) & 1;
        if (s->plan_pending[b] && hipEventQuery(s->ev_plan[b]) == hipSuccess) judge_frame(s, b);
    }
    (void)hipGetLastError();      // (hipErrorNotReady of a query is not an error of this call)
    s->frame_no++;
    bool exact = s->exact && !s->want_phase_stats && s->frame_no >= s->exact_resume;     // (the diagnostic builds have no second launch)
    // The near form: primary rays start on the tree as handed over when the camera (its lens included) is farther than `reach` from every
    // sphere — their walk on the rebuilt tree could not stand (vk_trace.h begin_segment).  Decided from the box around the small spheres
    // and the surfaces of the few big ones; when in doubt: no.
    if (s->host->near_form && !s->grid_on && (A.S.walk_start != 0u || s->exact) && s->host->n_big != 0xFFFFFFFFu) {
        const LinearScene &H = *s->host;
        const double reach = (double)H.reach + (double)fabsf(cam->lens_radius) * 1.5 + 1e-3 * (double)H.reach;
        double d2 = 0.0;
        for (int k = 0; k < 3; k++) {
            const double o = cam->origin[k], e = o < H.small_lo[k] ? H.small_lo[k] - o : (o > H.small_hi[k] ? o - H.small_hi[k] : 0.0);
            d2 += e * e;
        }
        bool far_from_all = d2 > reach * reach;
        for (uint32_t b = 0; b < H.n_big && far_from_all; b++) {
            double q = 0.0;
            for (int k = 0; k < 3; k++) q += ((double)cam->origin[k] - H.big[b][k]) * ((double)cam->origin[k] - H.big[b][k]);
            far_from_all = fabs(sqrt(q) - (double)H.big[b][3]) > reach;
        }
        if (A.S.walk_start != 0u) A.S.primary_ref = far_from_all ? 1u : 0u;
        else if (far_from_all) exact = false;      // (staged in LDS there is one tree per launch: such a frame on the tree as handed over)
    }
    A.C.cam = *cam;
    A.C.width = p->width; A.C.height = p->height; A.C.spp = p->samples_per_pixel; A.C.max_depth = p->max_depth;
    A.C.seed = p->seed; A.C.integrator = p->integrator; A.C.background = p->background;
    A.C.bg[0] = p->background_color[0]; A.C.bg[1] = p->background_color[1]; A.C.bg[2] = p->background_color[2];
    A.out = d_out;
    A.tiles_x = g.tiles_x; A.tiles_y = g.tiles_y;
    const uint32_t tiles = g.tiles;
    A.tile_rank = g.rank; A.tile_world = g.world;
    A.n_local_tiles = g.n_local;
    A.n_chunks = choose_chunks(s, p);
#ifdef VK_WAVE_TIMES
    if (getenv("VK_WAVE_TIMES")) {
        if (!s->wave_times) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->wave_times), 3u * 1024u * 16u * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(s->wave_times, 0, 3u * 1024u * 16u * sizeof(unsigned long long), st));
        A.wave_times = s->wave_times;
    }
#endif
    A.counter = s->counter;
    A.clamped = reinterpret_cast<unsigned long long *>(s->counter) + 1;     // bytes 8..15 of the counter block
    A.launch_units = s->counter + 4;                                         // bytes 16..23: units pulled by each launch of a dual launch
    A.accum_clamp = accum_clamp_for(p->samples_per_pixel);
    A.shade_defer = SHADE_DEFER;          // (C5: 1 / 2 / 4 / 8 -> 573 / 588 / 593 / 603-at-pw-2)
    if (s->env.shade_defer >= 1 && s->env.shade_defer <= 64) A.shade_defer = (uint32_t)s->env.shade_defer;   // diagnostics
    // scenes beyond an XCD's L2 (C5: a leaf every 6 box steps, every gather a possible L2 miss): pending sphere tests are served
    // sooner — when 3x their lanes outnumber the stepping ones (1 / 2 / 3 -> 606 / 623 / 631 Msamples/s); L2-resident scenes: 1
    A.prim_weight = s->hot_bytes > (4u << 20) ? 3u : 1u;
    if (s->env.prim_weight >= 1 && s->env.prim_weight <= 64) A.prim_weight = (uint32_t)s->env.prim_weight;   // diagnostics
    size_t n_pixels = (size_t)p->width * p->height;
    if (stats) {
        stats->samples = partition_samples(p, g);
        stats->kernel_launches = 1;
        stats->scene_in_lds = s->lds_bytes ? 1u : 0u;
        stats->kernel_ms = 0.0; stats->seconds = 0.0;
    }
    if (want_debug) {
        size_t need = n_pixels * p->samples_per_pixel * sizeof(float4);
        int rc = ensure(s->debug, s->debug_bytes, need);
        if (rc != VK_OK) return rc;
        HIP_TRY(hipMemsetAsync(s->debug, 0, need, st));
        A.debug = s->debug;
    }
    HIP_TRY(hipEventRecord(s->ev0, st));
    if (p->max_depth == 0) {
        // ray_color returns (0,0,0) before tracing anything when depth (1) > MAX_DEPTH (main.rs:126-128): a black partition
        HIP_TRY(hipMemsetAsync(s->counter, 0, 32, st));
        int rc = tile_move<TM_ZERO_F32>(nullptr, d_out, p, g, st);
        if (rc != VK_OK) return rc;
        HIP_TRY(hipEventRecord(s->ev1, st));
        s->last_timed = true;
        s->redo_last = false; s->dual_last = false;      // (nothing was launched: no second launch, no unit split to judge)
        return VK_OK;
    }
    // Heavy-first tile order.  A launch ends when its slowest unit does, and tile costs are skewed (C2's glass tiles cost 8x
    // the mean, units that start mid-launch finish last).  So a probe launch of a few samples per pixel times every tile
    // (the COST build of the same kernel variant), three tiny kernels bucket-sort the tiles dearest first, and the real
    // launch takes its units in that order (longest processing time first).  The order never changes a pixel.
    // One rank's 1/8 share of C2: 78.1 -> 68.6 ms (ideal 64.9); whole frame 522 -> 519 ms including the probe.
    // (whole frames on one GPU gain nothing from it any more — there is no per-unit drain — and the 1 M-sphere scene loses 12 %
    // with its dearest tiles all in flight at once; one rank's 1/8 share of C2: 59.7 ms in raster order, 53.9 dearest first, ideal 50.0)
    bool use_order = A.n_local_tiles >= 64 && p->samples_per_pixel >= 64 && s->env.tile_order && (g.world > 1u || s->env.tile_order_forced);
    if (use_order) {
        int rc = ensure(s->tile_cost, s->tile_cost_n, (size_t)tiles * sizeof(uint32_t));
        if (rc == VK_OK) rc = ensure(s->tile_order, s->tile_order_n, (size_t)A.n_local_tiles * sizeof(uint32_t));
        if (rc != VK_OK) return rc;
        if (!s->order_hist) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->order_hist), ORDER_BUCKETS * sizeof(uint32_t)));
    }
    {   // order-independent pixel sums (vk_kernels.h to_fixed): zeroed per frame, resolved into d_out after the launch
        int rc = ensure(s->accum, s->accum_bytes, n_pixels * 3 * sizeof(long long));
        if (rc != VK_OK) return rc;
        HIP_TRY(hipMemsetAsync(s->accum, 0, n_pixels * 3 * sizeof(long long), st));
        A.accum = s->accum;
    }
    s->redo_last = false;
    const bool lds_scene = s->lds_bytes != 0;
    uint64_t per_region = 0;
    if (exact) {
        // queues for the samples the first launch drops: room for 1/32 of the partition's samples (C2 drops 0.05 %; 8 bytes each: 0.5 GB
        // for C2's 2.1 G samples), spread over REDO_REGIONS; a full queue is reported where the caller synchronises
        // (vk_scene_last_requeued_samples), vk_render then renders the frame again on the tree as handed over
        // (at most 256 MB: a frame that needs more overflows, and the fallback launch renders it on the tree as handed over)
        s->redo_last_samples = partition_samples(p, g);
        per_region = std::min<uint64_t>(s->redo_last_samples / 32u / REDO_REGIONS + 4096u, (256ull << 20) / sizeof(uint2) / REDO_REGIONS);
        // (the grid form seen from far away — the 1 M-sphere scene's camera — requeues 4 % of its samples: hits reported before the ray
        // enters their leaf's box, see segment_unsafe; room for an eighth, up to 4 GB of the 288)
        if (s->grid_on && !lds_scene) per_region = std::min<uint64_t>(s->redo_last_samples / 8u / REDO_REGIONS + 4096u, (4096ull << 20) / sizeof(uint2) / REDO_REGIONS);
        if (s->env.redo_region_cap >= 1) per_region = (uint64_t)s->env.redo_region_cap;      // tests
    }
    if (exact) {
        // (no memory for the queues: this frame on the tree as handed over, which needs none)
        if (ensure(s->redo_list, s->redo_bytes, (size_t)per_region * REDO_REGIONS * sizeof(uint2)) != VK_OK) { (void)hipGetLastError(); exact = false; }
    }
    if (exact) {
        HIP_TRY(hipMemsetAsync(s->redo_count, 0, (REDO_REGIONS * REDO_COUNT_STRIDE + 16) * sizeof(uint32_t), st));
        A.redo_list = s->redo_list; A.redo_count = s->redo_count; A.redo_plan = s->redo_count + REDO_REGIONS * REDO_COUNT_STRIDE;
        A.redo_region_cap = (uint32_t)per_region;
    } else if (s->exact) {
        A.S = s->ref_view;      // no second launch (diagnostic builds, a scene switched off, an oversized frame): the tree as handed over
    }
    if (s->grid_on && s->want_phase_stats && A.S.grid.nu != 0u) {
        // (the diagnostic builds have no grid walk: the tree as handed over, which is what items[] holds for a grid scene in global memory)
        A.S.grid.nu = 0u; A.S.walk_start = 0u; A.S.t_pad = 0.0f; A.S.gate_scale = 1.0f; A.S.tmin_gate = T_MIN; A.S.tie_rank = nullptr;
    }
    // LDS residency of the hot records
    bool lds = s->lds_bytes != 0;
    const uint32_t waves_per_wg = s->wg_threads / 64;
    size_t shmem = (size_t)waves_per_wg * per_wave_lds_bytes(pick_variant(s));
    if (lds) { A.lds_items = A.S.grid.nu != 0u ? s->grid_slots : A.S.n_items; A.lds_spheres = s->dev.n_spheres; A.lds_boxes = s->dev.n_boxes;
        shmem += s->lds_bytes; }
    // persistent grid: enough workgroups to fill the chip, never more than there are units
    s->dual_last = false;
    const uint64_t n_units = (uint64_t)A.n_local_tiles * A.n_chunks;
    // (choose_chunks keeps it below)
    if (n_units >= 0xFFFFFFFFull) return fail(VK_ERR_BAD_ARG, "tiles x sample chunks exceeds the 32-bit unit counter");
    uint32_t grid = (uint32_t)s->num_cus * s->wgs_per_cu;
    uint64_t need_wgs = (n_units + waves_per_wg - 1) / waves_per_wg;
    if (grid > need_wgs) grid = (uint32_t)need_wgs;
    if (grid < 1) grid = 1;

    int rc = VK_OK;
    uint32_t F = pick_variant(s) | (p->integrator == VK_INTEGRATOR_PDF ? (uint32_t)VKF_INTEG_PDF : 0u);
    // (VK_ORDER_REUSE=0 probes every frame)
    bool reuse_order = false;
    if (use_order && !s->want_phase_stats && s->env.order_reuse) {
        auto &o = s->order_for;
        auto close3 = [](const float *a, const float *b, float scale) {
            float d = fabsf(a[0] - b[0]) + fabsf(a[1] - b[1]) + fabsf(a[2] - b[2]);
            return d <= 0.05f * scale;
        };
        const float hscale = fabsf(cam->horizontal[0]) + fabsf(cam->horizontal[1]) + fabsf(cam->horizontal[2]) +
                             fabsf(cam->vertical[0]) + fabsf(cam->vertical[1]) + fabsf(cam->vertical[2]);      // size of the view plane
        reuse_order = o.valid && o.width == p->width && o.height == p->height && o.rank == g.rank && o.world == g.world &&
                      o.depth == p->max_depth && o.age < 16u && close3(o.org, cam->origin, hscale) &&
                      close3(o.llc, cam->lower_left_corner, hscale);
        if (reuse_order) { o.age++; A.tile_order = s->tile_order; }
    }
    if (use_order && !s->want_phase_stats && !reuse_order) {
        KArgs B = A;                                   // the probe: the same view at 1..4 samples per pixel, one unit per tile
        // (C2, one rank's 1/8 share: probe of 1 / 2 / 4 / 8 / 16 spp -> 68.6 / 69.0 / 69.8 / 70.5 / 73.0 ms: more samples cost more than
        // they sort better)
        B.C.spp = p->samples_per_pixel / 1024u; B.C.spp = B.C.spp < 1u ? 1u : (B.C.spp > 4u ? 4u : B.C.spp);
        // diagnostics
        if (s->env.probe_spp >= 1 && (uint32_t)s->env.probe_spp <= p->samples_per_pixel) B.C.spp = (uint32_t)s->env.probe_spp;
        // the probe only ranks the tiles: its paths are cut at 16 segments, so that this short launch does not end on a handful of
        // 50-segment paths (one rank's 1/8 share, efficiency against the whole frame / 8 with the cut at 50 / 16 / 8: C2 0.937 / 0.949 /
        // 0.948, C3 0.93 / 0.95 / 0.96, C5 0.957 / 0.957 / 0.951 — deep paths are part of what makes C5's tiles dear)
        {
            const uint32_t cut = s->env.probe_depth >= 1 ? (uint32_t)s->env.probe_depth : 16u;
            if (B.C.max_depth > cut) B.C.max_depth = cut;
        }
        B.n_chunks = 1; B.accum = nullptr; B.debug = nullptr; B.tile_order = nullptr;   // no sums: the probe only times the tiles
        B.redo_list = nullptr;                         // ... and drops nothing
        B.tile_cost = s->tile_cost;
        HIP_TRY(hipMemsetAsync(s->tile_cost, 0, (size_t)tiles * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(s->counter, 0, sizeof(uint32_t), st));
        uint32_t pgrid = (uint32_t)s->num_cus * s->wgs_per_cu, pneed = (A.n_local_tiles + waves_per_wg - 1) / waves_per_wg;
        if (pgrid > pneed) pgrid = pneed;
        rc = launch_by_features(s, F, B, lds, dim3(pgrid), shmem, st, true);
        if (rc != VK_OK) return rc;
        uint32_t nb = (A.n_local_tiles + 255u) / 256u;
        HIP_TRY(hipMemsetAsync(s->order_hist, 0, ORDER_BUCKETS * sizeof(uint32_t), st));
        hipLaunchKernelGGL(order_hist_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t *)s->tile_cost, A.n_local_tiles, A.tile_rank,
            A.tile_world, s->order_hist);
        hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1), 0, st, s->order_hist);
        hipLaunchKernelGGL(order_scatter_kernel, dim3(nb), dim3(256), 0, st, s->tile_cost, A.n_local_tiles, A.tile_rank, A.tile_world,
            s->order_hist, s->tile_order);
        HIP_TRY(hipGetLastError());
        A.tile_order = s->tile_order;
        auto &o = s->order_for;
        o.valid = true; o.width = p->width; o.height = p->height; o.rank = g.rank; o.world = g.world; o.depth = p->max_depth; o.age = 0;
        for (int k = 0; k < 3; k++) { o.org[k] = cam->origin[k]; o.llc[k] = cam->lower_left_corner[k]; }
    }
    HIP_TRY(hipMemsetAsync(s->counter, 0, 32, st));       // work counter, this frame's clamped-sample count, per-launch unit counts
#ifdef VK_DEBUG_LIB
    if (s->want_phase_stats) {
        const uint32_t FULLPDF = VKF_ALL_SCENE | VKF_INTEG_PDF;
        const uint32_t CORNELLPDF = VKF_RECT | VKF_LIST | VKF_INSTANCE | VKF_BOX | VKF_INTEG_PDF;
        if (F != 0u && F != FULLPDF && F != CORNELLPDF)
            return fail(VK_ERR_UNSUPPORTED, "phase statistics are only built for the sphere-only/scatter, the Cornell-type/PDF and the full/PDF variants");
        if (!s->phase_stats) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->phase_stats), 24 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(s->phase_stats, 0, 24 * sizeof(unsigned long long), st));
        A.phase_stats = s->phase_stats;
        auto go = [&](auto kernel) -> int {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(s->wg_threads), shmem, st, A);
            return VK_OK;
        };
        if (F == 0u) rc = lds ? go(&render_kernel<0u, true, 6, true>) : go(&render_kernel<0u, false, 6, true>);
        else if (F == CORNELLPDF) rc = lds ? go(&render_kernel<CORNELLPDF, true, 6, true>) : go(&render_kernel<CORNELLPDF, false, 6, true>);
        else rc = lds ? go(&render_kernel<FULLPDF, true, 4, true>) : go(&render_kernel<FULLPDF, false, 4, true>);
        if (rc != VK_OK) return rc;
        HIP_TRY(hipGetLastError());
        F = 0xFFFFFFFFu;   // launched
    }
#endif
    if (F != 0xFFFFFFFFu) {
        // enough units for 28 waves per CU to stay busy: the dual launch; else (tiny frames) the single 2 x 768-thread shape
        const bool dual = s->dual_launch && lds && n_units >= (uint64_t)s->num_cus * 28u * 4u && s->stream2;
        if (dual) {
            // seven waves per SIMD hide more of a parked lane's wait: shading deferred 5x, pending sphere tests served at 2x weight
            // (C2 at 256 spp, (defer, weight): (4,1) 6 098, (5,1) 6 166, (5,2) 6 230, (6,2) 6 208, (8,2) 6 230, (5,3) 6 071 Msamples/s)
            // (on the rebuilt tree of exact re-treeing, 256 spp: (4,1) 7 295, (5,2) 7 395, (6,2) 7 445, (8,2) 7 435, (6,3) 7 444)
            if (!(s->env.shade_defer >= 1 && s->env.shade_defer <= 64)) A.shade_defer = 6u;
            if (!(s->env.prim_weight >= 1 && s->env.prim_weight <= 64)) A.prim_weight = 2u;
        }
        const bool gridw = A.S.grid.nu != 0u;
        if (dual && F == 0u) rc = gridw ? launch_dual<0u, true>(s, A, per_wave_lds_bytes(0u), st) : launch_dual<0u>(s, A, per_wave_lds_bytes(0u), st);
        else if (dual && F == (uint32_t)VKF_INTEG_PDF) rc = launch_dual<VKF_INTEG_PDF>(s, A, per_wave_lds_bytes(0u), st);
        else rc = launch_by_features(s, F, A, lds, dim3(grid), shmem, st, false);
    }
    if (rc != VK_OK) return rc;
    if (exact) {
        // the second launch: the queued samples on the scene as handed over, in the single-launch shape
        uint32_t *plan = s->redo_count + REDO_REGIONS * REDO_COUNT_STRIDE;
        hipLaunchKernelGGL(redo_plan_kernel, dim3(1), dim3(REDO_REGIONS), 0, st, (const uint32_t *)s->redo_count, A.redo_region_cap, plan,
            (uint32_t)s->num_cus * s->wgs_per_cu * waves_per_wg);
        HIP_TRY(hipMemsetAsync(s->counter, 0, sizeof(uint32_t), st));      // the unit counter only: clamped samples and unit counts add up
        KArgs B = A;
        B.S = s->ref_view; B.list_mode = 1u; B.tile_order = nullptr; B.wave_times = nullptr;
        if (lds) B.lds_items = B.S.n_items;
        B.shade_defer = SHADE_DEFER; B.prim_weight = s->hot_bytes > (4u << 20) ? 3u : 1u;
        rc = launch_by_features(s, F, B, lds, dim3((uint32_t)s->num_cus * s->wgs_per_cu), shmem, st, false);
        if (rc != VK_OK) return rc;
        // ... and the fallback behind it: should a queue have overflowed, the sums are cleared and the partition is rendered on the tree
        // as handed over, so that a frame is never incomplete whoever the caller is (nearly always: two launches that return at once)
        hipLaunchKernelGGL(redo_reset_kernel, dim3(1024), dim3(256), 0, st, (const uint32_t *)plan, reinterpret_cast<unsigned long long *>(s->accum),
            n_pixels * 3, s->counter);
        KArgs Fb = A;
        Fb.S = s->ref_view; Fb.list_mode = 2u; Fb.redo_list = nullptr; Fb.wave_times = nullptr;
        if (lds) Fb.lds_items = Fb.S.n_items;
        Fb.shade_defer = SHADE_DEFER; Fb.prim_weight = B.prim_weight;
        rc = launch_by_features(s, F, Fb, lds, dim3(grid), shmem, st, false);
        if (rc != VK_OK) return rc;
        // the plan travels to the slot the previous frame did not use — unless that slot's verdict is still in flight (two frames behind and
        // not finished: the caller runs far ahead), then this frame goes unjudged rather than overwriting it
        const int b = s->plan_last ^ 1;
        s->redo_last = true;
        if (!s->plan_pending[b]) {
            HIP_TRY(hipMemcpyAsync(s->plan_host + 4 * b, plan, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(s->ev_plan[b], st));
            s->plan_pending[b] = true; s->plan_samples[b] = s->redo_last_samples; s->plan_last = b;
            s->plan_copied = true;
        } else s->plan_copied = false;
    }
    {
        uint32_t blocks = (uint32_t)((n_pixels + 255) / 256);
        hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, st, (const long long *)A.accum, d_out, p->width, p->height,
                           p->samples_per_pixel, A.tiles_x, A.tile_rank, A.tile_world);
        HIP_TRY(hipGetLastError());
        if (stats) stats->kernel_launches = 2;
    }
    HIP_TRY(hipEventRecord(s->ev1, st));
    s->last_timed = true;
    return VK_OK;
}

// One device: render (f32) and, for RGB8 output, the fused output stage of this partition.
int enqueue_render_single(vk_scene *s, const vk_camera *cam, const vk_render_params *p, void *d_out, hipStream_t st, bool want_debug,
    vk_stats *stats) {
    if (p->output_format == VK_OUTPUT_F32) return enqueue_render_f32(s, cam, p, reinterpret_cast<float *>(d_out), st, want_debug, stats);
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure(s->fb, s->fb_bytes, (size_t)p->width * p->height * 3 * sizeof(float));
    if (rc != VK_OK) return rc;
    rc = enqueue_render_f32(s, cam, p, s->fb, st, want_debug, stats);
    if (rc != VK_OK) return rc;
    return tile_move<TM_CONVERT_U8>(s->fb, d_out, p, TileGeom(p), st);
}

// Multi-device group: part j renders the tiles {t : t = R + W*(j + n*i)} of this call's partition (R of W) on its
// own device and stream, packs them into a slab (RGB8: through to_color, 4x smaller), the slab travels to devices[0] with
// ONE peer copy (xGMI), and devices[0] scatters the slabs into the caller's image on the caller's stream.
int enqueue_render_multi(vk_scene *grp, const vk_camera *cam, const vk_render_params *p, void *d_out, hipStream_t st0, vk_stats *stats) {
    const uint32_t n = (uint32_t)grp->parts.size();
    const TileGeom g(p);
    const bool u8 = p->output_format == VK_OUTPUT_RGB8;
    const size_t slot_bytes = u8 ? 3 : 12;
    HIP_TRY(hipSetDevice(grp->device));
    HIP_TRY(hipEventRecord(grp->ev_begin, st0));      // the parts start after whatever the caller's stream held before this frame
    uint64_t samples = 0; uint32_t launches = 0;
    std::vector<TileGeom> geoms;
    std::vector<size_t> slab_size;
    for (uint32_t j = 0; j < n; j++) {
        vk_scene *q = grp->parts[j];
        vk_render_params pj = *p;
        pj.tile_rank = g.rank + g.world * j; pj.tile_world = g.world * n; pj.output_format = VK_OUTPUT_F32;
        const TileGeom gj(&pj);
        geoms.push_back(gj);
        HIP_TRY(hipSetDevice(q->device));
        HIP_TRY(hipStreamWaitEvent(q->stream, grp->ev_begin, 0));
        int rc = ensure(q->fb, q->fb_bytes, (size_t)p->width * p->height * 3 * sizeof(float));
        if (rc != VK_OK) return rc;
        vk_stats sj;
        memset(&sj, 0, sizeof(sj));
        rc = enqueue_render_f32(q, cam, &pj, q->fb, q->stream, false, &sj);
        if (rc != VK_OK) return rc;
        samples += sj.samples; launches += sj.kernel_launches;
        size_t bytes = (size_t)gj.n_local * 64u * slot_bytes;
        rc = ensure(q->slab, q->slab_bytes, bytes);
        if (rc != VK_OK) return rc;
        rc = u8 ? tile_move<TM_PACK_U8>(q->fb, q->slab, &pj, gj, q->stream) : tile_move<TM_PACK_F32>(q->fb, q->slab, &pj, gj, q->stream);
        if (rc != VK_OK) return rc;
        if (bytes > q->landing_bytes) {                // the landing buffer lives on devices[0]
            HIP_TRY(hipSetDevice(grp->device));
            rc = ensure(q->landing, q->landing_bytes, bytes);
            if (rc != VK_OK) return rc;
            HIP_TRY(hipSetDevice(q->device));
        }
        slab_size.push_back(bytes);
        if (grp->comms.empty()) {
            if (bytes) HIP_TRY(hipMemcpyPeerAsync(q->landing, grp->device, q->slab, q->device, bytes, q->stream));
            HIP_TRY(hipEventRecord(q->ev_landed, q->stream));
        }
    }
    if (!grp->comms.empty()) {
        // The same exchange as ONE RCCL group: part j sends its slab on its own stream (behind its render and pack), devices[0] receives
        // the slabs on the group's receive stream, which waits for nothing but the start of the frame (the previous frame's unpack
        // kernels have read the landing buffers by then) — so the transfers overlap devices[0]'s own render.  ncclSend / ncclRecv pairs
        // inside one ncclGroupStart / ncclGroupEnd progress together; the receives complete in that stream's order, so ONE event says
        // that every slab has landed.  Part 0's slab is on devices[0] already and is unpacked where it lies.
        const RcclApi &R = rccl_api();
        vk_scene *q0 = grp->parts[0];
        HIP_TRY(hipSetDevice(q0->device));
        HIP_TRY(hipEventRecord(q0->ev_landed, q0->stream));
        HIP_TRY(hipStreamWaitEvent(grp->stream, grp->ev_begin, 0));
        RCCL_TRY(R.GroupStart());
        for (uint32_t j = 1; j < n; j++) {
            vk_scene *q = grp->parts[j];
            if (!slab_size[j]) continue;
            RCCL_TRY(R.Send(q->slab, slab_size[j], ncclUint8, 0, grp->comms[j], q->stream));
            RCCL_TRY(R.Recv(q->landing, slab_size[j], ncclUint8, (int)j, grp->comms[0], grp->stream));
        }
        RCCL_TRY(R.GroupEnd());
        HIP_TRY(hipSetDevice(grp->device));
        HIP_TRY(hipEventRecord(grp->ev_landed, grp->stream));
    }
    HIP_TRY(hipSetDevice(grp->device));
    for (uint32_t j = 0; j < n; j++) {
        vk_scene *q = grp->parts[j];
        vk_render_params pj = *p;
        pj.tile_rank = geoms[j].rank; pj.tile_world = geoms[j].world;
        const bool rccl = !grp->comms.empty();
        HIP_TRY(hipStreamWaitEvent(st0, (rccl && j != 0u ? grp : q)->ev_landed, 0));
        const void *from = rccl && j == 0u ? q->slab : q->landing;
        int rc = u8 ? tile_move<TM_UNPACK_U8>(from, d_out, &pj, geoms[j], st0) : tile_move<TM_UNPACK_F32>(from, d_out, &pj, geoms[j], st0);
        if (rc != VK_OK) return rc;
    }
    grp->last_timed = true;
    if (stats) {
        stats->samples = samples; stats->kernel_launches = launches; stats->scene_in_lds = grp->parts[0]->lds_bytes ? 1u : 0u;
        stats->kernel_ms = 0.0; stats->seconds = 0.0;
    }
    return VK_OK;
}

int enqueue_render(vk_scene *s, const vk_camera *cam, const vk_render_params *p, void *d_out, hipStream_t st, bool want_debug,
    vk_stats *stats) {
    int rc = check_render_args(s, cam, p);
    if (rc != VK_OK) return rc;
    if (!s->parts.empty()) {
        if (want_debug) return fail(VK_ERR_UNSUPPORTED, "per-sample debug output is single-device only");
        return enqueue_render_multi(s, cam, p, d_out, st, stats);
    }
    return enqueue_render_single(s, cam, p, d_out, st, want_debug, stats);
}

void destroy_one(vk_scene *s) {
    if (!s) return;
    for (ncclComm_t c : s->comms) if (c) (void)rccl_api().CommDestroy(c);
    for (vk_scene *q : s->parts) destroy_one(q);
    (void)hipSetDevice(s->device);
    for (void *p : s->allocs) (void)hipFree(p);
    for (void *p : {(void *)s->counter, (void *)s->fb, (void *)s->fb8, (void *)s->accum, (void *)s->debug, (void *)s->phase_stats,
        (void *)s->tile_cost,
                    (void *)s->tile_order, (void *)s->order_hist, (void *)s->slab, (void *)s->redo_list, (void *)s->redo_count})
        if (p) (void)hipFree(p);
    if (s->plan_host) (void)hipHostFree(s->plan_host);
    if (s->landing) { (void)hipSetDevice(s->landing_device); (void)hipFree(s->landing); (void)hipSetDevice(s->device); }
    for (hipEvent_t e : {s->ev0, s->ev1, s->ev_landed, s->ev_begin, s->ev_fork, s->ev_join, s->ev_plan[0], s->ev_plan[1]})
        if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->stream2) (void)hipStreamDestroy(s->stream2);
    delete s;
}

struct SceneDeleter { void operator()(vk_scene *s) const { destroy_one(s); } };
using ScenePtr = std::unique_ptr<vk_scene, SceneDeleter>;

int check_device(int device, hipDeviceProp_t &pr) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(VK_ERR_NO_DEVICE,
        "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(VK_ERR_BAD_ARG, "device index out of range");
    HIP_TRY(hipGetDeviceProperties(&pr, device));
    if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) return fail(VK_ERR_NO_DEVICE,
        std::string("device is ") + pr.gcnArchName + ", this build targets gfx950 only");
    return VK_OK;
}

// uploads an already linearised scene to one device
int create_on_device(const std::shared_ptr<const LinearScene> &host, int device, bool own_stream, ScenePtr &out) {
    hipDeviceProp_t pr;
    int rc = check_device(device, pr);
    if (rc != VK_OK) return rc;
    ScenePtr s(new vk_scene);
    s->device = device;
    s->host = host;
    s->env = EnvSwitches::read();
    HIP_TRY(hipSetDevice(device));
    s->num_cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    const LinearScene &H = *host;
    DScene &D = s->dev;
    memset(&D, 0, sizeof(D));
#define UP(vec, field) do { rc = upload(s.get(), H.vec, D.field); if (rc != VK_OK) return rc; } while (0)
    // LDS residency: items + spheres + accumulators must leave room for >= 2 workgroups per CU
    // (exact re-treeing: the second launch stages the tree as handed over instead of the rebuilt one, whichever is larger counts)
    size_t hot = std::max(H.items.size(), H.ref_items.size()) * sizeof(DItem) + H.spheres.size() * sizeof(DSphere) +
                 H.boxes.size() * sizeof(DBox);
    // the grid form (DGrid): the first launch stages the table [cells | refs] instead of a tree, the second one the tree as handed over
    bool grid = H.grid.nu != 0u && !H.ref_items.empty() && !s->env.no_grid;
    const size_t grid_table_bytes = grid ? ((H.grid_cells.size() + H.grid_refs.size()) * sizeof(uint32_t) + 31u) / 32u * 32u : 0u;
    // (the probe and the diagnostic builds walk the rebuilt TREE, which therefore counts too where the scene is staged in LDS)
    if (grid) hot = std::max(grid_table_bytes, std::max(H.ref_items.size(), H.items.size()) * sizeof(DItem)) + H.spheres.size() * sizeof(DSphere);
    s->grid_on = grid;
    s->hot_bytes = hot;
    plan_residency(s.get(), hot);
    if (grid && s->lds_bytes == 0 && !s->env.grid_global) {
        // The grid form is for scenes staged in LDS.  From global memory every visited cell is three dependent cache misses (cell ->
        // references -> sphere) once the tables outgrow an XCD's L2, where a tree's top levels stay hot; and a large layer seen from far
        // away needs wide bands of cells around its primary rays (the dilation grows with the distance from the origin: 1.4 per 1 000).
        // Measured: 1 M spheres 500 against the near form's 1 280 Msamples/s, 160 000 spheres 850 against 1 290 (40 000: 1 930 against
        // 1 400, a win while everything fits L2: profiles/r05/experiments/README.md).
        grid = false;
        hot = std::max(H.items.size(), H.ref_items.size()) * sizeof(DItem) + H.spheres.size() * sizeof(DSphere) + H.boxes.size() * sizeof(DBox);
        s->grid_on = false; s->hot_bytes = hot;
        plan_residency(s.get(), hot);
    }
    D.gate_scale = 1.0f; D.tmin_gate = T_MIN;
    if (!H.ref_items.empty()) {
        // Exact re-treeing (vk_trace.h).  Staged in LDS: the rebuilt tree is the scene's, the tree as handed over serves the second launch
        // (`exact`).  Traversed from global memory: both trees in one array, early segments are walked again in place (DScene::walk_start).
        const DScene hv = H.host_view();
        D.t_pad = hv.t_pad; D.gate_scale = hv.gate_scale; D.tmin_gate = hv.tmin_gate;
        for (int k = 0; k < 3; k++) { D.trust_c0[k] = hv.trust_c0[k]; D.small_clo[k] = hv.small_clo[k]; D.small_chi[k] = hv.small_chi[k]; }
        D.trust_r0sq = hv.trust_r0sq;
        D.reach = hv.reach; D.clear_k = hv.clear_k; D.clear_r2 = hv.clear_r2; D.clear_slack = hv.clear_slack;
        if (grid) {
            // no ball, no reach: the grid's walk tests every sphere that can hold a candidate, wherever the ray starts
            use_grid(D);
            std::vector<uint32_t> table(grid_table_bytes / sizeof(uint32_t), 0u);
            std::copy(H.grid_cells.begin(), H.grid_cells.end(), table.begin());
            std::copy(H.grid_refs.begin(), H.grid_refs.end(), table.begin() + H.grid_cells.size());
            rc = upload(s.get(), table, D.grid_cells);
            if (rc != VK_OK) return rc;
            D.grid_refs = D.grid_cells + H.grid_cells.size();
            D.grid = H.grid;
            s->grid_slots = (uint32_t)(grid_table_bytes / 32u);
        }
        if (!H.unit_item.empty() && H.proven) { rc = upload(s.get(), H.unit_item, D.unit_item); if (rc != VK_OK) return rc; }
        if (grid && s->lds_bytes == 0) {
            // from global memory: the first launch reads the table and the spheres only; the tree as handed over serves the second one
            // (and the diagnostic builds)
            UP(ref_items, ref_items);
            D.items = D.ref_items; D.n_ref_items = hv.n_ref_items; D.n_items = D.n_ref_items; D.n_world_items = D.n_ref_items;
            D.unit_tree = D.ref_items;
            s->exact = true;
        } else
        if (s->lds_bytes != 0) {
            UP(items, items); UP(ref_items, ref_items);
            D.n_ref_items = hv.n_ref_items; D.n_items = (uint32_t)H.items.size(); D.n_world_items = H.world_items;
            D.unit_tree = D.ref_items;
            s->exact = true;
        } else {
            uint32_t walk_start = 0;
            const std::vector<DItem> both = H.combined_items(walk_start);
            rc = upload(s.get(), both, D.items);
            if (rc != VK_OK) return rc;
            D.n_items = (uint32_t)both.size(); D.n_world_items = (uint32_t)both.size(); D.walk_start = walk_start;
            D.unit_tree = D.items;       // (the tree as handed over comes first, item for item)
        }
    } else {
        UP(items, items);
        D.n_items = (uint32_t)H.items.size(); D.n_world_items = H.world_items;
    }
    UP(spheres, spheres); UP(sphere_mat, sphere_mat); UP(moving, moving); UP(rects, rects); UP(boxes, boxes);
    UP(lists, lists); UP(list_refs, list_refs); UP(media, media); UP(instances, instances);
    UP(materials, materials); UP(sphere_material, sphere_material); UP(textures, textures); UP(images, images);
    UP(image_bytes, image_bytes);
    UP(perlins, perlins); UP(lights, lights);
    if (!H.tie_rank.empty()) UP(tie_rank, tie_rank);
#undef UP
    D.tie_base_rect = H.tie_base_rect; D.tie_base_box = H.tie_base_box; D.tie_base_list = H.tie_base_list;
    D.fast_div = H.host_view().fast_div;
    D.n_noise_spheres = H.n_noise_spheres;
    for (int k = 0; k < 4; k++) { D.noise_sphere[k] = H.noise_sphere[k]; D.noise_tex[k] = H.noise_tex[k];
        D.noise_perlin[k] = H.noise_perlin[k]; }
    D.n_spheres = (uint32_t)H.spheres.size();
    D.n_lights = (uint32_t)H.lights.size(); D.features = H.features; D.n_boxes = (uint32_t)H.boxes.size();
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->counter), 256));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));
    if (own_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_landed, hipEventDisableTiming));
    }
    if (s->exact) {
        // the scene as handed over (tests/emu/emu.cpp reference_view is the same thing on the host)
        s->ref_view = D;
        s->ref_view.items = D.ref_items; s->ref_view.n_items = D.n_ref_items; s->ref_view.n_world_items = D.n_ref_items;
        s->ref_view.ref_items = nullptr; s->ref_view.n_ref_items = 0; s->ref_view.t_pad = 0.0f; s->ref_view.gate_scale = 1.0f;
        s->ref_view.tmin_gate = T_MIN; s->ref_view.tie_rank = nullptr; s->ref_view.grid.nu = 0u;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->redo_count), (REDO_REGIONS * REDO_COUNT_STRIDE + 16) * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s->plan_host), 8 * sizeof(uint32_t), hipHostMallocDefault));
        memset(s->plan_host, 0, 8 * sizeof(uint32_t));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_plan[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_plan[1], hipEventDisableTiming));
    }
    if (s->dual_launch) {
        HIP_TRY(hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
    out = std::move(s);
    return VK_OK;
}

int linearize_desc(const vk_scene_desc *desc, std::shared_ptr<const LinearScene> &out) {
    auto h = std::make_shared<LinearScene>();
    std::string err;
    LinearizeOptions opt;
    opt.retree = EnvSwitches::read().retree;
    // VK_GATE_PROOF=0: among the trees the description allows, prefer the empirical form (comparisons, the constructed counter-example on
    // the device).  It does NOT allow an unproven tree by itself: that takes VK_SCENE_EMPIRICAL_TREES in the description.
    if (const char *e = getenv("VK_GATE_PROOF")) opt.want_proof = e[0] != '0';
#ifdef VK_DEBUG_LIB
    // unsound test switches (LinearizeOptions): the debug build only; the product's trees are the proven one, the handed-over one, or
    // what the description's flags opt into
    if (const char *e = getenv("VK_GATE_GROW")) opt.gate_grow = e[0] != '0';
    if (const char *e = getenv("VK_T_PAD")) opt.t_pad = (float)atof(e);
    if (const char *e = getenv("VK_EMPIRICAL_TREES")) opt.allow_empirical = e[0] == '1';
    if (const char *e = getenv("VK_NEAR_FORM")) opt.near_form = e[0] != '0';
    if (const char *e = getenv("VK_UNIT_FORM")) opt.unit_form = e[0] != '0';
    if (const char *e = getenv("VK_NEAR_FIRST")) opt.near_first = e[0] != '0';
#endif
    int rc = linearize(desc, *h, err, opt);
    if (rc != VK_OK) return fail(rc, err);
    out = h;
    return VK_OK;
}

}  // namespace

extern "C" {

int vk_abi_version(void) { return VK_ABI_VERSION; }

int vk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, i) == hipSuccess && strncmp(pr.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

const char *vk_last_error(void) { return g_err.c_str(); }

int vk_gather_backends(void) { return 1 | (rccl_api().ok() ? 2 : 0); }

int vk_scene_create(const vk_scene_desc *desc, int device, vk_scene **out) {
    if (!out) return fail(VK_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    return guarded([&]() -> int {
        hipDeviceProp_t pr;
        int rc = check_device(device, pr);           // before the (possibly long) linearisation
        if (rc != VK_OK) return rc;
        std::shared_ptr<const LinearScene> host;
        rc = linearize_desc(desc, host);
        if (rc != VK_OK) return rc;
        ScenePtr s;
        rc = create_on_device(host, device, false, s);
        if (rc != VK_OK) return rc;
        *out = s.release();
        return VK_OK;
    });
}

int vk_scene_create_multi(const vk_scene_desc *desc, const int *devices, int n_devices, vk_scene **out) {
    if (!out) return fail(VK_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return fail(VK_ERR_BAD_ARG, "devices: need 1..64 entries");
    return guarded([&]() -> int {
        hipDeviceProp_t pr;
        for (int j = 0; j < n_devices; j++) { int rc = check_device(devices[j], pr); if (rc != VK_OK) return rc; }
        std::shared_ptr<const LinearScene> host;
        int rc = linearize_desc(desc, host);
        if (rc != VK_OK) return rc;
        ScenePtr grp(new vk_scene);
        grp->device = devices[0];
        grp->host = host;
        grp->env = EnvSwitches::read();
        memset(&grp->dev, 0, sizeof(grp->dev));
        for (int j = 0; j < n_devices; j++) {
            ScenePtr part;
            rc = create_on_device(host, devices[j], true, part);
            if (rc != VK_OK) return rc;
            part->landing_device = devices[0];
            if (devices[j] != devices[0]) {          // let devices[0] and this device address each other's memory (xGMI peer copies)
                int can = 0;
                HIP_TRY(hipDeviceCanAccessPeer(&can, devices[j], devices[0]));
                if (!can)
                    fprintf(stderr, "vecchio_amd: device %d cannot address device %d's memory (no peer access): its tile slab travels "
                        "through host memory\n", devices[j], devices[0]);
                if (can) {
                    HIP_TRY(hipSetDevice(devices[j]));
                    hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(VK_ERR_HIP,
                        std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                    (void)hipGetLastError();
                }
            }
            grp->parts.push_back(part.release());
        }
        const bool want_rccl = (desc->flags & VK_SCENE_RCCL_GATHER) != 0u || (getenv("VK_MULTI_GATHER") && !strcmp(getenv("VK_MULTI_GATHER"), "rccl"));
        if (want_rccl) {
            // one communicator rank per DEVICE: a device listed twice (the one-GPU test shape) cannot take part
            bool distinct = true;
            for (int a = 0; a < n_devices; a++) for (int b = a + 1; b < n_devices; b++) distinct = distinct && devices[a] != devices[b];
#ifdef VK_DEBUG_LIB
            if (getenv("VK_RCCL_ALLOW_DUPLICATE_DEVICES")) distinct = true;      // (with the test double of VK_RCCL_LIB only: real RCCL refuses)
#endif
            const RcclApi &R = rccl_api();
            if (!R.ok()) fprintf(stderr, "vecchio_amd: VK_SCENE_RCCL_GATHER: %s; the tile slabs travel by peer copies\n", R.why.c_str());
            else if (!distinct) fprintf(stderr, "vecchio_amd: VK_SCENE_RCCL_GATHER: a device is listed more than once (one communicator rank per "
                "device); the tile slabs travel by peer copies\n");
            else {
                grp->comms.assign((size_t)n_devices, nullptr);
                ncclResult_t r = R.CommInitAll(grp->comms.data(), n_devices, devices);
                if (r != ncclSuccess) {
                    fprintf(stderr, "vecchio_amd: ncclCommInitAll over %d devices failed (%s); the tile slabs travel by peer copies\n", n_devices,
                        R.GetErrorString(r));
                    grp->comms.clear();
                }
            }
            if (!grp->comms.empty()) {       // the receive stream of devices[0] and the event behind the frame's last receive
                HIP_TRY(hipSetDevice(devices[0]));
                HIP_TRY(hipStreamCreateWithFlags(&grp->stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&grp->ev_landed, hipEventDisableTiming));
            }
        }
        HIP_TRY(hipSetDevice(devices[0]));
        HIP_TRY(hipEventCreateWithFlags(&grp->ev_begin, hipEventDisableTiming));
        grp->lds_bytes = grp->parts[0]->lds_bytes; grp->hot_bytes = grp->parts[0]->hot_bytes;
        *out = grp.release();
        return VK_OK;
    });
}

void vk_scene_destroy(vk_scene *s) { destroy_one(s); }

int vk_scene_get_info(const vk_scene *s, vk_scene_info *out) {
    if (!s || !out) return fail(VK_ERR_BAD_ARG, "null argument");
    const LinearScene &H = *s->host;
    const vk_scene *one = s->parts.empty() ? s : s->parts[0];
    out->n_items = (uint32_t)H.items.size();
    out->n_prims = H.n_prims;
    out->n_instances = (uint32_t)H.instances.size();
    uint64_t b = 0;
    b += H.items.size() * sizeof(DItem) + H.spheres.size() * (sizeof(DSphere) + 4) + H.moving.size() * sizeof(DMoving) +
         H.rects.size() * sizeof(DRect) + H.lists.size() * sizeof(DList) + H.list_refs.size() * 4 +
         H.media.size() * sizeof(DMedium) + H.instances.size() * sizeof(DInstance) + H.materials.size() * sizeof(DMaterial) +
         H.textures.size() * sizeof(DTexture) + H.image_bytes.size() + H.perlins.size() * sizeof(DPerlin);
    out->device_bytes = b;
    out->lds_bytes = one->lds_bytes;
    out->features = pick_variant(one);
    out->tree = !H.ref_items.empty() ? (one->grid_on ? VK_TREE_REBUILT_GRID : H.near_form ? VK_TREE_REBUILT_NEAR : (H.proven ? VK_TREE_REBUILT_PROVEN : VK_TREE_REBUILT_EMPIRICAL))
                                     : (!H.tie_rank.empty() ? VK_TREE_REBUILT_FAST : VK_TREE_HANDED_OVER);
    out->gather = s->parts.empty() ? VK_GATHER_NONE : (s->comms.empty() ? VK_GATHER_PEER_COPY : VK_GATHER_RCCL);
    out->tree_suspended_frames = 0;
    for (const vk_scene *q : (s->parts.empty() ? std::vector<vk_scene *>{const_cast<vk_scene *>(s)} : s->parts))
        if (q->exact_resume > q->frame_no + 1u) out->tree_suspended_frames = std::max<uint32_t>(out->tree_suspended_frames,
            (uint32_t)(q->exact_resume - q->frame_no - 1u));
    return VK_OK;
}

int vk_render_device(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, void *d_rgb_out, void *hip_stream,
    vk_stats *stats_out) {
    if (!d_rgb_out) return fail(VK_ERR_BAD_ARG, "null device framebuffer");
    return guarded([&]() -> int { return enqueue_render(scene, cam, params, d_rgb_out, reinterpret_cast<hipStream_t>(hip_stream), false,
        stats_out); });
}

// HIP-event time (ms) of the launches enqueued by the last vk_render_device / vk_render on
// this scene; synchronises on their end event.  Multi-device: the slowest part.
int vk_scene_part_info(vk_scene *s, int part, vk_part_info *out) {
    if (!s || !out) return fail(VK_ERR_BAD_ARG, "null argument");
    const int n = s->parts.empty() ? 1 : (int)s->parts.size();
    if (part < 0 || part >= n) return fail(VK_ERR_BAD_ARG, "part index out of range");
    vk_scene *q = s->parts.empty() ? s : s->parts[(size_t)part];
    memset(out, 0, sizeof(*out));
    out->n_parts = (uint32_t)n;
    out->device = q->device;
    hipDeviceProp_t pr;
    HIP_TRY(hipGetDeviceProperties(&pr, q->device));
    snprintf(out->name, sizeof(out->name), "%s", pr.name);
    if (hipDeviceGetPCIBusId(out->pci_bus_id, (int)sizeof(out->pci_bus_id), q->device) != hipSuccess) { (void)hipGetLastError(); out->pci_bus_id[0] = 0; }
    const int landing = s->parts.empty() ? q->device : s->device;
    int can = 1;
    if (q->device != landing) HIP_TRY(hipDeviceCanAccessPeer(&can, q->device, landing));
    out->can_access_landing_device = (uint32_t)can;
    out->kernel_ms = -1.0;
    if (q->last_timed) {
        double ms = 0.0;
        int rc = vk_scene_last_kernel_ms(q, &ms);
        if (rc != VK_OK) return rc;
        out->kernel_ms = ms;
    }
    return VK_OK;
}

int vk_scene_last_kernel_ms(vk_scene *s, double *ms_out) {
    if (!s || !ms_out) return fail(VK_ERR_BAD_ARG, "null argument");
    if (!s->last_timed) return fail(VK_ERR_BAD_ARG, "no render enqueued yet");
    if (!s->parts.empty()) {
        double worst = 0.0;
        for (vk_scene *q : s->parts) {
            double ms = 0.0;
            int rc = vk_scene_last_kernel_ms(q, &ms);
            if (rc != VK_OK) return rc;
            if (ms > worst) worst = ms;
        }
        *ms_out = worst;
        return VK_OK;
    }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev1));
    if (s->wave_times && s->parts.empty()) {      // diagnostics: when the waves of the last render's FIRST launch started, pulled their last unit and ended
        std::vector<unsigned long long> w(3u * 1024u * 16u);
        HIP_TRY(hipMemcpy(w.data(), s->wave_times, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0; std::vector<double> ends, lasts;
        for (size_t k = 0; k < w.size(); k += 3) if (w[k]) { t0 = std::min(t0, w[k]); t1 = std::max(t1, w[k + 2]); }
        for (size_t k = 0; k < w.size(); k += 3) if (w[k]) { ends.push_back((double)(w[k + 2] - t0) * 1e-5); lasts.push_back((double)(w[k + 1] - t0) * 1e-5); }
        std::sort(ends.begin(), ends.end()); std::sort(lasts.begin(), lasts.end());
        if (!ends.empty()) {
            auto q = [&](const std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
            fprintf(stderr, "vecchio_amd: %zu waves, span %.2f ms; wave ends (ms): 1%% %.2f 10%% %.2f 50%% %.2f 90%% %.2f 99%% %.2f max %.2f; last unit pull: 50%% %.2f 99%% %.2f max %.2f\n",
                ends.size(), (double)(t1 - t0) * 1e-5, q(ends, 0.01), q(ends, 0.1), q(ends, 0.5), q(ends, 0.9), q(ends, 0.99), ends.back(), q(lasts, 0.5), q(lasts, 0.99), lasts.back());
        }
    }
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *ms_out = (double)ms;
    if (s->dual_last && s->dual_launch) {        // did the two launches of the last frame share the work?  (see vk_scene::dual_strikes)
        uint32_t u[2] = {0u, 0u};
        if (hipMemcpy(u, s->counter + 4, sizeof(u), hipMemcpyDeviceToHost) == hipSuccess && u[0] + u[1] > 0u) {
            const double share = (double)u[1] / (double)(u[0] + u[1]);       // ~12 / 28 when both run side by side
            const bool lopsided = share < 0.10 || share > 0.90;
            s->dual_strikes = lopsided ? s->dual_strikes + 1 : 0;
            if (s->env.dual_debug)
                fprintf(stderr, "vecchio_amd: dual launch: 1024-thread launch %u units, 768-thread launch %u units (%.2f)\n", u[0], u[1],
                    share);
            if (s->dual_strikes >= 2) {
                s->dual_launch = false;
                fprintf(stderr, "vecchio_amd: the two launches of the 7-waves-per-SIMD shape do not run side by side on this runtime "
                                "(unit split %u / %u); using the single-launch shape from now on\n", u[0], u[1]);
            }
        }
        (void)hipGetLastError();
        s->dual_last = false;
    }
    return VK_OK;
}

// Samples of the last render whose radiance was clamped on its way into the fixed-point pixel sums; waits for the render's end.
int vk_scene_last_clamped_samples(vk_scene *s, uint64_t *out) {
    if (!s || !out) return fail(VK_ERR_BAD_ARG, "null argument");
    if (!s->last_timed) return fail(VK_ERR_BAD_ARG, "no render enqueued yet");
    *out = 0;
    if (!s->parts.empty()) {
        for (vk_scene *q : s->parts) {
            uint64_t v = 0;
            int rc = vk_scene_last_clamped_samples(q, &v);
            if (rc != VK_OK) return rc;
            *out += v;
        }
        return VK_OK;
    }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev1));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, reinterpret_cast<unsigned long long *>(s->counter) + 1, sizeof(v), hipMemcpyDeviceToHost));
    *out = v;
    return VK_OK;
}

// Exact re-treeing: samples of the last render that the first launch handed to the second one (rendered on the tree as handed
// over); waits for the render's end.  A frame whose queues overflowed is complete all the same (the fallback launch rendered it on the
// tree as handed over: enqueue_render_f32); it counts as entirely requeued.
int vk_scene_last_requeued_samples(vk_scene *s, uint64_t *out) {
    if (!s || !out) return fail(VK_ERR_BAD_ARG, "null argument");
    if (!s->last_timed) return fail(VK_ERR_BAD_ARG, "no render enqueued yet");
    *out = 0;
    if (!s->parts.empty()) {
        int worst = VK_OK;
        for (vk_scene *q : s->parts) {
            uint64_t v = 0;
            int rc = vk_scene_last_requeued_samples(q, &v);
            if (rc != VK_OK) worst = rc;
            *out += v;
        }
        return worst;
    }
    if (!s->redo_last) return VK_OK;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev1));
    if (!s->plan_copied) {      // (the caller ran more than two frames ahead: this frame's plan was not copied; read it now)
        uint32_t plan[4];
        HIP_TRY(hipMemcpy(plan, s->redo_count + REDO_REGIONS * REDO_COUNT_STRIDE, sizeof(plan), hipMemcpyDeviceToHost));
        *out = plan[2] != 0u ? s->redo_last_samples : plan[1];
        return VK_OK;
    }
    const int b = s->plan_last;
    *out = s->plan_host[4 * b + 2] != 0u ? s->redo_last_samples : s->plan_host[4 * b + 1];
    for (int k = 1; k <= 2; k++) {
        const int c = (b + k) & 1;
        if (s->plan_pending[c] && hipEventQuery(s->ev_plan[c]) == hipSuccess) judge_frame(s, c);
    }
    (void)hipGetLastError();
    return VK_OK;
}

static int render_host(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, void *out_host, vk_stats *stats_out,
    float *debug_out) {
    if (!out_host) return fail(VK_ERR_BAD_ARG, "null framebuffer");
    int rc = check_render_args(scene, cam, params);
    if (rc != VK_OK) return rc;
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(scene->device));
    const bool u8 = params->output_format == VK_OUTPUT_RGB8;
    size_t n_pixels = (size_t)params->width * params->height;
    size_t bytes = n_pixels * 3 * (u8 ? 1 : sizeof(float));
    // a multi-device group scatters into the image on devices[0]; a single device renders f32 into fb and converts into fb8
    void *d_img;
    if (u8) { rc = ensure(scene->fb8, scene->fb8_bytes, bytes); d_img = scene->fb8; }
    else { rc = ensure(scene->fb, scene->fb_bytes, bytes); d_img = scene->fb; }
    if (rc != VK_OK) return rc;
    vk_stats st;
    double ms = 0.0;
    {
        memset(&st, 0, sizeof(st));
        rc = enqueue_render(scene, cam, params, d_img, nullptr, debug_out != nullptr, &st);
        if (rc != VK_OK) return rc;
        HIP_TRY(hipStreamSynchronize(nullptr));
        uint64_t requeued = 0;
        rc = vk_scene_last_requeued_samples(scene, &requeued);      // (judges the frame: see vk_scene::exact_resume)
        if (rc != VK_OK) return rc;
    }
    rc = vk_scene_last_kernel_ms(scene, &ms);
    if (rc != VK_OK) return rc;
    st.kernel_ms = ms;
    rc = vk_scene_last_clamped_samples(scene, &st.clamped_samples);
    if (rc != VK_OK) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    uint32_t world = params->tile_world ? params->tile_world : 1;
    if (world == 1) {
        HIP_TRY(hipMemcpy(out_host, d_img, bytes, hipMemcpyDeviceToHost));     // the ONE device-to-host copy of the frame
    } else {
        // a partial image: only this call's tiles may be touched in the caller's buffer
        std::vector<uint8_t> tmp(bytes);
        HIP_TRY(hipMemcpy(tmp.data(), d_img, bytes, hipMemcpyDeviceToHost));
        uint32_t tiles_x = (params->width + TILE - 1) / TILE;
        const size_t px_bytes = u8 ? 3 : 12;
        uint8_t *dst = reinterpret_cast<uint8_t *>(out_host);
        for (uint32_t y = 0; y < params->height; y++)
            for (uint32_t x = 0; x < params->width; x++) {
                uint32_t tile = (y / TILE) * tiles_x + (x / TILE);
                if (tile % world != params->tile_rank) continue;
                uint32_t row = u8 ? params->height - 1 - y : y;                  // RGB8 images are top-down
                size_t i = ((size_t)row * params->width + x) * px_bytes;
                memcpy(dst + i, tmp.data() + i, px_bytes);
            }
    }
    if (debug_out) HIP_TRY(hipMemcpy(debug_out, scene->debug, n_pixels * params->samples_per_pixel * sizeof(float4),
        hipMemcpyDeviceToHost));
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats_out) *stats_out = st;
    return VK_OK;
}

int vk_render(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, float *rgb_out, vk_stats *stats_out) {
    return guarded([&]() -> int { return render_host(scene, cam, params, rgb_out, stats_out, nullptr); });
}

// test hook: as vk_render, also returning every sample: samples_out[(pixel*spp + s)*4 + 0..2]
// = radiance before the finite filter, [+3] = the sample's draw count (bit pattern)
int vk_debug_render_samples(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, float *rgb_out, float *samples_out) {
    if (!samples_out) return fail(VK_ERR_BAD_ARG, "null samples buffer");
    if (params && params->output_format != VK_OUTPUT_F32) return fail(VK_ERR_BAD_ARG, "per-sample debug output needs VK_OUTPUT_F32");
    return guarded([&]() -> int { return render_host(scene, cam, params, rgb_out, nullptr, samples_out); });
}

size_t vk_tile_slab_bytes(uint32_t width, uint32_t height, uint32_t output_format, uint32_t tile_rank, uint32_t tile_world) {
    if (width == 0 || height == 0 || output_format > VK_OUTPUT_RGB8) return 0;
    vk_render_params p; memset(&p, 0, sizeof(p));
    p.width = width; p.height = height; p.tile_world = tile_world ? tile_world : 1u;
    p.tile_rank = tile_rank < p.tile_world ? tile_rank : 0u;       // (rank 0 holds the most tiles)
    return (size_t)TileGeom(&p).n_local * 64u * (output_format == VK_OUTPUT_RGB8 ? 3u : 12u);
}

static int tile_call_args(vk_scene *scene, const void *a, const void *b, uint32_t width, uint32_t height, uint32_t output_format,
    uint32_t tile_rank, uint32_t tile_world, vk_render_params &p) {
    if (!scene || !a || !b) return fail(VK_ERR_BAD_ARG, "null argument");
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 31) / 3 || width > 65535u || height > 65535u) return fail(VK_ERR_BAD_ARG,
        "image too large");
    if (output_format > VK_OUTPUT_RGB8) return fail(VK_ERR_BAD_ARG, "bad output_format");
    memset(&p, 0, sizeof(p));
    p.width = width; p.height = height; p.tile_world = tile_world ? tile_world : 1u; p.tile_rank = tile_rank;
    if (p.tile_rank >= p.tile_world) return fail(VK_ERR_BAD_ARG, "tile_rank >= tile_world");
    return VK_OK;
}

int vk_pack_tiles_device(vk_scene *scene, const void *d_fb, uint32_t width, uint32_t height, uint32_t output_format, uint32_t tile_rank,
    uint32_t tile_world, void *d_slab, void *hip_stream) {
    vk_render_params p;
    int rc = tile_call_args(scene, d_fb, d_slab, width, height, output_format, tile_rank, tile_world, p);
    if (rc != VK_OK) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    return output_format == VK_OUTPUT_RGB8 ? tile_move<TM_PACK_U8>(d_fb, d_slab, &p, TileGeom(&p), st)
                                           : tile_move<TM_PACK_F32>(d_fb, d_slab, &p, TileGeom(&p), st);
}

int vk_unpack_tiles_device(vk_scene *scene, const void *d_slab, uint32_t width, uint32_t height, uint32_t output_format, uint32_t tile_rank,
    uint32_t tile_world, void *d_img, void *hip_stream) {
    vk_render_params p;
    int rc = tile_call_args(scene, d_slab, d_img, width, height, output_format, tile_rank, tile_world, p);
    if (rc != VK_OK) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    return output_format == VK_OUTPUT_RGB8 ? tile_move<TM_UNPACK_U8>(d_slab, d_img, &p, TileGeom(&p), st)
                                           : tile_move<TM_UNPACK_F32>(d_slab, d_img, &p, TileGeom(&p), st);
}

int vk_to_color_device(vk_scene *scene, const void *d_rgb, uint32_t width, uint32_t height, void *d_rgb8_out, void *hip_stream) {
    if (!scene || !d_rgb || !d_rgb8_out) return fail(VK_ERR_BAD_ARG, "null argument");
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 31) / 3) return fail(VK_ERR_BAD_ARG, "image too large");
    HIP_TRY(hipSetDevice(scene->device));
    size_t n = (size_t)width * height * 3;
    hipLaunchKernelGGL(to_color_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                       reinterpret_cast<const float *>(d_rgb), width, height, reinterpret_cast<uint8_t *>(d_rgb8_out));
    HIP_TRY(hipGetLastError());
    return VK_OK;
}

#ifdef VK_DEBUG_LIB      // libvecchio_amd_debug.so only (build.py build_device_debug): the product library holds production kernels only
// diagnostic: render with the instrumented kernel build and return the phase scheduler's counters (vecchio_amd_debug.h)
int vk_debug_phase_stats(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, uint64_t out[24]) {
    if (!out) return fail(VK_ERR_BAD_ARG, "null argument");
    int rc = check_render_args(scene, cam, params);
    if (rc != VK_OK) return rc;
    if (!scene->parts.empty() || params->output_format != VK_OUTPUT_F32) return fail(VK_ERR_UNSUPPORTED,
        "phase statistics: single device, VK_OUTPUT_F32");
    return guarded([&]() -> int {
        HIP_TRY(hipSetDevice(scene->device));
        int rc2 = ensure(scene->fb, scene->fb_bytes, (size_t)params->width * params->height * 3 * sizeof(float));
        if (rc2 != VK_OK) return rc2;
        scene->want_phase_stats = true;
        rc2 = enqueue_render(scene, cam, params, scene->fb, nullptr, false, nullptr);
        scene->want_phase_stats = false;
        if (rc2 != VK_OK) return rc2;
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(hipMemcpy(out, scene->phase_stats, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return VK_OK;
    });
}

// test hook: evaluate shared-math functions on the device (host arrays in/out)
int vk_debug_math(int device, int op, const float *a, const float *b, float *out, size_t n) {
    if (!a || !b || !out) return fail(VK_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    struct DevBuf {           // freed on every exit path
        float *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } da, db, dout;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&da.p), n * 4 + 16));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&db.p), n * 4 + 16));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dout.p), n * 4 + 16));
    HIP_TRY(hipMemcpy(da.p, a, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db.p, b, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, (const float *)da.p,
        (const float *)db.p, dout.p, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
    return VK_OK;
}

#endif

}  // extern "C"
