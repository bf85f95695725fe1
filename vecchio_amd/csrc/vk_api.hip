// vk_api.hip — the HIP megakernel and the C ABI of include/vecchio_amd.h (libvecchio_amd.so).
//
// Kernel structure (gfx950 / CDNA4, wave64):
//   * persistent workgroups (512-768 threads, sized by plan_residency); each WAVE repeatedly pulls one
//     work unit = (8x8-pixel tile, sample chunk) from a global atomic counter;
//   * one ray per lane.  A lane whose path ended pulls the next (pixel, sample) of the wave's unit
//     through a ballot + prefix-popcount ("active-ray compaction": lanes never idle while the unit
//     still has samples); the RNG is keyed (seed, pixel, sample), so the result does not depend on
//     which lane, wave or GPU traces a sample;
//   * a wave-level phase scheduler runs, each round, the code of the state most lanes are in: BOX
//     (box_steps: nested steps under one shrinking EXEC mask), PRIM light / heavy, SHADE + REFILL;
//     lane state that only shading needs (throughput, radiance, RNG, depth) is parked in LDS between
//     SHADE phases so the traversal loops fit 80 VGPRs (6 waves/SIMD) or 128 (4 waves/SIMD);
//   * traversal is the stack-free threaded walk of vk_trace.h; when the linear BVH + spheres + boxes
//     fit next to that per-wave state WITHOUT costing occupancy, every workgroup stages them into its
//     LDS (160 KB/CU) once and item fetches are ds_read_b128; otherwise they are L1/L2 gathers;
//   * per-pixel sums live in LDS (one float3 per tile pixel per wave, ds_add_f32), written once per
//     unit as a chunk partial; resolve_kernel adds the partials of a pixel in chunk order;
//   * no MFMA: there is no dense contraction in a path tracer.
//
// There is NO CPU fallback in this library: every entry point either runs on a gfx950
// device or returns an error.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/vecchio_amd.h"
#include "vk_linearize.h"
#include "vk_trace.h"

using namespace vkd;

namespace {

thread_local std::string g_err = "";

int fail(int code, const std::string &m) { g_err = m; return code; }

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return fail(VK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

constexpr size_t LDS_PER_CU = 160 * 1024;
constexpr int TILE = 8;   // 8x8 pixels = one wave
#ifndef VK_BOX_UNROLL
#define VK_BOX_UNROLL 4
#endif
// The scheduler runs SHADE + REFILL only when its lanes outnumber the box lanes AND the primitive lanes SHADE_DEFER
// times over: a shading phase costs ~700 issue slots against ~42 of a box step, so it pays to keep traversing with
// thinning waves until nearly every lane waits for shading and then shade them all at once.  C2 / C4, Msamples/s:
// 1 (plain plurality): 4 070 / 3 435; 1.5: 4 270 / 3 645; 2: 4 430 / 3 725; 3: 4 580 / 3 815; 4: 4 630 / 3 820;
// 8: 4 600 / 3 765.  (Deferring only against BOX, not PRIM: 4 150.)  Scenes that miss L2 (C5) are bound by the
// latency of the item gathers, not by issue slots: there every parked lane is a gather less in flight, and plain
// plurality is better (C5: 69 against 58 Msamples/s), so the factor is a launch parameter.
#ifndef VK_SHADE_DEFER
#define VK_SHADE_DEFER 4
#endif
constexpr uint32_t SHADE_DEFER = VK_SHADE_DEFER;
constexpr int BOX_UNROLL = VK_BOX_UNROLL;   // box steps between two exit tests of the BOX loop

struct KArgs {
    DScene S;
    RenderConsts C;
    float *out;              // full framebuffer (width*height*3)
    float *partial;          // [n_chunks][width*height*3] when n_chunks > 1
    float4 *debug;           // optional per-sample (rgb, draws) dump
    uint32_t *counter;       // work-unit counter
    uint32_t *tile_cost;     // [tiles of the image] time spent on each tile (1.6 us ticks): written by the probe (COST) build only
    const uint32_t *tile_order;   // [n_local_tiles] local tile slots, dearest first (from the probe launch), or null = raster order
    uint32_t tiles_x, tiles_y;
    uint32_t n_local_tiles;  // tiles of this call's partition
    uint32_t tile_rank, tile_world;
    uint32_t n_chunks;
    uint32_t shade_defer;    // SHADE + REFILL runs when its lanes outnumber box and primitive lanes this many times (see SHADE_DEFER)
    uint32_t lds_items, lds_spheres, lds_boxes;   // record counts staged into LDS (LDS variant)
    unsigned long long *phase_stats;   // optional (diagnostic build of the kernel): 16 counters, see vk_debug_phase_stats
};

// LDS-resident hot records
struct LdsMem {
    const uint4 *items;      // first halves of all items (x/y bounds) ...
    const uint4 *items_hi;   // ... then the second halves (z bounds, w0, w1): 16-byte stride per array gives a
                             // ds_read_b128 16 bank slots instead of the 8 a 32-byte stride leaves it
    const float4 *spheres;
    const uint4 *boxes;      // 2 x uint4 per DBox
    const uint32_t *sphere_mat;
    __device__ __forceinline__ DBox box(uint32_t i) const {
        uint4 a = boxes[2 * i], b = boxes[2 * i + 1];
        DBox o;
        o.p0[0] = __uint_as_float(a.x); o.p0[1] = __uint_as_float(a.y); o.p0[2] = __uint_as_float(a.z);
        o.p1x = __uint_as_float(a.w); o.p1y = __uint_as_float(b.x); o.p1z = __uint_as_float(b.y);
        o.mat = b.z; o._p = 0;
        return o;
    }
    __device__ __forceinline__ DItem item(uint32_t i) const {
        uint4 a = items[i], b = items_hi[i];
        DItem n;
        n.mnx = __uint_as_float(a.x); n.mxx = __uint_as_float(a.y); n.mny = __uint_as_float(a.z); n.mxy = __uint_as_float(a.w);
        n.mnz = __uint_as_float(b.x); n.mxz = __uint_as_float(b.y);
        n.w0 = b.z; n.w1 = b.w;
        return n;
    }
    __device__ __forceinline__ DSphere sphere(uint32_t i) const {
        float4 s = spheres[i];
        DSphere o; o.cx = s.x; o.cy = s.y; o.cz = s.z; o.r = s.w;
        return o;
    }
    __device__ __forceinline__ uint32_t smat(uint32_t i) const { return sphere_mat[i]; }
};

extern __shared__ uint4 smem[];

// The persistent loop below is one big region; left alone, LLVM hoists every value that is
// invariant across it (seed hashes, camera terms, scene pointers, division magic numbers)
// into the prologue and then spills them (>1 KB of scratch per lane, reloaded inside the hot
// loop).  So nothing is read from the by-value kernel argument directly: each phase re-reads
// what it needs from the kernarg segment (scalar loads, K$-resident) through a pointer that
// is laundered by an empty asm, which pins the loads, and everything derived from them,
// inside the phase that uses them.
typedef const __attribute__((address_space(4))) KArgs *KArgsC;
__device__ __forceinline__ KArgsC kargs_fresh() {
    KArgsC p = (KArgsC)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#define KARG(p, field) (*(const decltype(KArgs::field) *)&((p)->field))

template <uint32_t F, bool LDS_SCENE>
__device__ __forceinline__ typename std::conditional<LDS_SCENE, LdsMem, GlobalMem>::type make_mem(const DScene &S, uint32_t lds_items) {
    typename std::conditional<LDS_SCENE, LdsMem, GlobalMem>::type M;
    if constexpr (LDS_SCENE) {
        M.items = smem; M.items_hi = smem + lds_items; M.spheres = reinterpret_cast<const float4 *>(smem + 2u * lds_items);
        M.boxes = smem + 2u * lds_items + S.n_spheres; M.sphere_mat = S.sphere_mat;
    } else {
        M.items = S.items; M.spheres = S.spheres; M.sphere_mat = S.sphere_mat; M.boxes = S.boxes;
    }
    return M;
}

// Cold per-lane path state (throughput, radiance, RNG, pixel/sample ids, world ray) lives in
// LDS between SHADE phases, SoA by field (word f of lane l at cold[f*64 + l]: conflict-free),
// so that the box/primitive loops keep only the traversal state in VGPRs.
constexpr int NCOLD_BASE = 11;      // thr3 acc3 depth key2 ctr (q | sample << 6)
constexpr int NCOLD_INST = 17;      // + world-space ray (o3 d3) for scenes with instances
template <uint32_t F> constexpr int ncold() { return (F & VKF_INSTANCE) ? NCOLD_INST : NCOLD_BASE; }

template <uint32_t F>
__device__ __forceinline__ void cold_store(float *c, uint32_t lane, const Lane &L, uint32_t q) {
    c[0 * 64 + lane] = L.thr.x; c[1 * 64 + lane] = L.thr.y; c[2 * 64 + lane] = L.thr.z;
    c[3 * 64 + lane] = L.acc.x; c[4 * 64 + lane] = L.acc.y; c[5 * 64 + lane] = L.acc.z;
    c[6 * 64 + lane] = __uint_as_float(L.depth);
    c[7 * 64 + lane] = __uint_as_float((uint32_t)L.rng.key); c[8 * 64 + lane] = __uint_as_float((uint32_t)(L.rng.key >> 32));
    c[9 * 64 + lane] = __uint_as_float(L.rng.ctr);
    c[10 * 64 + lane] = __uint_as_float(q | (L.sample << 6));   // the pixel is implied by the unit's tile and q
    if (F & VKF_INSTANCE) {
        c[11 * 64 + lane] = L.wo.x; c[12 * 64 + lane] = L.wo.y; c[13 * 64 + lane] = L.wo.z;
        c[14 * 64 + lane] = L.wd.x; c[15 * 64 + lane] = L.wd.y; c[16 * 64 + lane] = L.wd.z;
    }
}
template <uint32_t F>
__device__ __forceinline__ void cold_load(const float *c, uint32_t lane, Lane &L, uint32_t &q) {
    L.thr = v3(c[0 * 64 + lane], c[1 * 64 + lane], c[2 * 64 + lane]);
    L.acc = v3(c[3 * 64 + lane], c[4 * 64 + lane], c[5 * 64 + lane]);
    L.depth = __float_as_uint(c[6 * 64 + lane]);
    L.rng.key = (uint64_t)__float_as_uint(c[7 * 64 + lane]) | ((uint64_t)__float_as_uint(c[8 * 64 + lane]) << 32);
    L.rng.ctr = __float_as_uint(c[9 * 64 + lane]);
    uint32_t qs = __float_as_uint(c[10 * 64 + lane]);
    q = qs & 63u; L.sample = qs >> 6; L.pixel = 0;
    if (F & VKF_INSTANCE) {
        L.wo = v3(c[11 * 64 + lane], c[12 * 64 + lane], c[13 * 64 + lane]);
        L.wd = v3(c[14 * 64 + lane], c[15 * 64 + lane], c[16 * 64 + lane]);
    } else {
        L.wo = L.o; L.wd = L.d;       // no instances: the current space IS world space
    }
}
template <uint32_t F>
__device__ __forceinline__ void cold_load_world_ray(const float *c, uint32_t lane, Lane &L) {
    if (F & VKF_INSTANCE) {
        L.wo = v3(c[11 * 64 + lane], c[12 * 64 + lane], c[13 * 64 + lane]);
        L.wd = v3(c[14 * 64 + lane], c[15 * 64 + lane], c[16 * 64 + lane]);
    }
}

// LDS word of a wave that holds the time its previous unit ended.  Re-derived from freshly loaded kernel arguments at
// its uses (kernel start, unit end) so that no pointer stays live across the traversal loops.
template <uint32_t F, bool LDS_SCENE>
__device__ __forceinline__ uint32_t *unit_t0_word(uint32_t wave) {
    KArgsC P = kargs_fresh();
    uint32_t scene16 = LDS_SCENE ? 2u * KARG(P, lds_items) + KARG(P, lds_spheres) + 2u * KARG(P, lds_boxes) : 0u;
    float *dyn = reinterpret_cast<float *>(smem + scene16);
    return reinterpret_cast<uint32_t *>(dyn + (blockDim.x >> 6) * (64 * 3 + 64 * ncold<F>())) + wave;
}

// number of lanes of the wave for which p holds (v_cmp -> s_bcnt1, no VGPR round trip)
__device__ __forceinline__ uint32_t lanes_with(bool p) { return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }

template <uint32_t F, bool LDS_SCENE, int MINW, bool STATS, bool COST = false>
__global__ __launch_bounds__((MINW <= 4 ? 1024 : (MINW == 5 ? 640 : (MINW == 6 ? 768 : 1024))), MINW) void render_kernel(KArgs A_byval) {
    (void)A_byval;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    using Mem = typename std::conditional<LDS_SCENE, LdsMem, GlobalMem>::type;
    // diagnostic counters (STATS build only): phase executions and the lanes that had work in them
    unsigned long long st_box_steps = 0, st_box_lanes = 0, st_prim_execs = 0, st_prim_lanes = 0, st_shade_execs = 0, st_shade_lanes = 0, st_sched = 0;
    unsigned long long st_t_box = 0, st_t_light = 0, st_t_heavy = 0, st_t_shade = 0, st_heavy_execs = 0, st_t0 = 0, st_t_total = 0, st_t_mat = 0, st_t_refill = 0, st_t_install = 0, st_t1 = 0;
    if (STATS) st_t_total = clock64();

    // ---- LDS layout: [items][spheres][per-wave pixel accumulators][per-wave cold lane state]
    uint32_t lds_items = 0;
    float *acc_lds, *cold;
    {
        KArgsC P = kargs_fresh();
        lds_items = LDS_SCENE ? KARG(P, lds_items) : 0u;
        uint32_t lds_spheres = LDS_SCENE ? KARG(P, lds_spheres) : 0u;
        uint32_t lds_boxes = LDS_SCENE ? KARG(P, lds_boxes) : 0u;
        float *dyn = reinterpret_cast<float *>(smem + (2u * lds_items + lds_spheres + 2u * lds_boxes));
        acc_lds = dyn + wave * (64 * 3);
        cold = dyn + (blockDim.x >> 6) * (64 * 3) + wave * (64 * ncold<F>());
        if (LDS_SCENE) {
            const uint4 *gi = reinterpret_cast<const uint4 *>(KARG(P, S.items));
            for (uint32_t k = threadIdx.x; k < 2u * lds_items; k += blockDim.x) smem[(k >> 1) + ((k & 1u) ? lds_items : 0u)] = gi[k];
            const uint4 *gs = reinterpret_cast<const uint4 *>(KARG(P, S.spheres));
            for (uint32_t k = threadIdx.x; k < lds_spheres; k += blockDim.x) smem[2u * lds_items + k] = gs[k];
            const uint4 *gb = reinterpret_cast<const uint4 *>(KARG(P, S.boxes));
            for (uint32_t k = threadIdx.x; k < 2u * lds_boxes; k += blockDim.x) smem[2u * lds_items + lds_spheres + k] = gb[k];
            __syncthreads();
        }
    }

    if (COST) { if (lane == 0) *unit_t0_word<F, LDS_SCENE>(wave) = (uint32_t)(wall_clock64() >> 4); }
    // (per-XCD work queues — contiguous image bands per XCD, stealing when empty — were tried for L2 locality on
    // C5: no gain there, -1.7 % on C2, and the two extra live scalars doubled the everything-variant's spills)
    for (;;) {
        KArgsC U = kargs_fresh();
        uint32_t unit = 0;
        if (lane == 0) unit = atomicAdd(KARG(U, counter), 1u);
        unit = __builtin_amdgcn_readfirstlane(unit);
        const uint32_t n_chunks = KARG(U, n_chunks);
        if (unit >= KARG(U, n_local_tiles) * n_chunks) break;
        const uint32_t chunk = unit % n_chunks;
        // tiles are visited dearest-first when the probe launch left an order (see enqueue_render)
        uint32_t tslot = unit / n_chunks;
        { const uint32_t *ord = KARG(U, tile_order); if (ord) tslot = ord[tslot]; }
        const uint32_t tile = KARG(U, tile_rank) + tslot * KARG(U, tile_world);
        const uint32_t tiles_x = KARG(U, tiles_x);
        const uint32_t tx = (tile % tiles_x) * TILE, ty = (tile / tiles_x) * TILE;
        const uint32_t spp = KARG(U, C.spp);
        const uint32_t s0 = (uint32_t)(((uint64_t)spp * chunk) / n_chunks);
        const uint32_t s1 = (uint32_t)(((uint64_t)spp * (chunk + 1)) / n_chunks);
        const uint32_t total = 64u * (s1 - s0);       // items: k -> (pixel slot k & 63, sample s0 + (k >> 6))

        acc_lds[lane] = 0.0f; acc_lds[64 + lane] = 0.0f; acc_lds[128 + lane] = 0.0f;

        Lane L;
        memset(&L, 0, sizeof(L));
        bool need = true;          // lane wants a new (pixel, sample)
        bool active = false;       // lane holds a live path
        uint32_t next_item = 0;    // wave-uniform
        // Wave-level phase scheduler.  Every lane is in one of four states; each round the
        // wave runs the code of the most populated state with the lanes in it (64-bit ballots
        // + s_bcnt1), so the long box loop, the primitive tests and the (expensive, rare)
        // shading / ray-generation code each execute with as many lanes as possible instead
        // of all being paid for on every iteration:
        //   BOX    pend == 0 and items (or an instance to leave) remain  -> box_step
        //   PRIM   pend != 0                                             -> prim_step
        //   SHADE  live path whose segment is fully traversed            -> shade
        //   NEED   no path; the unit still has (pixel, sample) items     -> start_sample
        // SHADE and NEED run as one phase: a path that ends hands its lane straight to the
        // next item (ballot + prefix popcount = active-ray compaction).
        const uint32_t shade_defer = KARG(U, shade_defer);
        for (;;) {
            bool is_prim = active && has_prim_work(L);
            bool is_box = active && !is_prim && traversing(L);
            bool is_shade = active && !is_prim && !is_box;
            // primitives come in two weights (sphere/rect ~50 instructions; Boxy, list, medium, instance
            // entry several times that): scheduled separately so cheap tests never pay for heavy ones
            const bool HAS_HEAVY = (F & (VKF_LIST | VKF_MEDIUM | VKF_INSTANCE | VKF_BOX)) != 0;
            bool is_heavy = HAS_HEAVY && is_prim && prim_is_heavy(L.pend);
            uint32_t n_box = lanes_with(is_box);
            uint32_t n_heavy = HAS_HEAVY ? lanes_with(is_heavy) : 0u;
            uint32_t n_light = lanes_with(is_prim && !is_heavy);
            if (n_heavy > n_light) { is_prim = is_heavy; } else { is_prim = is_prim && !is_heavy; }
            uint32_t n_prim = n_heavy > n_light ? n_heavy : n_light;
            uint32_t n_sn = lanes_with(is_shade || need);
            if ((n_box | n_prim | n_sn) == 0) break;
            if (STATS) st_sched++;
            if (n_box >= n_prim && n_box * shade_defer >= n_sn) {
                // ---- BOX: UNROLL steps under a shrinking EXEC mask per exit test, while box lanes are the plurality
                KArgsC P = kargs_fresh();
                DScene S = KARG(P, S);
                Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
                if (STATS) st_t0 = clock64();
                cold_load_world_ray<F>(cold, lane, L);
                const uint32_t live = n_box + n_heavy + n_light + n_sn;   // lanes only change state here, none appear or vanish
                // steps between two exit tests (compile time: a run-time trip count costs 4-13 %).  With shading deferred the
                // optimum is 3 for the sphere-only and Cornell-type variants (against 4: C2 +0.7 %, C4 +1.3 %; 2: -3 %, 6: -1 %,
                // 8: -6 %) and 4 for the everything-variants (C3: 3 -> -2 %)
                constexpr int UNROLL = ((F & VKF_ALL_SCENE) == VKF_ALL_SCENE) ? BOX_UNROLL : BOX_UNROLL - 1;
                for (;;) {
                    if (F & VKF_INSTANCE) {     // end of an instance's item range: back to the parent space (rare)
                        if (is_box && L.i >= L.end && L.cur_inst >= 0) leave_instance<F, Mem>(L, S);
                    }
                    bool go = active && L.pend == 0u && L.i < range_end<F>(L, S);
                    if (STATS) {        // diagnostic build: same steps one at a time, counting the lanes in each
                        for (int u = 0; u < UNROLL; u++) {
                            st_box_steps += 1; st_box_lanes += lanes_with(go);
                            box_steps<F, Mem, 1>(L, S, M, go);
                            go = active && L.pend == 0u && L.i < range_end<F>(L, S);
                        }
                    } else {
                        box_steps<F, Mem, UNROLL>(L, S, M, go);
                    }
                    is_box = active && !has_prim_work(L) && traversing(L);
                    uint32_t nb = lanes_with(is_box);
                    uint32_t np = lanes_with(active && has_prim_work(L));
                    uint32_t ns = live - nb - np;
                    if (nb == 0 || nb < np || nb * shade_defer < ns) {               // another state now has more lanes parked than are stepping
                        // sphere-only variants: when that state is PRIM, test the pending spheres right here and
                        // keep stepping (saves the scheduler round trip that otherwise follows every ~10 box steps)
                        if (!HAS_HEAVY && !(F & VKF_MEDIUM) && np != 0 && np * shade_defer >= ns) {
                            if (active && has_prim_work(L)) prim_step<F, Mem>(L, S, M);
                            is_box = active && !has_prim_work(L) && traversing(L);
                            continue;
                        }
                        break;
                    }
                }
                if (STATS) st_t_box += clock64() - st_t0;
            } else if (n_prim * shade_defer >= n_sn) {
                // ---- PRIM: intersect / enter the pending object
                KArgsC P = kargs_fresh();
                DScene S = KARG(P, S);
                Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
                if (STATS) { st_prim_execs++; st_prim_lanes += n_prim; st_t0 = clock64(); if (n_heavy > n_light) st_heavy_execs++; }
                if (is_prim) {
                    if (F & VKF_MEDIUM) {    // ConstantMedium::hit draws inside traversal (hittable.rs:473)
                        L.rng.key = (uint64_t)__float_as_uint(cold[7 * 64 + lane]) | ((uint64_t)__float_as_uint(cold[8 * 64 + lane]) << 32);
                        L.rng.ctr = __float_as_uint(cold[9 * 64 + lane]);
                    }
                    prim_step<F, Mem>(L, S, M);
                    if (F & VKF_MEDIUM) cold[9 * 64 + lane] = __uint_as_float(L.rng.ctr);
                }
                if (STATS) { if (n_heavy > n_light) st_t_heavy += clock64() - st_t0; else st_t_light += clock64() - st_t0; }
            } else {
                // ---- SHADE + REFILL
                KArgsC P = kargs_fresh();
                RenderConsts C = KARG(P, C);
                DScene S = KARG(P, S);
                Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
                if (STATS) { st_shade_execs++; st_shade_lanes += n_sn; st_t0 = clock64(); }
                uint32_t q = 0;
                bool touched = is_shade;      // lanes whose cold state is in registers during this phase
                bool fresh = false;           // lanes that leave this phase with a new ray to install; it is parked in the
                                              // (dead) world-ray fields L.wo / L.wd / L.time, so it costs no registers
                // (the everything-variant keeps one begin_segment per call site: merging them there doubled its spills)
                constexpr bool ONE_INSTALL = (F & VKF_ALL_SCENE) != VKF_ALL_SCENE;
                if (is_shade) {
                    cold_load<F>(cold, lane, L, q);
                    if (STATS) st_t1 = clock64();
                    bool cont;
                    if (ONE_INSTALL) {
                        V3 no, nd; float nt;
                        cont = shade_core<F, Mem>(L, S, M, C, no, nd, nt);
                        if (cont) { L.wo = no; L.wd = nd; L.time = nt; fresh = true; }
                    } else {
                        cont = shade<F, Mem>(L, S, M, C);
                    }
                    if (STATS) st_t_mat += clock64() - st_t1;
                    if (!cont) {
                        float4 *dbg = KARG(P, debug);
                        if (dbg) dbg[((size_t)(ty + (q >> 3)) * C.width + (tx + (q & 7u))) * C.spp + L.sample] = make_float4(L.acc.x, L.acc.y, L.acc.z, __uint_as_float(L.rng.ctr));
                        if (isfinite(L.acc.x) && isfinite(L.acc.y) && isfinite(L.acc.z)) {   // main.rs:192-194
                            atomicAdd(&acc_lds[q * 3 + 0], L.acc.x);
                            atomicAdd(&acc_lds[q * 3 + 1], L.acc.y);
                            atomicAdd(&acc_lds[q * 3 + 2], L.acc.z);
                        }
                        active = false;
                        need = true;
                    }
                }
                if (STATS) st_t1 = clock64();
                unsigned long long need_mask = __builtin_amdgcn_ballot_w64(need);
                if (need_mask) {
                    uint32_t rank = __popcll(need_mask & ((1ull << lane) - 1ull));
                    if (need) {
                        uint32_t k = next_item + rank;
                        if (k < total) {
                            q = k & 63u;
                            uint32_t s = s0 + (k >> 6);
                            uint32_t px = tx + (q & 7u), py = ty + (q >> 3);
                            if (px < C.width && py < C.height) {   // slots outside the image (edge tiles) are skipped
                                if (ONE_INSTALL) {
                                    V3 no, nd; float nt;
                                    start_sample_core(L, C, px, py, s, no, nd, nt);
                                    L.wo = no; L.wd = nd; L.time = nt;
                                    fresh = true;
                                } else {
                                    start_sample(L, S, C, px, py, s);
                                }
                                active = true;
                                need = false;
                                touched = true;
                            }
                        } else {
                            need = false;                           // unit exhausted: this lane idles until the wave drains
                        }
                    }
                    next_item += (uint32_t)__popcll(need_mask);
                }
                if (STATS) { st_t_refill += clock64() - st_t1; st_t1 = clock64(); }
                if (ONE_INSTALL && fresh) begin_segment(L, S, L.wo, L.wd, L.time);   // one copy of the exact reciprocals for both kinds of new ray
                if (active && touched) cold_store<F>(cold, lane, L, q);
                if (STATS) { st_t_shade += clock64() - st_t0; st_t_install += clock64() - st_t1; }
            }
        }
        // ---- write the unit's pixel sums
        {
            KArgsC P = kargs_fresh();
            if (COST) {   // probe launch: time since this wave's previous unit ended = this unit's cost (units run back to back)
                uint32_t *tc = KARG(P, tile_cost);
                if (tc && lane == 0) {
                    uint32_t *w = unit_t0_word<F, LDS_SCENE>(wave);
                    uint32_t now = (uint32_t)(wall_clock64() >> 4);
                    atomicAdd(&tc[(ty / TILE) * KARG(P, tiles_x) + tx / TILE], now - *w);
                    *w = now;
                }
            }
            uint32_t width = KARG(P, C.width), height = KARG(P, C.height);
            uint32_t px = tx + (lane & 7u), py = ty + (lane >> 3);
            if (px < width && py < height) {
                size_t pix = (size_t)py * width + px;
                float r = acc_lds[lane * 3 + 0], g = acc_lds[lane * 3 + 1], b = acc_lds[lane * 3 + 2];
                if (n_chunks == 1) {
                    float n = (float)KARG(P, C.spp);                         // main.rs:196
                    float *o = KARG(P, out);
                    o[pix * 3 + 0] = r / n; o[pix * 3 + 1] = g / n; o[pix * 3 + 2] = b / n;
                } else {
                    float *p = KARG(P, partial) + ((size_t)chunk * ((size_t)width * height) + pix) * 3;
                    p[0] = r; p[1] = g; p[2] = b;
                }
            }
        }
    }
    if (STATS) {
        KArgsC P = kargs_fresh();
        unsigned long long *ps = KARG(P, phase_stats);
        if (ps && lane == 0) {
            atomicAdd(&ps[0], st_box_steps); atomicAdd(&ps[1], st_box_lanes); atomicAdd(&ps[2], st_prim_execs); atomicAdd(&ps[3], st_prim_lanes);
            atomicAdd(&ps[4], st_shade_execs); atomicAdd(&ps[5], st_shade_lanes); atomicAdd(&ps[6], st_sched); atomicAdd(&ps[7], st_heavy_execs);
            atomicAdd(&ps[8], st_t_box); atomicAdd(&ps[9], st_t_light); atomicAdd(&ps[10], st_t_heavy); atomicAdd(&ps[11], st_t_shade);
            atomicAdd(&ps[12], (unsigned long long)(clock64() - st_t_total));
            atomicAdd(&ps[13], st_t_mat); atomicAdd(&ps[14], st_t_refill); atomicAdd(&ps[15], st_t_install);
        }
    }
}

// ---- heavy-first tile order (bucket sort of the probe's per-tile times, dearest first).
// 8 buckets per octave of cost; the order inside a bucket is arbitrary, which is fine: any order renders the same image.
constexpr uint32_t ORDER_BUCKETS = 256;
__device__ __forceinline__ uint32_t cost_bucket(uint32_t c) {
    if (c < 8u) return c;
    uint32_t msb = 31u - (uint32_t)__builtin_clz(c);
    uint32_t b = (msb - 2u) * 8u + ((c >> (msb - 3u)) & 7u);
    return b < ORDER_BUCKETS ? b : ORDER_BUCKETS - 1u;
}
__global__ void order_hist_kernel(const uint32_t *cost, uint32_t n_local, uint32_t tile_rank, uint32_t tile_world, uint32_t *hist) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_local) atomicAdd(&hist[cost_bucket(cost[tile_rank + i * tile_world])], 1u);
}
__global__ void order_scan_kernel(uint32_t *hist) {      // one thread: start offset of every bucket, dearest bucket first
    uint32_t run = 0;
    for (int b = (int)ORDER_BUCKETS - 1; b >= 0; b--) { uint32_t c = hist[b]; hist[b] = run; run += c; }
}
__global__ void order_scatter_kernel(uint32_t *cost, uint32_t n_local, uint32_t tile_rank, uint32_t tile_world, uint32_t *hist, uint32_t *order) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_local) return;
    uint32_t t = tile_rank + i * tile_world;
    order[atomicAdd(&hist[cost_bucket(cost[t])], 1u)] = i;
}

// sums the sample chunks of each pixel in chunk order (deterministic) and divides by spp
__global__ void resolve_kernel(const float *partial, float *out, uint32_t width, uint32_t height, uint32_t n_chunks, uint32_t spp,
                               uint32_t tiles_x, uint32_t tile_rank, uint32_t tile_world) {
    uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_pixels = width * height;
    if (pix >= n_pixels) return;
    uint32_t x = pix % width, y = pix / width;
    uint32_t tile = (y / TILE) * tiles_x + (x / TILE);
    if (tile % tile_world != tile_rank) return;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t c = 0; c < n_chunks; c++) {
        const float *p = partial + ((size_t)c * n_pixels + pix) * 3;
        r += p[0]; g += p[1]; b += p[2];
    }
    float n = (float)spp;
    out[(size_t)pix * 3 + 0] = r / n; out[(size_t)pix * 3 + 1] = g / n; out[(size_t)pix * 3 + 2] = b / n;
}

// Vec3::to_color (vec3.rs:54-61) + top-down rows (main.rs:209)
__global__ void to_color_kernel(const float *rgb, uint32_t width, uint32_t height, uint8_t *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = width * height * 3;
    if (i >= n) return;
    uint32_t c = i % 3, pix = i / 3;
    uint32_t x = pix % width, row = pix / width;
    uint32_t y = height - 1 - row;
    float v = sqrtf(rgb[((size_t)y * width + x) * 3 + c]);
    float cl = v < 0.0f ? 0.0f : (v > 0.999f ? 0.999f : v);      // NaN falls through (vec3.rs:44-52) ...
    out[i] = (uint8_t)vk::sat_u32(256.0f * cl);                    // ... and `as u32` maps NaN to 0
}

// device math probe (tests: GPU transcendental/draw functions are bit-identical to the host's)
__global__ void math_probe_kernel(int op, const float *a, const float *b, float *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (op) {
        case 0: r = vk::sinf_(a[i]); break;
        case 1: r = vk::cosf_(a[i]); break;
        case 2: r = vk::logf_(a[i]); break;
        case 3: r = vk::asinf_(a[i]); break;
        case 4: r = vk::atan2f_(a[i], b[i]); break;
        case 5: r = vk::pow5f_(a[i]); break;
        case 6: r = a[i] / b[i]; break;
        case 7: r = sqrtf(a[i]); break;
        case 8: { vk::Rng g = vk::rng_for_sample(__float_as_uint(a[i]), (uint32_t)i, 0); r = vk::gen_range(g, -1.0f, 1.0f) + vk::gen_f32(g); break; }
        case 9: r = a[i] * b[i] + a[i]; break;   // must stay an unfused mul+add
    }
    out[i] = r;
}

}  // namespace

// =========================================================================================
struct vk_scene {
    int device = 0;
    LinearScene host;          // kept for introspection
    DScene dev;                // device pointers
    std::vector<void *> allocs;
    uint32_t *counter = nullptr;
    float *fb = nullptr; size_t fb_bytes = 0;        // framebuffer for vk_render
    float *partial = nullptr; size_t partial_bytes = 0;
    float4 *debug = nullptr; size_t debug_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cus = 256;
    uint32_t lds_bytes = 0;    // hot-record bytes staged per workgroup (0 = not LDS resident)
    size_t hot_bytes = 0;      // items + spheres + boxes: what traversal gathers from
    uint32_t wg_threads = 512; // workgroup size chosen by plan_residency()
    uint32_t sphere_waves = 6; // waves per SIMD of the sphere-only variant (8 was measured 3 % slower: it spills)
    uint32_t wgs_per_cu = 2;
    bool last_timed = false;
    unsigned long long *phase_stats = nullptr;   // device, 16 counters (diagnostic kernel build)
    bool want_phase_stats = false;
    // heavy-first tile order: per-tile times of the probe launch and the order derived from them
    uint32_t *tile_cost = nullptr, *tile_order = nullptr, *order_hist = nullptr;
    size_t tile_cost_n = 0, tile_order_n = 0;
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

uint32_t pick_variant(uint32_t features) {
    const uint32_t F_CORNELL = VKF_RECT | VKF_LIST | VKF_INSTANCE | VKF_BOX;
    if (const char *e = getenv("VK_FORCE_FULL_VARIANT")) { if (e[0] == '1') return VKF_ALL_SCENE; }   // diagnostics: cost of the general kernel
    if (features == 0) return 0u;
    if ((features & ~F_CORNELL) == 0) return F_CORNELL;
    return VKF_ALL_SCENE;
}

size_t per_wave_lds_bytes(uint32_t F) {   // pixel accumulators + cold lane state of one wave
    return (size_t)64 * (3 + ((F & VKF_INSTANCE) ? NCOLD_INST : NCOLD_BASE)) * sizeof(float) + sizeof(uint32_t);   // + the unit's start time
}

// LDS residency plan.  `hot` = bytes of items + spheres + boxes.  Measured on MI355X with the VALU-bound
// kernel: at EQUAL occupancy a scene staged in LDS beats the same scene read through L1/L2 by only 7 % (C2:
// 4.06 vs 3.77 Gsamples/s at 24 waves/CU; C4: no difference), while occupancy is worth much more (C3: the
// 118 KB scene in LDS leaves 11 waves/CU = 371 Msamples/s; from L2 at 16 waves/CU = 503).  So the scene is
// staged in LDS only when that costs no waves: choose the workgroup size (waves share one LDS copy) and
// workgroups per CU that reach the variant's full occupancy (24 waves/CU at 80 VGPRs, 16 at 128) with the
// scene resident, else traverse from global memory at full occupancy.
void plan_residency(vk_scene *s, size_t hot) {
    const size_t pw = per_wave_lds_bytes(pick_variant(s->host.features));
    uint32_t best_waves = 0, best_wg = 0, best_n = 0;
    const bool spheres_only = pick_variant(s->host.features) == 0u;
    uint32_t cap = spheres_only ? 4 * s->sphere_waves : 16;                  // waves per CU the variant's register budget admits
    const uint32_t max_wg_waves = (spheres_only && s->sphere_waves == 6) ? 12 : 16;   // = the variant's __launch_bounds__ thread limit / 64
    if (const char *e = getenv("VK_MAX_WAVES_PER_CU")) { int v = atoi(e); if (v >= 4 && (uint32_t)v < cap) cap = (uint32_t)v; }   // diagnostics: lower the occupancy
    for (uint32_t n_wg = 1; n_wg <= 4; n_wg++) {
        size_t budget = LDS_PER_CU / n_wg;
        if (hot + 4 * pw > budget) break;
        uint32_t w = (uint32_t)std::min<size_t>(std::min<uint32_t>(max_wg_waves, cap / n_wg), (budget - hot) / pw);
        if (w < 1) break;
        if (w * n_wg > best_waves) { best_waves = w * n_wg; best_wg = w; best_n = n_wg; }
    }
    bool no_lds = false;
    if (const char *e = getenv("VK_NO_LDS_SCENE")) no_lds = e[0] == '1';   // diagnostics: traverse from L2 at full occupancy
    if (best_waves >= cap && !no_lds) {
        s->lds_bytes = (uint32_t)hot; s->wg_threads = best_wg * 64; s->wgs_per_cu = best_n;
    } else {
        s->lds_bytes = 0; s->wg_threads = 512; s->wgs_per_cu = cap / 8;   // 24 (sphere-only) or 16 waves per CU
    }
}

template <uint32_t F, int MINW_SPHERES = 6>
int launch_variant(vk_scene *s, const KArgs &A, bool lds, dim3 grid, size_t shmem, hipStream_t st, bool cost) {
    // register budget: the sphere-only kernels fit 80 VGPRs (6 waves per SIMD, 24 per CU), the others are held to 128 (4 per SIMD)
    constexpr int MINW = ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) ? MINW_SPHERES : 4;
    // (global-memory traversal — C5, 33 MB of items — was also tried at 8 waves per SIMD / 64 VGPRs: 8 % slower.
    // It is bound by the L2-miss path: every box step gathers a 32-byte item but moves a 128-byte line,
    // ~3.5 TB/s of lines from the Infinity Cache at 21 Msamples/s; more waves in flight do not help.)
    auto go = [&](auto kernel) -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(kernel, grid, dim3(s->wg_threads), shmem, st, A);
        return VK_OK;
    };
    int rc;
    if (cost) rc = lds ? go(&render_kernel<F, true, MINW, false, true>) : go(&render_kernel<F, false, MINW, false, true>);
    else rc = lds ? go(&render_kernel<F, true, MINW, false, false>) : go(&render_kernel<F, false, MINW, false, false>);
    if (rc != VK_OK) return rc;
    HIP_TRY(hipGetLastError());
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
    if (p->width < 2 || p->height < 2) return fail(VK_ERR_BAD_ARG, "width and height must be >= 2 (u,v divide by width-1/height-1, main.rs:187-188)");
    if ((uint64_t)p->width * p->height > (1ull << 31) / 3) return fail(VK_ERR_BAD_ARG, "image too large");
    if (p->samples_per_pixel == 0 || p->samples_per_pixel > (1u << 26)) return fail(VK_ERR_BAD_ARG, "samples_per_pixel must be in 1..2^26");
    if (!(cam->time0 < cam->time1)) return fail(VK_ERR_BAD_ARG, "camera time0 >= time1 (gen_range panics, main.rs:118)");
    if (p->integrator > VK_INTEGRATOR_SCATTER || p->background > VK_BACKGROUND_SKY) return fail(VK_ERR_BAD_ARG, "bad integrator/background");
    uint32_t world = p->tile_world ? p->tile_world : 1;
    if (p->tile_rank >= world) return fail(VK_ERR_BAD_ARG, "tile_rank >= tile_world");
    if (p->integrator == VK_INTEGRATOR_PDF && scene->host.lights.empty())
        return fail(VK_ERR_UNSUPPORTED, "PDF integrator with an empty lights list (Vec::random unwraps None, hittable.rs:431)");
    if (p->integrator == VK_INTEGRATOR_SCATTER && (scene->host.features & VKF_SPEC_DIFFUSE))
        return fail(VK_ERR_UNSUPPORTED, "SpecDiffuse has no Material::scatter (default impl unwraps a None specular ray, material.rs:21-28)");
    return VK_OK;
}

// number of sample chunks per tile: a function of the image and spp ONLY (never of the tile
// partition), so 1-GPU and N-GPU renders sum every pixel in the same order
uint32_t choose_chunks(const vk_scene *s, const vk_render_params *p) {
    // Samples per pixel per unit: large enough that the end-of-unit tail (lanes idling while the last long
    // paths of a unit finish) stays small, small enough that tiles of very different cost (fog, glass,
    // grazing rays over 1M spheres) are spread over many waves; small images get smaller chunks so that
    // there are still ~64K units for ~6K waves.
    // A function of the image, spp and scene only, never of the tile partition (see above).
    // (Tried for the multi-GPU tail — one rank of 8 renders C2's 1/8 in 77.7 ms against 65.3 ideal: cutting the last
    // chunk of every tile into 8 small ones and ordering the units chunk by chunk so that the launch ends on short
    // units.  Slower, 80.4 ms: short units pay the per-unit drain; the tail is heavy tiles, not the last unit.)
    uint64_t tiles = (uint64_t)((p->width + TILE - 1) / TILE) * ((p->height + TILE - 1) / TILE);
    uint64_t c = (uint64_t)p->samples_per_pixel * tiles / 65536u;
    uint32_t cap = 64;    // measured on C2: 64..128 samples per pixel per unit is the optimum (256: -3 %, 32: -5 %)
    // Scenes that exceed an XCD's L2 (C5): tile costs are skewed by orders of magnitude (rays grazing a million
    // spheres at the horizon) and one wave's 64-spp unit of the dearest tile was the critical path of the whole
    // launch (C5 at 64 spp: 67 s, 28 s with 8-spp units).  Short units pay the per-unit drain (lanes idle while
    // the unit's last paths finish), so: at least 8 units per tile, 8..32 spp each (C5 at 256 spp: 8 -> 44, 16 -> 54,
    // 32 -> 55, 64 -> 45 Msamples/s).
    if (s->hot_bytes > (4u << 20)) {      // bigger than one XCD's L2
        uint32_t v = p->samples_per_pixel / 8u;
        cap = v < 8u ? 8u : (v > 32u ? 32u : v);
    }
    if (const char *e = getenv("VK_CHUNK_CAP")) { int v = atoi(e); if (v >= 1) cap = (uint32_t)v; }   // diagnostics
    uint32_t lo = cap < 32 ? cap : 32;
    uint32_t chunk_spp = (uint32_t)(c > cap ? cap : (c < lo ? lo : c));
    uint32_t n = (p->samples_per_pixel + chunk_spp - 1) / chunk_spp;
    // the chunk partials cost n x framebuffer bytes: keep them under 8 GiB
    const uint64_t fb = (uint64_t)p->width * p->height * 12u;
    while (n > 1 && (uint64_t)n * fb > (8ull << 30)) n = (n + 1) / 2;
    if (n < 1) n = 1;
    return n;
}

int enqueue_render(vk_scene *s, const vk_camera *cam, const vk_render_params *p, float *d_out, hipStream_t st, bool want_debug, vk_stats *stats) {
    int rc = check_render_args(s, cam, p);
    if (rc != VK_OK) return rc;
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t world = p->tile_world ? p->tile_world : 1;
    KArgs A;
    memset(&A, 0, sizeof(A));
    A.S = s->dev;
    A.C.cam = *cam;
    A.C.width = p->width; A.C.height = p->height; A.C.spp = p->samples_per_pixel; A.C.max_depth = p->max_depth;
    A.C.seed = p->seed; A.C.integrator = p->integrator; A.C.background = p->background;
    A.C.bg[0] = p->background_color[0]; A.C.bg[1] = p->background_color[1]; A.C.bg[2] = p->background_color[2];
    A.out = d_out;
    A.tiles_x = (p->width + TILE - 1) / TILE; A.tiles_y = (p->height + TILE - 1) / TILE;
    uint32_t tiles = A.tiles_x * A.tiles_y;
    A.tile_rank = p->tile_rank; A.tile_world = world;
    A.n_local_tiles = tiles > p->tile_rank ? (tiles - p->tile_rank + world - 1) / world : 0;
    A.n_chunks = choose_chunks(s, p);
    A.counter = s->counter;
    A.shade_defer = s->hot_bytes > (4u << 20) ? 1u : SHADE_DEFER;
    if (const char *e = getenv("VK_SHADE_DEFER")) { int v = atoi(e); if (v >= 1 && v <= 64) A.shade_defer = (uint32_t)v; }   // diagnostics
    // Heavy-first tile order.  A launch ends when its slowest unit does, and tile costs are skewed (C2's glass tiles cost 8x
    // the mean, units that start mid-launch finish last).  So a probe launch of a few samples per pixel times every tile
    // (the COST build of the same kernel variant), three tiny kernels bucket-sort the tiles dearest first, and the real
    // launch takes its units in that order (longest processing time first).  The order never changes a pixel.
    // One rank's 1/8 share of C2: 78.1 -> 68.6 ms (ideal 64.9); whole frame 522 -> 519 ms including the probe.
    bool use_order = A.n_local_tiles >= 64 && p->samples_per_pixel >= 64;
    if (const char *e = getenv("VK_TILE_ORDER")) use_order = use_order && e[0] != '0';      // diagnostics
    if (use_order) {
        if (tiles > s->tile_cost_n) {
            if (s->tile_cost) HIP_TRY(hipFree(s->tile_cost));
            s->tile_cost = nullptr; s->tile_cost_n = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->tile_cost), (size_t)tiles * sizeof(uint32_t)));
            s->tile_cost_n = tiles;
        }
        if (A.n_local_tiles > s->tile_order_n) {
            if (s->tile_order) HIP_TRY(hipFree(s->tile_order));
            s->tile_order = nullptr; s->tile_order_n = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->tile_order), (size_t)A.n_local_tiles * sizeof(uint32_t)));
            s->tile_order_n = A.n_local_tiles;
        }
        if (!s->order_hist) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->order_hist), ORDER_BUCKETS * sizeof(uint32_t)));
    }
    size_t n_pixels = (size_t)p->width * p->height;
    if (A.n_chunks > 1) {
        size_t need = (size_t)A.n_chunks * n_pixels * 3 * sizeof(float);
        if (need > s->partial_bytes) {
            if (s->partial) HIP_TRY(hipFree(s->partial));
            s->partial = nullptr; s->partial_bytes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->partial), need));
            s->partial_bytes = need;
        }
        A.partial = s->partial;
    }
    if (want_debug) {
        size_t need = n_pixels * p->samples_per_pixel * sizeof(float4);
        if (need > s->debug_bytes) {
            if (s->debug) HIP_TRY(hipFree(s->debug));
            s->debug = nullptr; s->debug_bytes = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->debug), need));
            s->debug_bytes = need;
        }
        HIP_TRY(hipMemsetAsync(s->debug, 0, need, st));
        A.debug = s->debug;
    }
    // LDS residency of the hot records
    bool lds = s->lds_bytes != 0;
    const uint32_t waves_per_wg = s->wg_threads / 64;
    size_t shmem = (size_t)waves_per_wg * per_wave_lds_bytes(pick_variant(s->host.features));
    if (lds) { A.lds_items = s->dev.n_items; A.lds_spheres = s->dev.n_spheres; A.lds_boxes = s->dev.n_boxes; shmem += s->lds_bytes; }
    // persistent grid: enough workgroups to fill the chip, never more than there are units
    uint32_t n_units = A.n_local_tiles * A.n_chunks;
    uint32_t grid = (uint32_t)s->num_cus * s->wgs_per_cu;
    uint32_t need_wgs = (n_units + waves_per_wg - 1) / waves_per_wg;
    if (grid > need_wgs) grid = need_wgs;
    if (grid < 1) grid = 1;

    HIP_TRY(hipEventRecord(s->ev0, st));
    uint32_t F = pick_variant(s->host.features) | (p->integrator == VK_INTEGRATOR_PDF ? (uint32_t)VKF_INTEG_PDF : 0u);
    if (use_order && !s->want_phase_stats) {
        KArgs B = A;                                   // the probe: the same view at 1..4 samples per pixel, one unit per tile
        // (C2, one rank's 1/8 share: probe of 1 / 2 / 4 / 8 / 16 spp -> 68.6 / 69.0 / 69.8 / 70.5 / 73.0 ms: more samples cost more than they sort better)
        B.C.spp = p->samples_per_pixel / 1024u; B.C.spp = B.C.spp < 1u ? 1u : (B.C.spp > 4u ? 4u : B.C.spp);
        if (const char *e = getenv("VK_PROBE_SPP")) { int v = atoi(e); if (v >= 1 && (uint32_t)v <= p->samples_per_pixel) B.C.spp = (uint32_t)v; }   // diagnostics
        B.n_chunks = 1; B.partial = nullptr; B.debug = nullptr; B.tile_order = nullptr;
        B.tile_cost = s->tile_cost;                    // its pixels land in d_out and are overwritten by the real launch
        HIP_TRY(hipMemsetAsync(s->tile_cost, 0, (size_t)tiles * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(s->counter, 0, sizeof(uint32_t), st));
        uint32_t pgrid = (uint32_t)s->num_cus * s->wgs_per_cu, pneed = (A.n_local_tiles + waves_per_wg - 1) / waves_per_wg;
        if (pgrid > pneed) pgrid = pneed;
        rc = launch_by_features(s, F, B, lds, dim3(pgrid), shmem, st, true);
        if (rc != VK_OK) return rc;
        uint32_t nb = (A.n_local_tiles + 255u) / 256u;
        HIP_TRY(hipMemsetAsync(s->order_hist, 0, ORDER_BUCKETS * sizeof(uint32_t), st));
        hipLaunchKernelGGL(order_hist_kernel, dim3(nb), dim3(256), 0, st, (const uint32_t *)s->tile_cost, A.n_local_tiles, A.tile_rank, A.tile_world, s->order_hist);
        hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1), 0, st, s->order_hist);
        hipLaunchKernelGGL(order_scatter_kernel, dim3(nb), dim3(256), 0, st, s->tile_cost, A.n_local_tiles, A.tile_rank, A.tile_world, s->order_hist, s->tile_order);
        HIP_TRY(hipGetLastError());
        A.tile_order = s->tile_order;
    }
    HIP_TRY(hipMemsetAsync(s->counter, 0, sizeof(uint32_t), st));
    if (s->want_phase_stats) {
        const uint32_t FULLPDF = VKF_ALL_SCENE | VKF_INTEG_PDF;
        if ((F != 0u && F != FULLPDF) || !lds) return fail(VK_ERR_UNSUPPORTED, "phase statistics are only built for the LDS-resident sphere-only and full/PDF variants");
        if (!s->phase_stats) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->phase_stats), 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(s->phase_stats, 0, 16 * sizeof(unsigned long long), st));
        A.phase_stats = s->phase_stats;
        if (F == 0u) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<0u, true, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            hipLaunchKernelGGL((render_kernel<0u, true, 6, true>), dim3(grid), dim3(s->wg_threads), shmem, st, A);
        } else {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<FULLPDF, true, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            hipLaunchKernelGGL((render_kernel<FULLPDF, true, 4, true>), dim3(grid), dim3(s->wg_threads), shmem, st, A);
        }
        HIP_TRY(hipGetLastError());
        F = 0xFFFFFFFFu;   // launched
    }
    if (F != 0xFFFFFFFFu) rc = launch_by_features(s, F, A, lds, dim3(grid), shmem, st, false);
    if (rc != VK_OK) return rc;
    uint32_t launches = 1;
    if (A.n_chunks > 1) {
        uint32_t blocks = (uint32_t)((n_pixels + 255) / 256);
        hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, st, (const float *)A.partial, d_out, p->width, p->height, A.n_chunks,
                           p->samples_per_pixel, A.tiles_x, A.tile_rank, A.tile_world);
        HIP_TRY(hipGetLastError());
        launches = 2;
    }
    HIP_TRY(hipEventRecord(s->ev1, st));
    s->last_timed = true;
    if (stats) {
        // samples of this partition
        uint64_t px = 0;
        for (uint32_t t = p->tile_rank; t < tiles; t += world) {
            uint32_t tx = (t % A.tiles_x) * TILE, ty = (t / A.tiles_x) * TILE;
            px += (uint64_t)std::min<uint32_t>(TILE, p->width - tx) * std::min<uint32_t>(TILE, p->height - ty);
        }
        stats->samples = px * p->samples_per_pixel;
        stats->kernel_launches = launches;
        stats->scene_in_lds = lds ? 1u : 0u;
        stats->kernel_ms = 0.0; stats->seconds = 0.0;
    }
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

int vk_scene_create(const vk_scene_desc *desc, int device, vk_scene **out) {
    if (!out) return fail(VK_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(VK_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(VK_ERR_BAD_ARG, "device index out of range");
    hipDeviceProp_t pr;
    HIP_TRY(hipGetDeviceProperties(&pr, device));
    if (strncmp(pr.gcnArchName, "gfx950", 6) != 0) return fail(VK_ERR_NO_DEVICE, std::string("device is ") + pr.gcnArchName + ", this build targets gfx950 only");
    vk_scene *s = new vk_scene;
    s->device = device;
    std::string err;
    int rc = linearize(desc, s->host, err);
    if (rc != VK_OK) { delete s; return fail(rc, err); }
    HIP_TRY(hipSetDevice(device));
    s->num_cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    const LinearScene &H = s->host;
    DScene &D = s->dev;
    memset(&D, 0, sizeof(D));
#define UP(vec, field) do { rc = upload(s, H.vec, D.field); if (rc != VK_OK) { vk_scene_destroy(s); return rc; } } while (0)
    UP(items, items); UP(spheres, spheres); UP(sphere_mat, sphere_mat); UP(moving, moving); UP(rects, rects); UP(boxes, boxes);
    UP(lists, lists); UP(list_refs, list_refs); UP(media, media); UP(instances, instances);
    UP(materials, materials); UP(textures, textures); UP(images, images); UP(image_bytes, image_bytes);
    UP(perlins, perlins); UP(lights, lights);
#undef UP
    D.n_items = (uint32_t)H.items.size(); D.n_world_items = H.world_items; D.n_spheres = (uint32_t)H.spheres.size();
    D.n_lights = (uint32_t)H.lights.size(); D.features = H.features; D.n_boxes = (uint32_t)H.boxes.size();
    void *c = nullptr;
    if (hipMalloc(&c, 256) != hipSuccess) { vk_scene_destroy(s); return fail(VK_ERR_OOM, "hipMalloc failed"); }
    s->counter = reinterpret_cast<uint32_t *>(c);
    if (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) { vk_scene_destroy(s); return fail(VK_ERR_HIP, "hipEventCreate failed"); }
    // LDS residency: items + spheres + accumulators must leave room for >= 2 workgroups per CU
    size_t hot = H.items.size() * sizeof(DItem) + H.spheres.size() * sizeof(DSphere) + H.boxes.size() * sizeof(DBox);
    s->hot_bytes = hot;
    plan_residency(s, hot);
    *out = s;
    return VK_OK;
}

void vk_scene_destroy(vk_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->counter) (void)hipFree(s->counter);
    if (s->fb) (void)hipFree(s->fb);
    if (s->partial) (void)hipFree(s->partial);
    if (s->debug) (void)hipFree(s->debug);
    if (s->phase_stats) (void)hipFree(s->phase_stats);
    if (s->tile_cost) (void)hipFree(s->tile_cost);
    if (s->tile_order) (void)hipFree(s->tile_order);
    if (s->order_hist) (void)hipFree(s->order_hist);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

int vk_scene_get_info(const vk_scene *s, vk_scene_info *out) {
    if (!s || !out) return fail(VK_ERR_BAD_ARG, "null argument");
    out->n_items = (uint32_t)s->host.items.size();
    out->n_prims = s->host.n_prims;
    out->n_instances = (uint32_t)s->host.instances.size();
    uint64_t b = 0;
    b += s->host.items.size() * sizeof(DItem) + s->host.spheres.size() * (sizeof(DSphere) + 4) + s->host.moving.size() * sizeof(DMoving) +
         s->host.rects.size() * sizeof(DRect) + s->host.lists.size() * sizeof(DList) + s->host.list_refs.size() * 4 +
         s->host.media.size() * sizeof(DMedium) + s->host.instances.size() * sizeof(DInstance) + s->host.materials.size() * sizeof(DMaterial) +
         s->host.textures.size() * sizeof(DTexture) + s->host.image_bytes.size() + s->host.perlins.size() * sizeof(DPerlin);
    out->device_bytes = b;
    out->lds_bytes = s->lds_bytes;
    out->features = pick_variant(s->host.features);
    return VK_OK;
}

int vk_render_device(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, void *d_rgb_out, void *hip_stream, vk_stats *stats_out) {
    if (!d_rgb_out) return fail(VK_ERR_BAD_ARG, "null device framebuffer");
    return enqueue_render(scene, cam, params, reinterpret_cast<float *>(d_rgb_out), reinterpret_cast<hipStream_t>(hip_stream), false, stats_out);
}

// HIP-event time (ms) of the launches enqueued by the last vk_render_device / vk_render on
// this scene; synchronises on their end event.
int vk_scene_last_kernel_ms(vk_scene *s, double *ms_out) {
    if (!s || !ms_out) return fail(VK_ERR_BAD_ARG, "null argument");
    if (!s->last_timed) return fail(VK_ERR_BAD_ARG, "no render enqueued yet");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *ms_out = (double)ms;
    return VK_OK;
}

static int render_host(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, float *rgb_out, vk_stats *stats_out, float *debug_out) {
    if (!rgb_out) return fail(VK_ERR_BAD_ARG, "null framebuffer");
    int rc = check_render_args(scene, cam, params);
    if (rc != VK_OK) return rc;
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(scene->device));
    size_t n_pixels = (size_t)params->width * params->height;
    size_t bytes = n_pixels * 3 * sizeof(float);
    if (bytes > scene->fb_bytes) {
        if (scene->fb) HIP_TRY(hipFree(scene->fb));
        scene->fb = nullptr; scene->fb_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&scene->fb), bytes));
        scene->fb_bytes = bytes;
    }
    vk_stats st;
    memset(&st, 0, sizeof(st));
    rc = enqueue_render(scene, cam, params, scene->fb, nullptr, debug_out != nullptr, &st);
    if (rc != VK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    double ms = 0.0;
    rc = vk_scene_last_kernel_ms(scene, &ms);
    if (rc != VK_OK) return rc;
    st.kernel_ms = ms;
    uint32_t world = params->tile_world ? params->tile_world : 1;
    if (world == 1) {
        HIP_TRY(hipMemcpy(rgb_out, scene->fb, bytes, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp(n_pixels * 3);
        HIP_TRY(hipMemcpy(tmp.data(), scene->fb, bytes, hipMemcpyDeviceToHost));
        uint32_t tiles_x = (params->width + TILE - 1) / TILE;
        for (uint32_t y = 0; y < params->height; y++)
            for (uint32_t x = 0; x < params->width; x++) {
                uint32_t tile = (y / TILE) * tiles_x + (x / TILE);
                if (tile % world != params->tile_rank) continue;
                size_t i = ((size_t)y * params->width + x) * 3;
                rgb_out[i] = tmp[i]; rgb_out[i + 1] = tmp[i + 1]; rgb_out[i + 2] = tmp[i + 2];
            }
    }
    if (debug_out) HIP_TRY(hipMemcpy(debug_out, scene->debug, n_pixels * params->samples_per_pixel * sizeof(float4), hipMemcpyDeviceToHost));
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats_out) *stats_out = st;
    return VK_OK;
}

int vk_render(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, float *rgb_out, vk_stats *stats_out) {
    return render_host(scene, cam, params, rgb_out, stats_out, nullptr);
}

// test hook: as vk_render, also returning every sample: samples_out[(pixel*spp + s)*4 + 0..2]
// = radiance before the finite filter, [+3] = the sample's draw count (bit pattern)
int vk_debug_render_samples(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, float *rgb_out, float *samples_out) {
    if (!samples_out) return fail(VK_ERR_BAD_ARG, "null samples buffer");
    return render_host(scene, cam, params, rgb_out, nullptr, samples_out);
}

int vk_to_color_device(vk_scene *scene, const void *d_rgb, uint32_t width, uint32_t height, void *d_rgb8_out, void *hip_stream) {
    if (!scene || !d_rgb || !d_rgb8_out) return fail(VK_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(scene->device));
    uint32_t n = width * height * 3;
    hipLaunchKernelGGL(to_color_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                       reinterpret_cast<const float *>(d_rgb), width, height, reinterpret_cast<uint8_t *>(d_rgb8_out));
    HIP_TRY(hipGetLastError());
    return VK_OK;
}

// diagnostic: render with the instrumented kernel build and return the phase scheduler's counters:
// [0] box steps executed (wave level), [1] lanes that had box work summed over those steps,
// [2] PRIM phase executions, [3] lanes with prim work in them, [4] SHADE+REFILL executions,
// [5] lanes shading or refilling in them, [6] scheduler rounds, [7] unused
int vk_debug_phase_stats(vk_scene *scene, const vk_camera *cam, const vk_render_params *params, uint64_t out[16]) {
    if (!scene || !out) return fail(VK_ERR_BAD_ARG, "null argument");
    size_t bytes = (size_t)params->width * params->height * 3 * sizeof(float);
    HIP_TRY(hipSetDevice(scene->device));
    if (bytes > scene->fb_bytes) {
        if (scene->fb) HIP_TRY(hipFree(scene->fb));
        scene->fb = nullptr; scene->fb_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&scene->fb), bytes));
        scene->fb_bytes = bytes;
    }
    scene->want_phase_stats = true;
    int rc = enqueue_render(scene, cam, params, scene->fb, nullptr, false, nullptr);
    scene->want_phase_stats = false;
    if (rc != VK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    HIP_TRY(hipMemcpy(out, scene->phase_stats, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return VK_OK;
}

// test hook: evaluate shared-math functions on the device (host arrays in/out)
int vk_debug_math(int device, int op, const float *a, const float *b, float *out, size_t n) {
    if (!a || !b || !out) return fail(VK_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&da), n * 4 + 16));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&db), n * 4 + 16));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dout), n * 4 + 16));
    HIP_TRY(hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, (const float *)da, (const float *)db, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return VK_OK;
}

}  // extern "C"
